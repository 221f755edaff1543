// The host walk of the drop-in filter (itkCuberilleImageToMeshFilter.txx: HostGradient / HostWalk / SplitQuads) through an
// interpolator whose Evaluate() is NOT the linear one: the value it returns blends the filter's image with a second,
// smoothed image.  tests/test_host.py and tests/test_gpu_parity.py hold the results to a restatement of txx:439-474 in
// Python over the oracle's pinned interpolation and gradient primitives.
//
//   host_walk walk   <vol.raw> <second.raw> <n> <iso> <thr> <step> <relax> <maxSteps> <start.raw> <nPoints> <out.raw> [threads]
//       HostWalk alone (no GPU): start points (float xyz) -> walked points
//   host_walk filter <vol.raw> <second.raw> <n> <iso> <thr> <step> <relax> <maxSteps> <outPoints.raw> <outCells.raw> <tri> [threads]
//       the whole filter with that interpolator type (GPU topology + host walk)
//   host_walk throw  <vol.raw> <second.raw> <n> <iso> <threads>
//       an interpolator that throws inside Evaluate(): Update() must throw, not terminate
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <vector>

#include "itkImage.h"
#include "itkMesh.h"
#include "itkCuberilleImageToMeshFilter.h"

typedef itk::Image<float, 3> ImageType;
typedef itk::Mesh<float, 3> MeshType;

template <class TImage> class BlendInterpolator : public itk::LinearInterpolateImageFunction<TImage, double>
{
public:
  typedef BlendInterpolator Self;
  typedef itk::LinearInterpolateImageFunction<TImage, double> Superclass;
  typedef itk::SmartPointer<Self> Pointer;
  itkNewMacro(Self);
  typename Superclass::Pointer second;       // linear interpolator over the smoothed image
  long throwAfter;                           // >= 0: Evaluate throws once this many calls have been made
  mutable long calls;
  typename Superclass::OutputType Evaluate(const typename Superclass::PointType &p) const
  {
    if (throwAfter >= 0 && __sync_fetch_and_add(&calls, 1) >= throwAfter) throw std::runtime_error("interpolator gave up");
    const double a = Superclass::Evaluate(p), b = second->Evaluate(p);
    return 0.25 * a + 0.75 * b;
  }
protected:
  BlendInterpolator() : throwAfter(-1), calls(0) {}
};

static ImageType::Pointer load(const char *path, int n)
{
  ImageType::Pointer image = ImageType::New();
  ImageType::RegionType region;
  ImageType::IndexType start;
  ImageType::SizeType size;
  start.Fill(0);
  size.Fill(n);
  region.SetIndex(start);
  region.SetSize(size);
  image->SetRegions(region);
  image->Allocate();
  FILE *f = std::fopen(path, "rb");
  if (!f || std::fread(image->GetBufferPointer(), sizeof(float), (size_t)n * n * n, f) != (size_t)n * n * n)
    { std::fprintf(stderr, "cannot read %s\n", path); std::exit(3); }
  std::fclose(f);
  return image;
}

static void dump(const char *path, const void *p, size_t bytes)
{
  FILE *f = std::fopen(path, "wb");
  if (!f || std::fwrite(p, 1, bytes, f) != bytes) { std::fprintf(stderr, "cannot write %s\n", path); std::exit(3); }
  std::fclose(f);
}

int main(int argc, char **argv)
{
  if (argc < 6) { std::fprintf(stderr, "usage: see the head of host_walk.cxx\n"); return 1; }
  const char *mode = argv[1];
  const int n = std::atoi(argv[4]);
  ImageType::Pointer image = load(argv[2], n), smooth = load(argv[3], n);
  typedef BlendInterpolator<ImageType> InterpolatorType;
  InterpolatorType::Pointer interp = InterpolatorType::New();
  interp->SetInputImage(image);
  interp->second = InterpolatorType::Superclass::New();
  interp->second->SetInputImage(smooth);
  try
    {
    if (!std::strcmp(mode, "walk"))
      {
      const size_t np = (size_t)std::atoll(argv[11]);
      std::vector<float> pts(3 * np);
      FILE *f = std::fopen(argv[10], "rb");
      if (!f || std::fread(&pts[0], sizeof(float), 3 * np, f) != 3 * np) { std::fprintf(stderr, "cannot read start points\n"); return 3; }
      std::fclose(f);
      itk::cuberille_detail::HostGradient<ImageType> gradient(image.GetPointer());
      itk::cuberille_detail::HostWalk<ImageType, InterpolatorType> walk =
        {&gradient, interp.GetPointer(), &pts[0], std::atof(argv[5]), std::atof(argv[6]), std::atof(argv[7]), std::atof(argv[8]),
         (unsigned int)std::atoi(argv[9])};
      itk::cuberille_detail::ParallelRanges(np, walk, argc > 13 ? (unsigned int)std::atoi(argv[13]) : 1u);
      dump(argv[12], &pts[0], sizeof(float) * 3 * np);
      return 0;
      }
    typedef itk::CuberilleImageToMeshFilter<ImageType, MeshType, InterpolatorType> FilterType;
    FilterType::Pointer filter = FilterType::New();
    filter->SetInput(image);
    filter->SetInterpolator(interp);
    filter->SetIsoSurfaceValue((float)std::atof(argv[5]));
    if (!std::strcmp(mode, "throw"))
      {
      interp->throwAfter = 200;
      filter->SetHostWalkThreads((unsigned int)std::atoi(argv[6]));
      try { filter->Update(); }
      catch (std::exception &e) { std::cout << "caught: " << e.what() << std::endl; return 0; }
      std::cout << "no exception" << std::endl;
      return 4;
      }
    filter->SetProjectVertexSurfaceDistanceThreshold(std::atof(argv[6]));
    filter->SetProjectVertexStepLength(std::atof(argv[7]));
    filter->SetProjectVertexStepLengthRelaxationFactor(std::atof(argv[8]));
    filter->SetProjectVertexMaximumNumberOfSteps((unsigned int)std::atoi(argv[9]));
    filter->SetGenerateTriangleFaces(std::atoi(argv[12]) != 0);
    if (argc > 13) filter->SetHostWalkThreads((unsigned int)std::atoi(argv[13]));
    filter->Update();
    MeshType::Pointer mesh = filter->GetOutput();
    std::vector<float> pts(3 * mesh->GetNumberOfPoints());
    for (unsigned long i = 0; i < mesh->GetNumberOfPoints(); i++)
      {
      MeshType::PointType p;
      mesh->GetPoint(i, &p);
      for (int k = 0; k < 3; k++) pts[3 * i + k] = p[k];
      }
    std::vector<unsigned long long> ids;
    for (unsigned long c = 0; c < mesh->GetNumberOfCells(); c++)
      {
      MeshType::CellAutoPointer cell;
      mesh->GetCell(c, cell);
      for (unsigned int k = 0; k < cell->GetNumberOfPoints(); k++) ids.push_back(cell->PointIdsBegin()[k]);
      }
    dump(argv[10], pts.empty() ? 0 : &pts[0], sizeof(float) * pts.size());
    dump(argv[11], ids.empty() ? 0 : &ids[0], sizeof(unsigned long long) * ids.size());
    std::cout << mesh->GetNumberOfPoints() << " " << mesh->GetNumberOfCells() << std::endl;
    return 0;
    }
  catch (itk::ExceptionObject &e)
    {
    std::cerr << e << std::endl;
    return 2;
    }
}
