// Instantiates the drop-in filter for pixel types other than the reference driver's unsigned char
// ("template must still compile for other pixel types") and runs each on a small
// synthetic sphere.  Prints one line per instantiation: "<type> points cells euler"; exit code 0 if every
// mesh is a closed genus-0 surface (V - E + F = 2 for quads) with the same topology for every type.
#include <cmath>
#include <iostream>
#include <set>
#include <utility>
#include <vector>

#include "itkImage.h"
#include "itkMesh.h"
#include "itkDefaultDynamicMeshTraits.h"
#include "itkCuberilleImageToMeshFilter.h"

// A user's interpolator type (TInterpolator): a distinct class the filter has never seen.
// It inherits the linear Evaluate, so the host walk that calls it must land exactly where the GPU walk lands.
template <class TImage> class UserInterpolator : public itk::LinearInterpolateImageFunction<TImage, double>
{
public:
  typedef UserInterpolator Self;
  typedef itk::SmartPointer<Self> Pointer;
  itkNewMacro(Self);
  mutable unsigned long calls;
  typename itk::LinearInterpolateImageFunction<TImage, double>::OutputType
  Evaluate(const typename itk::LinearInterpolateImageFunction<TImage, double>::PointType &p) const
  {
    return itk::LinearInterpolateImageFunction<TImage, double>::Evaluate(p);
  }
protected:
  UserInterpolator() : calls(0) {}
};

// the filter with the user's interpolator type against the filter with the default one, same image, unit spacing
static bool user_interpolator_matches(bool triangles)
{
  typedef itk::Image<float, 3> ImageType;
  typedef itk::Mesh<float, 3> MeshType;
  typedef itk::CuberilleImageToMeshFilter<ImageType, MeshType> GpuFilter;
  typedef itk::CuberilleImageToMeshFilter<ImageType, MeshType, UserInterpolator<ImageType> > UserFilter;
  const int n = 48;
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
  for (int z = 0; z < n; z++)
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++)
        {
        ImageType::IndexType idx;
        idx[0] = x; idx[1] = y; idx[2] = z;
        const double r = std::sqrt((x - 23.3) * (x - 23.3) + (y - 23.6) * (y - 23.6) + (z - 23.1) * (z - 23.1));
        image->SetPixel(idx, static_cast<float>(17.0 - r + 0.3 * std::sin(0.9 * x) * std::cos(0.7 * y)));
        }
  GpuFilter::Pointer a = GpuFilter::New();
  UserFilter::Pointer b = UserFilter::New();
  a->SetInput(image); b->SetInput(image);
  a->SetIsoSurfaceValue(0.0f); b->SetIsoSurfaceValue(0.0f);
  a->SetGenerateTriangleFaces(triangles); b->SetGenerateTriangleFaces(triangles);
  a->SetProjectVertexSurfaceDistanceThreshold(0.01); b->SetProjectVertexSurfaceDistanceThreshold(0.01);
  a->Update();
  b->Update();
  MeshType::Pointer ma = a->GetOutput(), mb = b->GetOutput();
  bool same = ma->GetNumberOfPoints() == mb->GetNumberOfPoints() && ma->GetNumberOfCells() == mb->GetNumberOfCells();
  unsigned long moved = 0;
  for (unsigned long i = 0; same && i < ma->GetNumberOfPoints(); i++)
    {
    MeshType::PointType pa, pb;
    ma->GetPoint(i, &pa); mb->GetPoint(i, &pb);
    for (int k = 0; k < 3; k++) same = same && (pa[k] == pb[k]);
    if (pa[0] != std::floor(pa[0]) + 0.5f) moved++;
    }
  for (unsigned long c = 0; same && c < ma->GetNumberOfCells(); c++)
    {
    MeshType::CellAutoPointer ca, cb;
    ma->GetCell(c, ca); mb->GetCell(c, cb);
    same = ca->GetNumberOfPoints() == cb->GetNumberOfPoints();
    for (unsigned int k = 0; same && k < ca->GetNumberOfPoints(); k++) same = ca->PointIdsBegin()[k] == cb->PointIdsBegin()[k];
    }
  std::cout << "user-interpolator" << (triangles ? "-triangles " : " ") << ma->GetNumberOfPoints() << " " << ma->GetNumberOfCells() << " "
            << (same && moved > 100 ? 2 : -1) << std::endl;
  return same && moved > 100;
}

template <class TPixel>
bool run(const char *name, bool triangles, unsigned long &points, unsigned long &cells)
{
  typedef itk::Image<TPixel, 3> ImageType;
  typedef itk::Mesh<float, 3> MeshType;
  typedef itk::CuberilleImageToMeshFilter<ImageType, MeshType> FilterType;
  const int n = 40;
  typename ImageType::Pointer image = ImageType::New();
  typename ImageType::RegionType region;
  typename ImageType::IndexType start;
  typename ImageType::SizeType size;
  start.Fill(0);
  size.Fill(n);
  region.SetIndex(start);
  region.SetSize(size);
  image->SetRegions(region);
  image->Allocate();
  typename ImageType::SpacingType spacing;
  spacing[0] = 0.5; spacing[1] = 1.0; spacing[2] = 2.0;
  image->SetSpacing(spacing);
  for (int z = 0; z < n; z++)
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++)
        {
        typename ImageType::IndexType idx;
        idx[0] = x; idx[1] = y; idx[2] = z;
        const double r = std::sqrt((x - 19.3) * (x - 19.3) + (y - 19.6) * (y - 19.6) + (z - 19.1) * (z - 19.1));
        const double v = 100.0 - 6.0 * r;                            // 100 at the centre, clamped at 0 far out
        image->SetPixel(idx, static_cast<TPixel>(v > 0.0 ? v : 0.0));
        }
  typename FilterType::Pointer filter = FilterType::New();
  filter->SetInput(image);
  filter->SetIsoSurfaceValue(static_cast<TPixel>(30));
  filter->SetGenerateTriangleFaces(triangles);
  filter->SetProjectVertexSurfaceDistanceThreshold(0.5);
  filter->Update();
  typename MeshType::Pointer mesh = filter->GetOutput();
  points = mesh->GetNumberOfPoints();
  cells = mesh->GetNumberOfCells();
  std::set<std::pair<unsigned long, unsigned long> > edges;
  for (unsigned long c = 0; c < cells; c++)
    {
    typename MeshType::CellAutoPointer cell;
    mesh->GetCell(c, cell);
    const unsigned int k = cell->GetNumberOfPoints();
    typename MeshType::CellType::PointIdConstIterator it = cell->PointIdsBegin();
    for (unsigned int i = 0; i < k; i++)
      {
      unsigned long a = it[i], b = it[(i + 1) % k];
      if (a > b) std::swap(a, b);
      edges.insert(std::make_pair(a, b));
      }
    }
  const long euler = (long)points - (long)edges.size() + (long)cells;
  std::cout << name << " " << points << " " << cells << " " << euler << std::endl;
  return euler == 2 && points > 0;
}

// Ownership of the bulk-filled cells (txx:309-329 hands every cell to the mesh): the mesh must carry its cells through
// DisconnectPipeline() and past the end of the filter (Testing/CuberilleTest01.cxx:161-162 takes the output and lets the
// filter go), a second Update() of the same filter must replace them, and an output that is re-initialised must let go.
static bool mesh_outlives_the_filter()
{
  typedef itk::Image<unsigned char, 3> ImageType;
  typedef itk::Mesh<unsigned char, 3> MeshType;
  typedef itk::CuberilleImageToMeshFilter<ImageType, MeshType> FilterType;
  const int n = 24;
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
  for (int z = 0; z < n; z++)
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++)
        {
        ImageType::IndexType idx;
        idx[0] = x; idx[1] = y; idx[2] = z;
        const double r = std::sqrt((x - 11.3) * (x - 11.3) + (y - 11.6) * (y - 11.6) + (z - 11.1) * (z - 11.1));
        image->SetPixel(idx, static_cast<unsigned char>(r < 7.0 ? 200 : 0));
        }
  MeshType::Pointer kept;
  unsigned long cells = 0, points = 0;
  unsigned long long checksum = 0;
  {
    FilterType::Pointer filter = FilterType::New();
    filter->SetInput(image);
    filter->SetIsoSurfaceValue(100);
    filter->Update();
    const unsigned long first = filter->GetOutput()->GetNumberOfCells();
    filter->SetGenerateTriangleFaces(false);          // a second run of the same filter: the first run's cells are replaced
    filter->Update();
    if (filter->GetOutput()->GetNumberOfCells() * 2 != first) return false;
    filter->SetGenerateTriangleFaces(true);
    filter->Update();
    kept = filter->GetOutput();
    kept->DisconnectPipeline();
    cells = kept->GetNumberOfCells();
    points = kept->GetNumberOfPoints();
    if (cells != first || cells == 0) return false;
    for (unsigned long c = 0; c < cells; c++)
      {
      MeshType::CellAutoPointer cell;
      kept->GetCell(c, cell);
      MeshType::CellType::PointIdConstIterator it = cell->PointIdsBegin();
      for (unsigned int i = 0; i < cell->GetNumberOfPoints(); i++) checksum = checksum * 1000003ull + it[i];
      }
  }                                                   // the filter (and its GPU context) is gone
  unsigned long long again = 0;
  for (unsigned long c = 0; c < cells; c++)
    {
    MeshType::CellAutoPointer cell;
    if (!kept->GetCell(c, cell) || cell->GetNumberOfPoints() != 3) return false;
    MeshType::CellType::PointIdConstIterator it = cell->PointIdsBegin();
    for (unsigned int i = 0; i < 3; i++)
      {
      if (it[i] >= points) return false;
      again = again * 1000003ull + it[i];
      }
    }
  if (again != checksum) return false;
  kept->Initialize();                                 // lets go of the cell pointers; the slab goes with the mesh
  if (kept->GetNumberOfCells() != 0) return false;
  kept = 0;
  std::cout << "mesh-outlives-filter " << points << " " << cells << " 2" << std::endl;
  return true;
}

// A mesh whose containers are NOT vectors (itk::DefaultDynamicMeshTraits: MapContainers): the filter fills it element by
// element through SetPoint / SetCell, as the reference does (txx:275, 309-329), and the mesh equals the static-traits one
// (round-4 advisor finding: the bulk fill alone did not compile for such a mesh).
static bool dynamic_traits_mesh_equals_static()
{
  typedef itk::Image<unsigned char, 3> ImageType;
  typedef itk::Mesh<unsigned char, 3> StaticMesh;
  typedef itk::Mesh<unsigned char, 3, itk::DefaultDynamicMeshTraits<unsigned char, 3, 3> > DynamicMesh;
  typedef itk::CuberilleImageToMeshFilter<ImageType, StaticMesh> StaticFilter;
  typedef itk::CuberilleImageToMeshFilter<ImageType, DynamicMesh> DynamicFilter;
  const int n = 20;
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
  for (int z = 0; z < n; z++)
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++)
        {
        ImageType::IndexType idx;
        idx[0] = x; idx[1] = y; idx[2] = z;
        const double r = std::sqrt((x - 9.3) * (x - 9.3) + (y - 9.6) * (y - 9.6) + (z - 9.1) * (z - 9.1));
        image->SetPixel(idx, static_cast<unsigned char>(r < 6.0 ? 200 : 0));
        }
  bool same = true;
  for (int tri = 0; tri < 2 && same; tri++)
    {
    // (second round: the filters set their GPU context up inside the first Update() instead of in the constructor, and the
    //  static one gives the context's host copy of the mesh back as soon as its itk::Mesh is filled -- same meshes)
    StaticFilter::SetEagerDeviceSetup(tri == 0);
    StaticFilter::Pointer a = StaticFilter::New();
    DynamicFilter::Pointer b = DynamicFilter::New();
    StaticFilter::SetEagerDeviceSetup(true);
    a->SetReleaseHostMeshAfterFill(tri != 0);
    a->SetInput(image); b->SetInput(image);
    a->SetIsoSurfaceValue(100); b->SetIsoSurfaceValue(100);
    a->SetGenerateTriangleFaces(tri != 0); b->SetGenerateTriangleFaces(tri != 0);
    a->Update(); b->Update();
    b->Update();                                      // a second fill of the same dynamic mesh replaces the first
    StaticMesh::Pointer ma = a->GetOutput();
    DynamicMesh::Pointer mb = b->GetOutput();
    same = ma->GetNumberOfPoints() == mb->GetNumberOfPoints() && ma->GetNumberOfCells() == mb->GetNumberOfCells() &&
           ma->GetNumberOfCells() > 0 && mb->GetCellsAllocationMethod() == DynamicMesh::CellsAllocatedDynamicallyCellByCell;
    for (unsigned long i = 0; same && i < ma->GetNumberOfPoints(); i++)
      {
      StaticMesh::PointType pa;
      DynamicMesh::PointType pb;
      same = ma->GetPoint(i, &pa) && mb->GetPoint(i, &pb);
      for (int k = 0; same && k < 3; k++) same = pa[k] == pb[k];
      }
    for (unsigned long c = 0; same && c < ma->GetNumberOfCells(); c++)
      {
      StaticMesh::CellAutoPointer ca;
      DynamicMesh::CellAutoPointer cb;
      same = ma->GetCell(c, ca) && mb->GetCell(c, cb) && ca->GetNumberOfPoints() == cb->GetNumberOfPoints();
      for (unsigned int k = 0; same && k < ca->GetNumberOfPoints(); k++) same = ca->PointIdsBegin()[k] == cb->PointIdsBegin()[k];
      }
    std::cout << "dynamic-traits " << (tri ? "triangles " : "quads ") << mb->GetNumberOfPoints() << " " << mb->GetNumberOfCells()
              << (same ? " same" : " DIFFERENT") << std::endl;
    }
  return same;
}

// SetReproduceStaleGradient (not in the reference: its behaviour -- quirk Q3, txx:484 -- on request): a filter object that has
// projected once goes on walking along that first input's gradient.  Self-consistency here (a fresh filter, and the same
// filter with the switch off again, give the second input's own mesh; the stale one differs in coordinates only); with a file
// name the stale mesh is written out and tests/test_gpu_boundary.py holds it against the oracle's run_after.
static itk::Image<float, 3>::Pointer stale_field(int nx, int ny, int nz, double cx, double radius)
{
  typedef itk::Image<float, 3> ImageType;
  ImageType::Pointer image = ImageType::New();
  ImageType::RegionType region;
  ImageType::IndexType start;
  ImageType::SizeType size;
  start.Fill(0);
  size[0] = nx; size[1] = ny; size[2] = nz;
  region.SetIndex(start);
  region.SetSize(size);
  image->SetRegions(region);
  image->Allocate();
  for (int z = 0; z < nz; z++)
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++)
        {
        ImageType::IndexType idx;
        idx[0] = x; idx[1] = y; idx[2] = z;
        const double r = std::sqrt((x - cx) * (x - cx) + (y - 10.25) * (y - 10.25) + (z - 9.5) * (z - 9.5));
        image->SetPixel(idx, static_cast<float>(radius - r + 0.03125 * ((x * 7 + y * 13 + z * 5) % 11)));
        }
  return image;
}

template <class TMesh> static std::vector<float> flat_points(const TMesh *m)
{
  std::vector<float> out;
  for (unsigned long i = 0; i < m->GetNumberOfPoints(); i++)
    {
    typename TMesh::PointType p;
    m->GetPoint(i, &p);
    for (int k = 0; k < 3; k++) out.push_back(p[k]);
    }
  return out;
}

static unsigned long differing_points(const std::vector<float> &a, const std::vector<float> &b)
{
  unsigned long n = 0;
  for (size_t i = 0; i + 2 < a.size() && i + 2 < b.size(); i += 3)
    if (a[i] != b[i] || a[i + 1] != b[i + 1] || a[i + 2] != b[i + 2]) n++;
  return n;
}

static bool stale_gradient_switch(const char *vtkName)
{
  typedef itk::Image<float, 3> ImageType;
  typedef itk::Mesh<float, 3> MeshType;
  typedef itk::CuberilleImageToMeshFilter<ImageType, MeshType> FilterType;
  ImageType::Pointer first = stale_field(26, 22, 20, 12.5, 7.0), second = stale_field(31, 24, 21, 15.0, 8.5);
  FilterType::Pointer stale = FilterType::New(), fresh = FilterType::New();
  FilterType *both[2] = {stale.GetPointer(), fresh.GetPointer()};
  for (int i = 0; i < 2; i++)
    {
    both[i]->SetIsoSurfaceValue(0.0f);
    both[i]->GenerateTriangleFacesOff();
    both[i]->SetProjectVertexSurfaceDistanceThreshold(0.01);
    both[i]->SetProjectVertexStepLength(0.25);
    }
  stale->ReproduceStaleGradientOn();
  stale->SetInput(first);
  stale->Update();
  stale->SetInput(second);
  stale->Update();
  if (vtkName) stale->WriteLastMeshAsVTKPolyData(vtkName, 2);
  const std::vector<float> kept = flat_points(stale->GetOutput());
  const unsigned long keptCells = stale->GetOutput()->GetNumberOfCells();
  fresh->SetInput(second);
  fresh->Update();
  const std::vector<float> own = flat_points(fresh->GetOutput());
  const unsigned long moved = differing_points(kept, own);
  bool ok = kept.size() == own.size() && keptCells == fresh->GetOutput()->GetNumberOfCells() && moved > 0;
  stale->ReproduceStaleGradientOff();              // (Modified(): the next Update() runs again, on the input's own gradient)
  stale->Update();
  const std::vector<float> back = flat_points(stale->GetOutput());
  ok = ok && back.size() == own.size() && differing_points(back, own) == 0;
  std::cout << "stale-gradient " << kept.size() / 3 << " " << keptCells << " moved " << moved
            << (ok ? " consistent" : " INCONSISTENT") << std::endl;
  return ok;
}

// A buffered region that does not start at index 0 (what a crop or a paste leaves behind): the filter hands ITK's own origin and
// the region's start index to the library (cuberille_image_desc::index_start) instead of moving the origin.  With unit spacing
// and integral origins both descriptions are exact: the mesh of an image whose region starts at (5, -3, 100) equals, bit for
// bit, the mesh of the same pixels in a region at 0 with the origin moved by as much.
static bool region_index_matches_moved_origin()
{
  typedef itk::Image<float, 3> ImageType;
  typedef itk::Mesh<float, 3> MeshType;
  typedef itk::CuberilleImageToMeshFilter<ImageType, MeshType> FilterType;
  const int n = 24;
  const long off[3] = {5, -3, 100};
  ImageType::Pointer img[2] = {ImageType::New(), ImageType::New()};
  for (int v = 0; v < 2; v++)
    {
    ImageType::RegionType region;
    ImageType::IndexType start;
    ImageType::SizeType size;
    ImageType::PointType origin;
    for (int k = 0; k < 3; k++) { start[k] = v ? off[k] : 0; origin[k] = v ? 2.0 : 2.0 + off[k]; }
    size.Fill(n);
    region.SetIndex(start);
    region.SetSize(size);
    img[v]->SetRegions(region);
    img[v]->SetOrigin(origin);
    img[v]->Allocate();
    for (int z = 0; z < n; z++)
      for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
          {
          ImageType::IndexType idx;
          idx[0] = x + start[0]; idx[1] = y + start[1]; idx[2] = z + start[2];
          const double r = std::sqrt((x - 11.3) * (x - 11.3) + (y - 11.6) * (y - 11.6) + (z - 11.1) * (z - 11.1));
          img[v]->SetPixel(idx, static_cast<float>(8.0 - r + 0.03125 * ((x * 7 + y * 13 + z * 5) % 11)));
          }
    }
  std::vector<float> pts[2];
  unsigned long cells[2];
  for (int v = 0; v < 2; v++)
    {
    FilterType::Pointer f = FilterType::New();
    f->SetInput(img[v]);
    f->SetIsoSurfaceValue(0.0f);
    f->SetProjectVertexSurfaceDistanceThreshold(0.01);
    f->SetProjectVertexStepLength(0.25);
    f->Update();
    pts[v] = flat_points(f->GetOutput());
    cells[v] = f->GetOutput()->GetNumberOfCells();
    }
  const bool ok = pts[0].size() == pts[1].size() && cells[0] == cells[1] && cells[0] > 0 && differing_points(pts[0], pts[1]) == 0;
  std::cout << "region-index " << pts[1].size() / 3 << " " << cells[1] << (ok ? " same" : " DIFFERENT") << std::endl;
  return ok;
}

int main(int argc, char **argv)
{
  try
    {
    bool ok = true;
    ok &= region_index_matches_moved_origin();
    ok &= stale_gradient_switch(argc > 1 ? argv[1] : 0);
    ok &= mesh_outlives_the_filter();
    ok &= dynamic_traits_mesh_equals_static();
    unsigned long p[9], c[9];
    ok &= run<unsigned char>("uchar", false, p[0], c[0]);
    ok &= run<short>("short", false, p[1], c[1]);
    ok &= run<unsigned short>("ushort", false, p[2], c[2]);
    ok &= run<int>("int", false, p[3], c[3]);
    ok &= run<float>("float", false, p[4], c[4]);
    ok &= run<double>("double", false, p[5], c[5]);
    // value >= 30 is the same set of voxels whether the field is truncated to an integer or not:
    // identical topology for every pixel type
    // the 64-bit integer pixel types (LP64 long / unsigned long, long long)
    ok &= run<long>("long", false, p[6], c[6]);
    ok &= run<unsigned long>("ulong", false, p[7], c[7]);
    ok &= run<long long>("longlong", false, p[8], c[8]);
    for (int i = 1; i < 9; i++) ok &= (p[i] == p[0]) && (c[i] == c[0]);
    unsigned long pt, ct;
    ok &= run<float>("float-triangles", true, pt, ct);
    ok &= (pt == p[4]) && (ct == 2 * c[4]);
    // TInterpolator other than the linear one the kernels implement: GPU topology, host walk through the user's class
    ok &= user_interpolator_matches(false);
    ok &= user_interpolator_matches(true);
    return ok ? 0 : 1;
    }
  catch (itk::ExceptionObject &e)
    {
    std::cerr << e << std::endl;
    return 2;
    }
}
