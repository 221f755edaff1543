// Host-side cost around the device path: runs the drop-in filter on an n^3 float
// sphere distance field and reports, per stage, what a user of the reference's driver pays after the GPU is
// done -- the itk::Mesh fill (one heap cell per face, as txx:310-329) and itk::VTKPolyDataWriter -- next to the
// flat-buffer route (WriteLastMeshAsVTKPolyData), and checks that both routes write the same bytes.
//   usage: end_to_end <n> <out-prefix> [triangles=1]
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "itkImage.h"
#include "itkMesh.h"
#include "itkVTKPolyDataWriter.h"
#include "itkCuberilleImageToMeshFilter.h"

static double now()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool same_bytes(const std::string &a, const std::string &b)
{
  std::ifstream fa(a.c_str(), std::ios::binary), fb(b.c_str(), std::ios::binary);
  std::vector<char> ba(1 << 20), bb(1 << 20);
  while (fa && fb)
    {
    fa.read(&ba[0], ba.size());
    fb.read(&bb[0], bb.size());
    if (fa.gcount() != fb.gcount() || !std::equal(ba.begin(), ba.begin() + fa.gcount(), bb.begin())) return false;
    }
  return fa.eof() && fb.eof();
}

int main(int argc, char *argv[])
{
  if (argc < 3) { std::cerr << "usage: end_to_end <n> <out-prefix> [triangles]" << std::endl; return 2; }
  const int n = std::atoi(argv[1]);
  const std::string prefix = argv[2];
  const bool triangles = argc > 3 ? std::atoi(argv[3]) != 0 : true;
  typedef itk::Image<float, 3> ImageType;
  typedef itk::Mesh<float, 3> MeshType;
  typedef itk::CuberilleImageToMeshFilter<ImageType, MeshType> FilterType;
  try
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
    float *px = image->GetBufferPointer();
    const double c = 0.5 * (n - 1), R = 0.4 * n;
    for (int z = 0; z < n; z++)
      for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
          {
          const double dx = x - (c + 0.25), dy = y - (c + 0.125), dz = z - (c + 0.0625);
          px[((size_t)z * n + y) * n + x] = static_cast<float>(R - std::sqrt(dx * dx + dy * dy + dz * dz));
          }
    FilterType::Pointer filter = FilterType::New();
    filter->SetInput(image);
    filter->SetIsoSurfaceValue(0.0f);
    filter->SetGenerateTriangleFaces(triangles);
    filter->SetProjectVertexSurfaceDistanceThreshold(0.05);
    filter->SetProjectVertexStepLength(0.25);
    double t0 = now();
    filter->Update();                                  // first call also creates the context and its workspace
    const double tFirst = now() - t0;
    filter->Modified();
    t0 = now();
    filter->Update();
    const double tUpdate = now() - t0;
    MeshType::Pointer mesh = filter->GetOutput();
    const double volumeBytes = static_cast<double>(n) * n * n * sizeof(float);
    const double linkSeconds = filter->MeasureHostToDeviceSeconds(static_cast<unsigned long long>(volumeBytes));

    typedef itk::VTKPolyDataWriter<MeshType> WriterType;
    WriterType::Pointer writer = WriterType::New();
    writer->SetInput(mesh);
    writer->SetFileName((prefix + "_mesh.vtk").c_str());
    t0 = now();
    writer->Update();
    const double tWriter = now() - t0;
    t0 = now();
    filter->WriteLastMeshAsVTKPolyData((prefix + "_flat.vtk").c_str());
    const double tFlat = now() - t0;
    const bool same = same_bytes(prefix + "_mesh.vtk", prefix + "_flat.vtk");

    std::cout << "{\"n\": " << n << ", \"points\": " << mesh->GetNumberOfPoints() << ", \"cells\": "
              << mesh->GetNumberOfCells() << ", \"first_update_s\": " << tFirst << ", \"update_s\": " << tUpdate
              << ", \"device_s\": " << filter->GetLastDeviceSeconds() << ", \"upload_extract_s\": "
              << filter->GetLastExtractSeconds() << ", \"download_s\": " << filter->GetLastDownloadSeconds()
              << ", \"mesh_fill_s\": " << filter->GetLastMeshFillSeconds() << ", \"itk_writer_s\": " << tWriter << ", \"flat_writer_s\": " << tFlat
              << ", \"volume_bytes\": " << volumeBytes << ", \"pinned_h2d_s\": " << linkSeconds
              << ", \"pinned_h2d_GBps\": " << volumeBytes / linkSeconds * 1e-9
              << ", \"same_bytes\": " << (same ? "true" : "false") << "}" << std::endl;
    return same ? 0 : 1;
    }
  catch (itk::ExceptionObject &e)
    {
    std::cerr << e << std::endl;
    return 1;
    }
}
