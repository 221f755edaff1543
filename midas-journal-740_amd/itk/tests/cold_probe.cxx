// Where one cold Update() goes (the only way the reference's driver calls the filter: one Update() per process,
// Testing/CuberilleTest01.cxx:158-160): the drop-in filter on a MetaImage, its three host-side intervals, and the
// same calls made directly on the C ABI with a clock around each.
//   usage: cold_probe <image.mha> <iso> [triangles=1] [project=1]
#include <chrono>
#include <cstdlib>
#include <iostream>

#include "itkImage.h"
#include "itkImageFileReader.h"
#include "itkMesh.h"
#include "itkCuberilleImageToMeshFilter.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char *argv[])
{
  if (argc < 3) { std::cerr << "usage: cold_probe <image.mha> <iso> [triangles] [project]" << std::endl; return 2; }
  typedef itk::Image<unsigned char, 3> ImageType;
  typedef itk::Mesh<unsigned char, 3> MeshType;
  typedef itk::CuberilleImageToMeshFilter<ImageType, MeshType> FilterType;
  try
    {
    itk::ImageFileReader<ImageType>::Pointer reader = itk::ImageFileReader<ImageType>::New();
    reader->SetFileName(argv[1]);
    reader->UpdateLargestPossibleRegion();
    ImageType::Pointer image = reader->GetOutput();
    image->DisconnectPipeline();
    double t0 = now();
    FilterType::Pointer filter = FilterType::New();
    const double tNew = now() - t0;
    t0 = now();
    filter->SetInput(image);
    const double tSetInput = now() - t0;
    filter->SetIsoSurfaceValue(static_cast<unsigned char>(std::atoi(argv[2])));
    filter->SetGenerateTriangleFaces(argc > 3 ? std::atoi(argv[3]) != 0 : true);
    filter->SetProjectVerticesToIsoSurface(argc > 4 ? std::atoi(argv[4]) != 0 : true);
    double tUpdate[3];
    for (int i = 0; i < 3; i++)
      {
      filter->Modified();
      t0 = now();
      filter->Update();
      tUpdate[i] = now() - t0;
      if (i == 0)
        std::cout << "first Update(): extract " << filter->GetLastExtractSeconds() * 1e3 << " ms (device " << filter->GetLastDeviceSeconds() * 1e3
                  << "), mesh to host " << filter->GetLastDownloadSeconds() * 1e3 << ", itk::Mesh fill " << filter->GetLastMeshFillSeconds() * 1e3 << std::endl;
      }
    std::cout << "New() " << tNew * 1e3 << " ms, SetInput " << tSetInput * 1e3 << " ms, Update() " << tUpdate[0] * 1e3 << " / " << tUpdate[1] * 1e3
              << " / " << tUpdate[2] * 1e3 << " ms; third: extract " << filter->GetLastExtractSeconds() * 1e3 << " (device "
              << filter->GetLastDeviceSeconds() * 1e3 << "), mesh to host " << filter->GetLastDownloadSeconds() * 1e3 << ", fill "
              << filter->GetLastMeshFillSeconds() * 1e3 << "; " << filter->GetOutput()->GetNumberOfPoints() << " points" << std::endl;
    }
  catch (itk::ExceptionObject &e)
    {
    std::cerr << e << std::endl;
    return 1;
    }
  return 0;
}
