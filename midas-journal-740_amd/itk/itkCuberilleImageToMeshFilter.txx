// Implementation of the MI355X drop-in filter: the host half of GenerateData().
// Reference counterpart: /root/reference/Source/itkCuberilleImageToMeshFilter.txx:29-216,500-520
// (constructor defaults, SetInput, parameter resolution, PrintSelf); everything from the
// neighbourhood sweep to the triangle split runs behind include/cuberille_hip.h.
#ifndef __itkCuberilleImageToMeshFilter_txx
#define __itkCuberilleImageToMeshFilter_txx

#include "itkCuberilleImageToMeshFilter.h"
#include "cuberille_hip.h"

#include <ctime>
#include <vector>

namespace itk
{

namespace cuberille_detail
{
template <class T> struct PixelCode { enum { Value = -1 }; };
template <> struct PixelCode<unsigned char>  { enum { Value = CUBERILLE_PIX_U8 }; };
template <> struct PixelCode<signed char>    { enum { Value = CUBERILLE_PIX_I8 }; };
template <> struct PixelCode<char>           { enum { Value = CUBERILLE_PIX_I8 }; };
template <> struct PixelCode<unsigned short> { enum { Value = CUBERILLE_PIX_U16 }; };
template <> struct PixelCode<short>          { enum { Value = CUBERILLE_PIX_I16 }; };
template <> struct PixelCode<unsigned int>   { enum { Value = CUBERILLE_PIX_U32 }; };
template <> struct PixelCode<int>            { enum { Value = CUBERILLE_PIX_I32 }; };
template <> struct PixelCode<float>          { enum { Value = CUBERILLE_PIX_F32 }; };
template <> struct PixelCode<double>         { enum { Value = CUBERILLE_PIX_F64 }; };

// the one interpolator the kernels implement (I5: linear, double coordinates)
template <class TInterpolator, class TImage> struct IsGpuInterpolator { enum { Value = 0 }; };
template <class TImage> struct IsGpuInterpolator<LinearInterpolateImageFunction<TImage, double>, TImage> { enum { Value = 1 }; };
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::CuberilleImageToMeshFilter()
{
  this->SetNumberOfRequiredInputs(1);
  m_IsoSurfaceValue = NumericTraits<InputPixelType>::One;
  m_MaxSpacing = 1.0;
  m_GenerateTriangleFaces = true;
  m_ProjectVerticesToIsoSurface = true;
  m_ProjectVertexSurfaceDistanceThreshold = 0.5;
  m_ProjectVertexStepLength = -1.0;            // resolved to max spacing / 4 at the first update
  m_ProjectVertexStepLengthRelaxationFactor = 0.95;
  m_ProjectVertexMaximumNumberOfSteps = 50;
  m_Device = 0;
  m_LastDeviceSeconds = 0.0;
  m_LastMeshFillSeconds = 0.0;
  m_Context = 0;
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::~CuberilleImageToMeshFilter()
{
  if (m_Context) cuberille_destroy(m_Context);
  m_Context = 0;
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
void CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::SetInput(const InputImageType *image)
{
  this->ProcessObject::SetNthInput(0, const_cast<InputImageType *>(image));
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
void CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::GenerateData()
{
  InputImageConstPointer image = Superclass::GetInput(0);
  typename OutputMeshType::Pointer mesh = Superclass::GetOutput();
  const unsigned int Dim = InputImageType::ImageDimension;
  if (Dim != 3) itkExceptionMacro(<< "the cuberille path is three-dimensional");
  if (cuberille_detail::PixelCode<InputPixelType>::Value < 0) itkExceptionMacro(<< "unsupported pixel type");
  if (m_ProjectVerticesToIsoSurface && !cuberille_detail::IsGpuInterpolator<TInterpolator, TInputImage>::Value)
    itkExceptionMacro(<< "only itk::LinearInterpolateImageFunction<TInputImage,double> is implemented by the MI355X path");

  // parameter resolution exactly where the reference does it: largest spacing, then the default
  // step length, which sticks to the filter object once resolved
  m_MaxSpacing = image->GetSpacing()[0];
  for (unsigned int i = 1; i < Dim; i++)
    if (image->GetSpacing()[i] > m_MaxSpacing) m_MaxSpacing = image->GetSpacing()[i];
  if (m_ProjectVertexStepLength < 0.0) m_ProjectVertexStepLength = m_MaxSpacing * 0.25;
  if (m_Interpolator.IsNull()) m_Interpolator = InterpolatorType::New();
  m_Interpolator->SetInputImage(image);

  cuberille_image_desc desc;
  desc.pixel_type = cuberille_detail::PixelCode<InputPixelType>::Value;
  const typename InputImageType::RegionType region = image->GetBufferedRegion();
  typename InputImageType::PointType firstPixel;
  image->TransformIndexToPhysicalPoint(region.GetIndex(), firstPixel);   // buffered index 0 of the C ABI
  for (unsigned int i = 0; i < 3; i++)
    {
    desc.dims[i] = static_cast<int64_t>(region.GetSize()[i]);
    desc.spacing[i] = image->GetSpacing()[i];
    desc.origin[i] = firstPixel[i];
    for (unsigned int j = 0; j < 3; j++) desc.direction[i * 3 + j] = image->GetDirection()[i][j];
    }

  cuberille_params prm;
  prm.iso_value = static_cast<double>(m_IsoSurfaceValue);
  prm.generate_triangles = m_GenerateTriangleFaces ? 1 : 0;
  prm.project_vertices = m_ProjectVerticesToIsoSurface ? 1 : 0;
  prm.distance_threshold = m_ProjectVertexSurfaceDistanceThreshold;
  prm.step_length = m_ProjectVertexStepLength;
  prm.relaxation = m_ProjectVertexStepLengthRelaxationFactor;
  prm.max_steps = m_ProjectVertexMaximumNumberOfSteps;
  prm.emulate_empty_slice_aliasing = 1;

  if (!m_Context && cuberille_create(&m_Context, m_Device) != CUBERILLE_OK)
    itkExceptionMacro(<< "cuberille_create: " << cuberille_last_error(0));
  cuberille_result res;
  if (cuberille_extract_host(m_Context, &desc, image->GetBufferPointer(), &prm, &res) != CUBERILLE_OK)
    itkExceptionMacro(<< "cuberille_extract_host: " << cuberille_last_error(m_Context));
  m_LastDeviceSeconds = 1e-3 * res.ms_total;

  std::vector<float> points(res.n_points * 3 + 1);
  std::vector<uint64_t> cells(res.n_cells * res.verts_per_cell + 1);
  if (cuberille_mesh_download(m_Context, &points[0], &cells[0]) != CUBERILLE_OK)
    itkExceptionMacro(<< "cuberille_mesh_download: " << cuberille_last_error(m_Context));

  const std::clock_t fillStart = std::clock();
  // pour the flat buffers into the mesh the way the reference does element by element: points by
  // value, one heap cell per face handed to the mesh, which owns it from then on
  if (res.n_points) mesh->GetPoints()->Reserve(static_cast<PointIdentifier>(res.n_points));
  PointType p;
  for (uint64_t i = 0; i < res.n_points; i++)
    {
    p[0] = points[3 * i]; p[1] = points[3 * i + 1]; p[2] = points[3 * i + 2];
    mesh->GetPoints()->InsertElement(static_cast<PointIdentifier>(i), p);
    }
  if (res.verts_per_cell == 3)
    {
    PointIdentifier ids[3];
    for (uint64_t c = 0; c < res.n_cells; c++)
      {
      for (int k = 0; k < 3; k++) ids[k] = static_cast<PointIdentifier>(cells[3 * c + k]);
      TriangleCellAutoPointer cell;
      cell.TakeOwnership(new TriangleCellType);
      cell->SetPointIds(ids);
      mesh->SetCell(static_cast<CellIdentifier>(c), cell);
      }
    }
  else
    {
    PointIdentifier ids[4];
    for (uint64_t c = 0; c < res.n_cells; c++)
      {
      for (int k = 0; k < 4; k++) ids[k] = static_cast<PointIdentifier>(cells[4 * c + k]);
      QuadrilateralCellAutoPointer cell;
      cell.TakeOwnership(new QuadrilateralCellType);
      cell->SetPointIds(ids);
      mesh->SetCell(static_cast<CellIdentifier>(c), cell);
      }
    }
  m_LastMeshFillSeconds = static_cast<double>(std::clock() - fillStart) / CLOCKS_PER_SEC;
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
void CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::WriteLastMeshAsVTKPolyData(const char *fileName, int threads)
{
  if (!m_Context) itkExceptionMacro(<< "WriteLastMeshAsVTKPolyData: no Update() has run on this filter");
  if (cuberille_mesh_write_vtk(m_Context, fileName, threads) != CUBERILLE_OK)
    itkExceptionMacro(<< "cuberille_mesh_write_vtk: " << cuberille_last_error(m_Context));
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
void CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::PrintSelf(std::ostream &os, Indent indent) const
{
  Superclass::PrintSelf(os, indent);
  os << indent << "IsoSurfaceValue: "
     << static_cast<typename NumericTraits<InputPixelType>::PrintType>(m_IsoSurfaceValue) << std::endl;
  os << indent << "GenerateTriangleFaces: " << m_GenerateTriangleFaces << std::endl;
  os << indent << "ProjectVerticesToIsoSurface: " << m_ProjectVerticesToIsoSurface << std::endl;
}

} // end namespace itk

#endif
