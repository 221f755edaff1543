// itkCuberilleImageToMeshFilter.h -- MI355X drop-in for the filter of midas-journal-740.
//
// Same class name, namespace, template parameters (and default), base class and public surface
// as /root/reference/Source/itkCuberilleImageToMeshFilter.h:110-236, so that the reference's
// Testing/CuberilleTest01.cxx and Source/examples.cxx compile against it unchanged.  Nothing of
// the reference's CPU algorithm lives here: GenerateData() (implementation file) marshals the
// image and the eight parameters into the C ABI of include/cuberille_hip.h, the HIP kernels in
// libcuberille_hip.so do the work on the GPU, and the flat buffers that come back are poured into
// the itk::Mesh with ITK's own ownership rules.  Builds against real ITK 3.x/4.x headers, or
// against the ITK-lite headers in itk_lite/ when ITK is not installed.
#ifndef __itkCuberilleImageToMeshFilter_h
#define __itkCuberilleImageToMeshFilter_h

// The reference fixes both at 0 (h:22-23), compiling the two alternative branches of ProjectVertexToIsoSurface out
// (txx:340-437).  A build of the reference with one of them switched on is matched by defining the same macro
// before this header is included: the choice travels as cuberille_params::projection_variant.
#ifndef USE_ADVANCED_PROJECTION
#define USE_ADVANCED_PROJECTION 0
#endif
#ifndef USE_LINESEARCH_PROJECTION
#define USE_LINESEARCH_PROJECTION 0
#endif
// ... and h:21 fixes this one at 0 (itk::GradientImageFilter); 1: itk::GradientRecursiveGaussianImageFilter with
// sigma = the largest spacing (txx:488-491), as cuberille_params::gradient_variant -- ITK's filter restated, parity unpinned
#ifndef USE_GRADIENT_RECURSIVE_GAUSSIAN
#define USE_GRADIENT_RECURSIVE_GAUSSIAN 0
#endif

#include "itkMacro.h"
#include "itkMesh.h"
#include "itkImageToMeshFilter.h"
#include "itkCellInterface.h"
#include "itkTriangleCell.h"
#include "itkQuadrilateralCell.h"
#include "itkDefaultStaticMeshTraits.h"
#include "itkConstShapedNeighborhoodIterator.h"
#include "itkLinearInterpolateImageFunction.h"
#include "itkGradientImageFilter.h"
#include "itkVectorLinearInterpolateImageFunction.h"
#include "itkNumericTraits.h"

struct cuberille_ctx;   // include/cuberille_hip.h

namespace itk
{

namespace cuberille_detail
{
// process-wide: do the filters constructed from now on set up their GPU context in the constructor (see SetEagerDeviceSetup)
inline bool &EagerDeviceSetup() { static bool on = true; return on; }
}


template <class TInputImage, class TOutputMesh,
          class TInterpolator = itk::LinearInterpolateImageFunction<TInputImage> >
class ITK_EXPORT CuberilleImageToMeshFilter : public ImageToMeshFilter<TInputImage, TOutputMesh>
{
public:
  typedef CuberilleImageToMeshFilter Self;
  typedef ImageToMeshFilter<TInputImage, TOutputMesh> Superclass;
  typedef SmartPointer<Self> Pointer;
  typedef SmartPointer<const Self> ConstPointer;

  itkNewMacro(Self);
  itkTypeMacro(CuberilleImageToMeshFilter, ImageToMeshFilter);

  // -- output side (reference h:127-145) --
  typedef TOutputMesh OutputMeshType;
  typedef typename OutputMeshType::Pointer OutputMeshPointer;
  typedef typename OutputMeshType::MeshTraits OutputMeshTraits;
  typedef typename OutputMeshType::PointType OutputPointType;
  typedef typename OutputMeshType::PointType PointType;
  typedef typename OutputMeshTraits::PixelType OutputPixelType;
  typedef typename OutputMeshType::CellTraits CellTraits;
  typedef typename OutputMeshType::PointsContainer PointsContainer;
  typedef typename OutputMeshType::PointsContainerPointer PointsContainerPointer;
  typedef typename OutputMeshType::CellsContainer CellsContainer;
  typedef typename OutputMeshType::CellsContainerPointer CellsContainerPointer;
  typedef typename OutputMeshType::PointIdentifier PointIdentifier;
  typedef typename OutputMeshType::CellIdentifier CellIdentifier;
  typedef CellInterface<OutputPixelType, CellTraits> CellInterfaceType;
  typedef TriangleCell<CellInterfaceType> TriangleCellType;
  typedef typename TriangleCellType::SelfAutoPointer TriangleAutoPointer;
  typedef typename TriangleCellType::CellAutoPointer TriangleCellAutoPointer;
  typedef QuadrilateralCell<CellInterfaceType> QuadrilateralCellType;
  typedef typename QuadrilateralCellType::SelfAutoPointer QuadrilateralAutoPointer;
  typedef typename QuadrilateralCellType::CellAutoPointer QuadrilateralCellAutoPointer;

  // -- input side (reference h:147-159) --
  typedef TInputImage InputImageType;
  typedef typename InputImageType::Pointer InputImagePointer;
  typedef typename InputImageType::ConstPointer InputImageConstPointer;
  typedef typename InputImageType::PixelType InputPixelType;
  typedef typename InputImageType::SizeType SizeType;
  typedef typename InputImageType::SpacingType SpacingType;
  typedef typename InputImageType::SpacingValueType SpacingValueType;
  typedef typename InputImageType::IndexType IndexType;
  typedef TInterpolator InterpolatorType;
  typedef typename InterpolatorType::Pointer InterpolatorPointer;
  typedef typename InterpolatorType::OutputType InterpolatorOutputType;

  // -- names the reference exposes for its CPU internals (h:162-173); kept so user code that
  //    mentions them still compiles, unused by the GPU path --
  typedef ConstShapedNeighborhoodIterator<InputImageType> InputImageIteratorType;
  typedef GradientImageFilter<InputImageType> GradientFilterType;
  typedef typename GradientFilterType::Pointer GradientFilterPointer;
  typedef typename GradientFilterType::OutputImageType GradientImageType;
  typedef typename GradientImageType::Pointer GradientImagePointer;
  typedef typename GradientFilterType::OutputPixelType GradientPixelType;
  typedef itk::VectorLinearInterpolateImageFunction<GradientImageType> GradientInterpolatorType;
  typedef typename GradientInterpolatorType::Pointer GradientInterpolatorPointer;

  /** Iso-surface value: pixels >= this value are inside (reference h:180-181, txx:139-141). */
  itkGetMacro(IsoSurfaceValue, InputPixelType);
  itkSetMacro(IsoSurfaceValue, InputPixelType);

  /** The image to polygonize (reference h:184, txx:53-56). */
  virtual void SetInput(const InputImageType *inputImage);

  /** Interpolator (reference h:187-188).  The kernels implement LinearInterpolateImageFunction<TInputImage,double>;
   *  with any other type the GPU still does the topology and the lattice points, and the walk of txx:439-474 runs on
   *  the host through the user's Evaluate() -- on the calling thread, as the reference does (txx:455), unless
   *  SetHostWalkThreads asks for more (the interpolator must then be safe to call from several threads at once). */
  itkGetObjectMacro(Interpolator, InterpolatorType);
  itkSetObjectMacro(Interpolator, InterpolatorType);

  /** Triangles (true, default) or quadrilaterals (reference h:193-195). */
  itkGetMacro(GenerateTriangleFaces, bool);
  itkSetMacro(GenerateTriangleFaces, bool);
  itkBooleanMacro(GenerateTriangleFaces);

  /** Project vertices onto the iso-surface (default true; reference h:199-201). */
  itkGetMacro(ProjectVerticesToIsoSurface, bool);
  itkSetMacro(ProjectVerticesToIsoSurface, bool);
  itkBooleanMacro(ProjectVerticesToIsoSurface);

  /** Projection knobs with the reference's clamps (h:209-228); defaults 0.5, max spacing / 4, 0.95, 50. */
  itkGetMacro(ProjectVertexSurfaceDistanceThreshold, double);
  itkSetClampMacro(ProjectVertexSurfaceDistanceThreshold, double, 0.0, NumericTraits<InputPixelType>::max());
  itkGetMacro(ProjectVertexStepLength, double);
  itkSetClampMacro(ProjectVertexStepLength, double, 0.0, 100000.0);
  itkGetMacro(ProjectVertexStepLengthRelaxationFactor, double);
  itkSetClampMacro(ProjectVertexStepLengthRelaxationFactor, double, 0.0, 1.0);
  itkGetMacro(ProjectVertexMaximumNumberOfSteps, unsigned int);
  itkSetMacro(ProjectVertexMaximumNumberOfSteps, unsigned int);

  /** Not in the reference: GPU device to run on (default 0), and the device time in seconds of
   *  the last GenerateData() (all kernels, excluding the PCIe copies and the itk::Mesh fill). */
  itkGetMacro(Device, int);
  itkSetMacro(Device, int);
  itkGetMacro(LastDeviceSeconds, double);
  /** Not in the reference: seconds the last GenerateData() spent pouring the flat buffers into the output
   *  mesh (all cells in one slab the mesh carries in its MetaDataDictionary, CellsAllocatedAsStaticArray, where the
   *  reference's txx:310-329 makes one heap object per face), and a way around that cost for callers that only want the file: the mesh of the last Update(), written as the legacy-ASCII VTK
   *  polydata itk::VTKPolyDataWriter would give for GetOutput(), straight from the device buffers. */
  itkGetMacro(LastMeshFillSeconds, double);
  /** Wall time of the last update's cuberille_extract_host call (upload overlapped with the sweep, then the rest
   * of the extraction) and of copying the flat mesh buffers back to the host. */
  itkGetMacro(LastExtractSeconds, double);
  itkGetMacro(LastDownloadSeconds, double);
  void WriteLastMeshAsVTKPolyData(const char *fileName, int threads = 0);
  /** Not in the reference: seconds the device of this filter takes to receive `bytes` from pinned host memory
   *  (cuberille_debug_h2d_seconds) -- what the upload inside Update() is measured against. */
  double MeasureHostToDeviceSeconds(unsigned long long bytes);
  /** Not in the reference: host threads of the walk taken for a TInterpolator the kernels do not implement.
   *  Default 1 -- the reference calls Evaluate() from one thread (txx:455) and an interpolator may keep mutable
   *  state; more only for interpolators whose Evaluate() const is thread-safe. */
  itkGetMacro(HostWalkThreads, unsigned int);
  itkSetClampMacro(HostWalkThreads, unsigned int, 1u, 256u);
  /** Not in the reference.  By default the constructor sets up the GPU context (runtime start, code objects, a toy
   *  extraction: 100-300 ms once per process) and SetInput sizes the device workspace for its image, so that the ONE cold
   *  Update() the reference's driver times (Testing/CuberilleTest01.cxx:158-160) is the extraction alone; the cost has not
   *  gone away -- profiles/r4_cold_update.log reports constructor and SetInput beside Update().  A process that builds
   *  filters it may never update, forks after constructing them, or picks the device later (SetDevice) turns that off for
   *  the filters it constructs afterwards: everything then happens inside the first Update(), as in round 3. */
  static void SetEagerDeviceSetup(bool on) { cuberille_detail::EagerDeviceSetup() = on; }
  static bool GetEagerDeviceSetup() { return cuberille_detail::EagerDeviceSetup(); }
  /** Not in the reference.  The context keeps the flat mesh in host memory of its own between updates (the second mesh of
   *  a process then copies at the link's rate); true gives that memory -- about 1.125 times the flat mesh -- back to the
   *  system as soon as the itk::Mesh is filled (default false). */
  itkGetMacro(ReleaseHostMeshAfterFill, bool);
  itkSetMacro(ReleaseHostMeshAfterFill, bool);
  itkBooleanMacro(ReleaseHostMeshAfterFill);

  /** Not in the reference -- its behaviour, on request.  The reference builds its gradient image and gradient interpolator
   *  only while the interpolator is null (txx:484) and never resets it: from the second Update() of a filter object on, the
   *  walk follows the gradient of the image the FIRST projecting Update() saw, whatever the input is by then.  This drop-in
   *  uses the current input's gradient; true makes the filter object behave like the reference's, update for update
   *  (cuberille_hold_gradient: the first input's gradient image stays on the device, 12 bytes per voxel).  Default false.
   *  Takes effect with the default LinearInterpolateImageFunction (the device walk); setting it back to false drops the
   *  held image. */
  itkGetMacro(ReproduceStaleGradient, bool);
  itkSetMacro(ReproduceStaleGradient, bool);
  itkBooleanMacro(ReproduceStaleGradient);

protected:
  CuberilleImageToMeshFilter();
  ~CuberilleImageToMeshFilter();
  void PrintSelf(std::ostream &os, Indent indent) const;

  void GenerateData();
  virtual void GenerateOutputInformation() {}   // as the reference (h:236)

private:
  CuberilleImageToMeshFilter(const Self &);   // not implemented
  void operator=(const Self &);               // not implemented

  InputPixelType m_IsoSurfaceValue;
  InterpolatorPointer m_Interpolator;
  SpacingValueType m_MaxSpacing;
  bool m_GenerateTriangleFaces;
  bool m_ProjectVerticesToIsoSurface;
  double m_ProjectVertexSurfaceDistanceThreshold;
  double m_ProjectVertexStepLength;
  double m_ProjectVertexStepLengthRelaxationFactor;
  unsigned int m_ProjectVertexMaximumNumberOfSteps;
  int m_Device;
  unsigned int m_HostWalkThreads;
  bool m_ReleaseHostMeshAfterFill;
  bool m_ReproduceStaleGradient;
  double m_LastDeviceSeconds;
  double m_LastMeshFillSeconds;
  double m_LastExtractSeconds;
  double m_LastDownloadSeconds;
  ::cuberille_ctx    *m_Context;
  int m_ContextDevice;                        // the device m_Context lives on (SetDevice may come after the constructor)
  bool AcquireContext(bool mustSucceed);      // create + cuberille_warm_up when there is none (or it sits on another device)
};

} // end namespace itk

#ifndef ITK_MANUAL_INSTANTIATION
#include "itkCuberilleImageToMeshFilter.txx"
#endif

#endif
