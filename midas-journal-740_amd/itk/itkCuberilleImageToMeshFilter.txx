// Implementation of the MI355X drop-in filter: the host half of GenerateData().
// Reference counterpart: /root/reference/Source/itkCuberilleImageToMeshFilter.txx:29-216,500-520
// (constructor defaults, SetInput, parameter resolution, PrintSelf); everything from the
// neighbourhood sweep to the triangle split runs behind include/cuberille_hip.h.
#ifndef __itkCuberilleImageToMeshFilter_txx
#define __itkCuberilleImageToMeshFilter_txx

#include "itkCuberilleImageToMeshFilter.h"
#include "itkMetaDataObject.h"
#include "cuberille_hip.h"

#include <cmath>
#include <cstdlib>
#if defined(__linux__)
#include <sys/mman.h>
#endif
#include <ctime>
#include <new>
#include <string>
#include <vector>
#if __cplusplus >= 201103L
#include <chrono>
#include <exception>
#include <thread>
#endif

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
// the 64-bit integer types: long / unsigned long where they are 64 bits wide (LP64), long long everywhere
template <> struct PixelCode<long>               { enum { Value = sizeof(long) == 8 ? CUBERILLE_PIX_I64 : CUBERILLE_PIX_I32 }; };
template <> struct PixelCode<unsigned long>      { enum { Value = sizeof(long) == 8 ? CUBERILLE_PIX_U64 : CUBERILLE_PIX_U32 }; };
template <> struct PixelCode<long long>          { enum { Value = CUBERILLE_PIX_I64 }; };
template <> struct PixelCode<unsigned long long> { enum { Value = CUBERILLE_PIX_U64 }; };

// cuberille_params::iso_value_int: the iso value of those types as an integer (a double cannot hold it past 2^53)
template <class T> struct IsoInt { static int64_t Get(T) { return 0; } };
template <> struct IsoInt<long>               { static int64_t Get(long v) { return static_cast<int64_t>(v); } };
template <> struct IsoInt<unsigned long>      { static int64_t Get(unsigned long v) { return static_cast<int64_t>(v); } };
template <> struct IsoInt<long long>          { static int64_t Get(long long v) { return static_cast<int64_t>(v); } };
template <> struct IsoInt<unsigned long long> { static int64_t Get(unsigned long long v) { return static_cast<int64_t>(v); } };

inline double WallSeconds()
{
#if __cplusplus >= 201103L
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
#else
  return static_cast<double>(std::clock()) / CLOCKS_PER_SEC;
#endif
}

// f(i0, i1) over the ranges of `nT` host threads (0: a few, for plain copies into independent elements).  An exception
// thrown inside a worker is carried to the calling thread and rethrown after the join.
template <class F> void ParallelRanges(uint64_t n, F f, unsigned int nT = 0)
{
#if __cplusplus >= 201103L
  if (nT == 0)
    {
    // (plain copies into fresh memory: what they wait for is the page faults of the destination, which more threads take
    //  in parallel -- 16 on a host with cores to spare)
    const unsigned int hw = std::thread::hardware_concurrency();
    nT = n < 65536 ? 1u : (hw >= 64 ? 16u : (hw >= 16 ? 8u : (hw >= 2 ? hw / 2 : 1u)));
    }
  if (nT > 1 && n >= nT)
    {
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> thrown(nT);
    for (unsigned int t = 0; t < nT; t++)
      {
      std::exception_ptr *slot = &thrown[t];
      const uint64_t i0 = n * t / nT, i1 = n * (t + 1) / nT;
      th.push_back(std::thread([f, i0, i1, slot]() { try { f(i0, i1); } catch (...) { *slot = std::current_exception(); } }));
      }
    for (unsigned int t = 0; t < nT; t++) th[t].join();
    for (unsigned int t = 0; t < nT; t++) if (thrown[t]) std::rethrow_exception(thrown[t]);
    return;
    }
#endif
  f(static_cast<uint64_t>(0), n);
}

// The storage of a mesh's cells when they are made in ONE allocation instead of one `new` per face (txx:309-329 makes
// 6.3 M heap objects for the bench's sphere).  itk::Mesh offers the mode for it -- CellsAllocatedAsStaticArray: the
// mesh keeps pointers to cells it does not free (itk::Mesh::ReleaseCellsMemory) -- and the storage is hung into the mesh's
// own MetaDataDictionary, reference counted: it is released when the mesh is destroyed, or when the next fill of the same
// mesh replaces it.  So the mesh still carries everything it needs, may be disconnected from the pipeline
// (Testing/CuberilleTest01.cxx:161-162) and may outlive the filter.  Only ITK API: the same code for ITK and ITK-lite.
template <class TCell> class CellSlab : public LightObject
{
public:
  typedef CellSlab Self;
  typedef LightObject Superclass;
  typedef SmartPointer<Self> Pointer;
  typedef SmartPointer<const Self> ConstPointer;
  itkNewMacro(Self);
  itkTypeMacro(CellSlab, LightObject);
  // raw storage for n cells; the filling threads construct them in place and say how many exist
  TCell *Allocate(uint64_t n)
  {
    Release();
    const size_t bytes = sizeof(TCell) * static_cast<size_t>(n ? n : 1);
#if defined(__linux__) && defined(MADV_HUGEPAGE)
    // a large slab on huge pages where the system offers them: 512 times fewer page faults under the filling threads
    if (bytes >= (static_cast<size_t>(8) << 20))
      {
      void *p = 0;
      if (posix_memalign(&p, static_cast<size_t>(2) << 20, bytes) == 0)
        {
        (void)madvise(p, bytes, MADV_HUGEPAGE);
        m_Cells = static_cast<TCell *>(p);
        m_Malloced = true;
        return m_Cells;
        }
      }
#endif
    m_Cells = static_cast<TCell *>(::operator new(bytes));
    m_Malloced = false;
    return m_Cells;
  }
  void SetNumberOfConstructedCells(uint64_t n) { m_Constructed = n; }
protected:
  CellSlab() : m_Cells(0), m_Constructed(0), m_Malloced(false) {}
  ~CellSlab() { Release(); }
private:
  CellSlab(const Self &);
  void operator=(const Self &);
  void Release()
  {
    for (uint64_t c = 0; c < m_Constructed; c++) m_Cells[c].~TCell();
    if (m_Malloced) std::free(m_Cells);
    else ::operator delete(m_Cells);
    m_Cells = 0;
    m_Constructed = 0;
  }
  TCell *m_Cells;
  uint64_t m_Constructed;
  bool m_Malloced;
};

// Does container C hand out a std::vector of E through CastToSTLContainer()?  itk::VectorContainer (DefaultStaticMeshTraits,
// the reference's mesh type) does; itk::MapContainer (DefaultDynamicMeshTraits) hands out a std::map: meshes of such traits are
// filled element by element, the way the reference does it (round-4 advisor finding: the bulk fill alone did not compile there).
template <class C, class E> struct CastsToVectorOf
{
  template <class U> static char Probe(U *, typename std::enable_if<std::is_same<decltype(std::declval<U &>().CastToSTLContainer()),
                                                                                   std::vector<E> &>::value>::type * = 0);
  template <class U> static long Probe(...);
  static const bool Value = sizeof(Probe<C>(static_cast<C *>(0))) == sizeof(char);
};
template <bool> struct FillTag {};

// The reference's own loop (txx:309-329): one heap object per cell, handed to the mesh, which owns and later deletes it.
template <class TMesh, class TCell, unsigned int K>
void FillCells(TMesh *mesh, const uint64_t *ids, uint64_t nCells, FillTag<false>)
{
  typedef typename TMesh::PointIdentifier PointIdentifier;
  typedef typename TMesh::CellAutoPointer CellAutoPointer;
  for (uint64_t c = 0; c < nCells; c++)
    {
    PointIdentifier v[K];
    for (unsigned int k = 0; k < K; k++) v[k] = static_cast<PointIdentifier>(ids[K * c + k]);
    CellAutoPointer cell;
    TCell *one = new TCell;
    one->SetPointIds(v);
    cell.TakeOwnership(one);
    mesh->SetCell(static_cast<typename TMesh::CellIdentifier>(c), cell);
    }
}

template <class TMesh>
void FillPoints(TMesh *mesh, const float *src, uint64_t nPoints, FillTag<false>)
{
  typename TMesh::PointType p;
  for (uint64_t i = 0; i < nPoints; i++)
    {
    p[0] = src[3 * i]; p[1] = src[3 * i + 1]; p[2] = src[3 * i + 2];
    mesh->SetPoint(static_cast<typename TMesh::PointIdentifier>(i), p);      // txx:275
    }
}

// points by value, straight into the vector's elements
template <class TMesh>
void FillPoints(TMesh *mesh, const float *src, uint64_t nPoints, FillTag<true>)
{
  typedef typename TMesh::PointType PointType;
  std::vector<PointType> &pc = mesh->GetPoints()->CastToSTLContainer();
  pc.resize(static_cast<size_t>(nPoints));
  struct Fill
    {
    PointType *dst; const float *src;
    void operator()(uint64_t i0, uint64_t i1) const
      { for (uint64_t i = i0; i < i1; i++) { dst[i][0] = src[3 * i]; dst[i][1] = src[3 * i + 1]; dst[i][2] = src[3 * i + 2]; } }
    } fp = {nPoints ? &pc[0] : 0, src};
  ParallelRanges(nPoints, fp);
}

// Bulk form of the loop at txx:309-329: all cells of the mesh constructed in one slab, the cell container filled with
// pointers into it by a few threads.
template <class TMesh, class TCell, unsigned int K>
void FillCells(TMesh *mesh, const uint64_t *ids, uint64_t nCells, FillTag<true>)
{
  typedef typename TMesh::CellType BaseCellType;
  typedef typename TMesh::PointIdentifier PointIdentifier;
  typedef typename TMesh::CellsContainer CellsContainer;
  typename CellSlab<TCell>::Pointer holder = CellSlab<TCell>::New();
  TCell *slab = holder->Allocate(nCells);
  if (mesh->GetCells() == 0)
    {
    typename CellsContainer::Pointer fresh = CellsContainer::New();
    mesh->SetCells(fresh);
    }
  mesh->SetCellsAllocationMethod(TMesh::CellsAllocatedAsStaticArray);
  std::vector<BaseCellType *> &container = mesh->GetCells()->CastToSTLContainer();
  container.resize(static_cast<size_t>(nCells));
  BaseCellType **slots = nCells ? &container[0] : 0;
  struct Fill
    {
    TCell *slab; BaseCellType **slots; const uint64_t *ids;
    void operator()(uint64_t c0, uint64_t c1) const
      {
      PointIdentifier v[K];
      for (uint64_t c = c0; c < c1; c++)
        {
        TCell *cell = new (slab + c) TCell;
        for (unsigned int k = 0; k < K; k++) v[k] = static_cast<PointIdentifier>(ids[K * c + k]);
        cell->SetPointIds(v);
        slots[c] = cell;
        }
      }
    } fill = {slab, slots, ids};
  ParallelRanges(nCells, fill);
  holder->SetNumberOfConstructedCells(nCells);
  EncapsulateMetaData<typename CellSlab<TCell>::Pointer>(mesh->GetMetaDataDictionary(), std::string("CuberilleCellSlab"), holder);
}

// the image as the C ABI takes it: itk::Image::{GetBufferedRegion, GetSpacing, GetOrigin, GetDirection} (txx:71-99,266-270),
// the region's start index handed over as it is -- the library applies ITK's index <-> point transforms to buffer position +
// start index, like ITK itself (a cropped image keeps the index it was cut at); false when nothing is buffered yet
template <class TImage> bool DescribeImage(const TImage *image, cuberille_image_desc &desc)
{
  desc.pixel_type = PixelCode<typename TImage::PixelType>::Value;
  if (!image || TImage::ImageDimension != 3) return false;
  const typename TImage::RegionType region = image->GetBufferedRegion();
  bool any = true;
  for (unsigned int i = 0; i < 3; i++)
    {
    desc.dims[i] = static_cast<int64_t>(region.GetSize()[i]);
    desc.spacing[i] = image->GetSpacing()[i];
    desc.origin[i] = image->GetOrigin()[i];
    desc.index_start[i] = static_cast<int64_t>(region.GetIndex()[i]);
    for (unsigned int j = 0; j < 3; j++) desc.direction[i * 3 + j] = image->GetDirection()[i][j];
    if (desc.dims[i] < 1) any = false;
    }
  return any;
}

// ---------------------------------------------------------------------------------------------------------------
// Host projection for interpolators the kernels do not implement (TInterpolator other than
// LinearInterpolateImageFunction<TInputImage,double>).  The GPU still does the whole topology -- inside test, faces,
// vertex ids and order, lattice points -- and hands back unprojected quads; this walks every vertex as txx:439-474
// does, calling the USER'S interpolator for the value (txx:455), and splits the quads afterwards (txx:286-321).
// The gradient is what txx:478-498 sets up: GradientImageFilter (central differences in float, image spacing,
// direction applied) read through a vector linear interpolator; restated here on the image buffer so that no
// gradient image is materialised.
// ---------------------------------------------------------------------------------------------------------------
template <class TImage> class HostGradient
{
public:
  typedef typename TImage::PixelType PixelType;
  explicit HostGradient(const TImage *image)
  {
    // (like the library since ABI 13: ITK's own origin and the region's start index -- the continuous index of a point is an INDEX,
    //  a buffer position is that minus the start)
    const typename TImage::RegionType region = image->GetBufferedRegion();
    m_Buffer = image->GetBufferPointer();
    double i2p[9];
    for (int r = 0; r < 3; r++)
      {
      m_N[r] = static_cast<long>(region.GetSize()[r]);
      m_Start[r] = static_cast<long>(region.GetIndex()[r]);
      m_Origin[r] = image->GetOrigin()[r];
      m_Coef[r] = static_cast<float>(0.5 * (1.0 / image->GetSpacing()[r]));
      for (int c = 0; c < 3; c++)
        {
        m_Dir[r * 3 + c] = image->GetDirection()[r][c];
        i2p[r * 3 + c] = m_Dir[r * 3 + c] * image->GetSpacing()[c];
        }
      }
    // PhysicalPointToIndex = inverse of Direction * diag(spacing), by cofactors
    const double c00 = i2p[4] * i2p[8] - i2p[5] * i2p[7], c01 = i2p[5] * i2p[6] - i2p[3] * i2p[8];
    const double c02 = i2p[3] * i2p[7] - i2p[4] * i2p[6];
    const double det = i2p[0] * c00 + i2p[1] * c01 + i2p[2] * c02;
    m_P2I[0] = c00 / det; m_P2I[1] = (i2p[2] * i2p[7] - i2p[1] * i2p[8]) / det; m_P2I[2] = (i2p[1] * i2p[5] - i2p[2] * i2p[4]) / det;
    m_P2I[3] = c01 / det; m_P2I[4] = (i2p[0] * i2p[8] - i2p[2] * i2p[6]) / det; m_P2I[5] = (i2p[2] * i2p[3] - i2p[0] * i2p[5]) / det;
    m_P2I[6] = c02 / det; m_P2I[7] = (i2p[1] * i2p[6] - i2p[0] * i2p[7]) / det; m_P2I[8] = (i2p[0] * i2p[4] - i2p[1] * i2p[3]) / det;
  }

  // linearly interpolated gradient at a physical point, as a CovariantVector<float,3> would hold it
  void Evaluate(const double p[3], float g[3]) const
  {
    double cv[3], d[3];
    long lo[3], hi[3];
    for (int k = 0; k < 3; k++) cv[k] = p[k] - m_Origin[k];
    for (int r = 0; r < 3; r++)
      {
      double ci = 0.0;
      for (int k = 0; k < 3; k++) ci += m_P2I[r * 3 + k] * cv[k];
      const double b = std::floor(ci);
      d[r] = ci - b;
      // the neighbour indices are clamped into the image (a NaN coordinate lands on index 0)
      const double bl = b - static_cast<double>(m_Start[r]);                 // (exact: both are integers well inside 2^53)
      long bi = (bl >= -1.0) ? ((bl <= static_cast<double>(m_N[r])) ? static_cast<long>(bl) : m_N[r]) : -1;
      if (!(b == b)) bi = 0;
      lo[r] = bi < 0 ? 0 : (bi > m_N[r] - 1 ? m_N[r] - 1 : bi);
      hi[r] = bi + 1 < 0 ? 0 : (bi + 1 > m_N[r] - 1 ? m_N[r] - 1 : bi + 1);
      }
    double acc[3] = {0.0, 0.0, 0.0}, total = 0.0;
    for (unsigned int counter = 0; counter < 8; counter++)
      {
      double overlap = 1.0;
      long idx[3];
      for (int k = 0; k < 3; k++)
        {
        if (counter & (1u << k)) { idx[k] = hi[k]; overlap *= d[k]; }
        else { idx[k] = lo[k]; overlap *= 1.0 - d[k]; }
        }
      if (overlap != 0.0 && total != 1.0)        // "if (overlap)" and "break once the weights add up to 1" of ITK's loop
        {
        float site[3];
        SiteGradient(idx, site);
        for (int k = 0; k < 3; k++) acc[k] += overlap * static_cast<double>(site[k]);
        total += overlap;
        }
      }
    for (int k = 0; k < 3; k++) g[k] = static_cast<float>(acc[k]);
  }

private:
  float Pixel(long x, long y, long z) const
  {
    x = x < 0 ? 0 : (x > m_N[0] - 1 ? m_N[0] - 1 : x);
    y = y < 0 ? 0 : (y > m_N[1] - 1 ? m_N[1] - 1 : y);
    z = z < 0 ? 0 : (z > m_N[2] - 1 ? m_N[2] - 1 : z);
    return static_cast<float>(m_Buffer[(static_cast<size_t>(z) * m_N[1] + y) * m_N[0] + x]);
  }
  // GradientImageFilter at one pixel: per axis the 3-tap inner product (-c, 0, +c) accumulated in float, then the
  // direction matrix (float accumulator, double products)
  void SiteGradient(const long idx[3], float out[3]) const
  {
    float local[3];
    const float f0 = Pixel(idx[0], idx[1], idx[2]);
    for (int a = 0; a < 3; a++)
      {
      const float fm = Pixel(idx[0] - (a == 0), idx[1] - (a == 1), idx[2] - (a == 2));
      const float fp = Pixel(idx[0] + (a == 0), idx[1] + (a == 1), idx[2] + (a == 2));
      float sum = 0.0f;
      sum += (-m_Coef[a]) * fm;
      sum += 0.0f * f0;
      sum += m_Coef[a] * fp;
      local[a] = sum;
      }
    for (int r = 0; r < 3; r++)
      {
      float sum = 0.0f;
      for (int c = 0; c < 3; c++) sum = static_cast<float>(static_cast<double>(sum) + m_Dir[r * 3 + c] * static_cast<double>(local[c]));
      out[r] = sum;
      }
  }
  const PixelType *m_Buffer;
  long m_N[3], m_Start[3];
  double m_Origin[3], m_Dir[9], m_P2I[9];
  float m_Coef[3];
};

// txx:439-474 for the vertices [i0, i1): points are float[3] each, moved in place
template <class TImage, class TInterpolator> struct HostWalk
{
  const HostGradient<TImage> *gradient;
  const TInterpolator *interpolator;
  float *points;
  double iso, threshold, stepLength, relaxation;
  unsigned int maxSteps;
  void operator()(uint64_t i0, uint64_t i1) const
  {
    typename TInterpolator::PointType q;
    for (uint64_t i = i0; i < i1; i++)
      {
      float *v = points + 3 * i;
      double step = stepLength;
      unsigned int numberOfSteps = 0;
      bool done = false;
      while (!done)
        {
        const double p[3] = {static_cast<double>(v[0]), static_cast<double>(v[1]), static_cast<double>(v[2])};
        float normal[3];
        gradient->Evaluate(p, normal);
        double sq = 0.0;
        for (int k = 0; k < 3; k++) sq += static_cast<double>(normal[k]) * static_cast<double>(normal[k]);
        const double norm = std::sqrt(sq);
        for (int k = 0; k < 3; k++) normal[k] = static_cast<float>(static_cast<double>(normal[k]) / norm);   // no zero guard (txx:452)
        for (int k = 0; k < 3; k++) q[k] = p[k];
        const double value = static_cast<double>(interpolator->Evaluate(q));
        const double diff = value - iso;
        done = (diff < 0 ? -diff : diff) < threshold;
        if (!done)
          {
          const double sign = (value < iso) ? +1.0 : -1.0;
          for (int k = 0; k < 3; k++)
            v[k] = static_cast<float>(static_cast<double>(v[k]) + (static_cast<double>(normal[k]) * sign * step));
          step *= relaxation;
          done = numberOfSteps++ > maxSteps;
          }
        }
      }
  }
};

// txx:286-321 on flat buffers: quads (4 ids) -> two triangles along the shorter diagonal, ties to the first form
inline void SplitQuads(const float *points, const uint64_t *quads, uint64_t nQuads, uint64_t *triangles)
{
  for (uint64_t c = 0; c < nQuads; c++)
    {
    const uint64_t *f = quads + 4 * c;
    double d02 = 0.0, d13 = 0.0;
    for (int k = 0; k < 3; k++)
      {
      const double a = static_cast<double>(points[3 * f[2] + k]) - static_cast<double>(points[3 * f[0] + k]);
      const double b = static_cast<double>(points[3 * f[3] + k]) - static_cast<double>(points[3 * f[1] + k]);
      d02 += a * a;
      d13 += b * b;
      }
    uint64_t *t = triangles + 6 * c;
    if (d02 >= d13) { t[0] = f[0]; t[1] = f[1]; t[2] = f[3]; t[3] = f[1]; t[4] = f[2]; t[5] = f[3]; }
    else { t[0] = f[0]; t[1] = f[1]; t[2] = f[2]; t[3] = f[0]; t[4] = f[2]; t[5] = f[3]; }
    }
}

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
  m_HostWalkThreads = 1;
  m_ReleaseHostMeshAfterFill = false;
  m_ReproduceStaleGradient = false;
  m_LastDeviceSeconds = 0.0;
  m_LastMeshFillSeconds = 0.0;
  m_LastExtractSeconds = 0.0;
  m_LastDownloadSeconds = 0.0;
  m_Context = 0;
  m_ContextDevice = -1;
  // The reference's driver constructs the filter, sets its input and only then starts its clock around ONE Update() in a
  // fresh process (Testing/CuberilleTest01.cxx:144-160): the GPU context, the code objects and the runtime's queues are
  // therefore set up here, not inside that Update().  Without a usable device this is silent: GenerateData() tries
  // again and reports.  (SetEagerDeviceSetup(false): not here, inside the first Update().)
  if (cuberille_detail::EagerDeviceSetup()) this->AcquireContext(false);
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::~CuberilleImageToMeshFilter()
{
  if (m_Context) cuberille_destroy(m_Context);
  m_Context = 0;
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
bool CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::AcquireContext(bool mustSucceed)
{
  if (m_Context && m_ContextDevice != m_Device)      // SetDevice after the constructor
    {
    cuberille_destroy(m_Context);
    m_Context = 0;
    }
  if (!m_Context)
    {
    if (cuberille_create(&m_Context, m_Device) != CUBERILLE_OK)
      {
      m_Context = 0;
      if (mustSucceed) itkExceptionMacro(<< "cuberille_create: " << cuberille_last_error(0));
      return false;
      }
    m_ContextDevice = m_Device;
    (void)cuberille_warm_up(m_Context, 0, 0);
    }
  return true;
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
void CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::SetInput(const InputImageType *image)
{
  this->ProcessObject::SetNthInput(0, const_cast<InputImageType *>(image));
  // an image that is buffered already (the driver reads it first, test:113-117): size the device workspace for it now
  cuberille_image_desc desc;
  if (cuberille_detail::EagerDeviceSetup() && cuberille_detail::PixelCode<InputPixelType>::Value >= 0 &&
      cuberille_detail::DescribeImage(image, desc) && this->AcquireContext(false))
    (void)cuberille_warm_up(m_Context, &desc, 0);
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
void CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::GenerateData()
{
  InputImageConstPointer image = Superclass::GetInput(0);
  typename OutputMeshType::Pointer mesh = Superclass::GetOutput();
  const unsigned int Dim = InputImageType::ImageDimension;
  if (Dim != 3) itkExceptionMacro(<< "the cuberille path is three-dimensional");
  if (cuberille_detail::PixelCode<InputPixelType>::Value < 0) itkExceptionMacro(<< "unsupported pixel type");
  // any other TInterpolator than the linear one the kernels implement: topology and lattice points on the GPU,
  // the walk on the host through the user's interpolator (see HostWalk above)
  const bool hostWalk = m_ProjectVerticesToIsoSurface && !cuberille_detail::IsGpuInterpolator<TInterpolator, TInputImage>::Value;

  // parameter resolution exactly where the reference does it: largest spacing, then the default
  // step length, which sticks to the filter object once resolved
  m_MaxSpacing = image->GetSpacing()[0];
  for (unsigned int i = 1; i < Dim; i++)
    if (image->GetSpacing()[i] > m_MaxSpacing) m_MaxSpacing = image->GetSpacing()[i];
  if (m_ProjectVertexStepLength < 0.0) m_ProjectVertexStepLength = m_MaxSpacing * 0.25;
  if (m_Interpolator.IsNull()) m_Interpolator = InterpolatorType::New();
  m_Interpolator->SetInputImage(image);

  cuberille_image_desc desc;
  (void)cuberille_detail::DescribeImage(image.GetPointer(), desc);      // (an empty region is the library's to refuse)

  cuberille_params prm;
  prm.iso_value = static_cast<double>(m_IsoSurfaceValue);
  prm.generate_triangles = (m_GenerateTriangleFaces && !hostWalk) ? 1 : 0;   // the split needs the projected points
  prm.project_vertices = (m_ProjectVerticesToIsoSurface && !hostWalk) ? 1 : 0;
  prm.distance_threshold = m_ProjectVertexSurfaceDistanceThreshold;
  prm.step_length = m_ProjectVertexStepLength;
  prm.relaxation = m_ProjectVertexStepLengthRelaxationFactor;
  prm.max_steps = m_ProjectVertexMaximumNumberOfSteps;
  prm.emulate_empty_slice_aliasing = 1;
  // same precedence as the reference's #if / #elif (txx:340,398)
  prm.projection_variant = USE_ADVANCED_PROJECTION ? CUBERILLE_PROJECT_ADVANCED
                         : (USE_LINESEARCH_PROJECTION ? CUBERILLE_PROJECT_LINESEARCH : CUBERILLE_PROJECT_DEFAULT);
  prm.gradient_variant = USE_GRADIENT_RECURSIVE_GAUSSIAN ? CUBERILLE_GRADIENT_RECURSIVE_GAUSSIAN : CUBERILLE_GRADIENT_CENTRAL;
  prm.iso_value_int = cuberille_detail::IsoInt<InputPixelType>::Get(m_IsoSurfaceValue);
  if (hostWalk && m_ProjectVerticesToIsoSurface &&
      (prm.projection_variant != CUBERILLE_PROJECT_DEFAULT || prm.gradient_variant != CUBERILLE_GRADIENT_CENTRAL))
    itkExceptionMacro(<< "USE_ADVANCED_PROJECTION / USE_LINESEARCH_PROJECTION / USE_GRADIENT_RECURSIVE_GAUSSIAN are only "
                         "offered with the default LinearInterpolateImageFunction");

  this->AcquireContext(true);
  if (cuberille_hold_gradient(m_Context, (m_ReproduceStaleGradient && !hostWalk) ? 1 : 0) != CUBERILLE_OK)
    itkExceptionMacro(<< "cuberille_hold_gradient: " << cuberille_last_error(m_Context));
  cuberille_result res;
  const double extractStart = cuberille_detail::WallSeconds();
  if (cuberille_extract_host(m_Context, &desc, image->GetBufferPointer(), &prm, &res) != CUBERILLE_OK)
    itkExceptionMacro(<< "cuberille_extract_host: " << cuberille_last_error(m_Context));
  m_LastDeviceSeconds = 1e-3 * res.ms_total;

  m_LastExtractSeconds = cuberille_detail::WallSeconds() - extractStart;

  // the flat buffers, in host memory the context owns and keeps (cuberille_mesh_host): no allocation of ours, no page
  // fault per 4 KiB of a fresh destination; valid until the next extraction on the context
  const double downloadStart = cuberille_detail::WallSeconds();
  float *points = 0;
  uint64_t *cells = 0;
  if (cuberille_mesh_host(m_Context, &points, &cells) != CUBERILLE_OK)
    itkExceptionMacro(<< "cuberille_mesh_host: " << cuberille_last_error(m_Context));
  m_LastDownloadSeconds = cuberille_detail::WallSeconds() - downloadStart;
  struct FreeOnExit { void *p; ~FreeOnExit() { std::free(p); } } ownCells = {0};   // the host walk's triangles, when it makes them

  if (hostWalk)
    {
    cuberille_detail::HostGradient<InputImageType> gradient(image.GetPointer());
    cuberille_detail::HostWalk<InputImageType, TInterpolator> walk =
      {&gradient, m_Interpolator.GetPointer(), points, static_cast<double>(m_IsoSurfaceValue),
       m_ProjectVertexSurfaceDistanceThreshold, m_ProjectVertexStepLength, m_ProjectVertexStepLengthRelaxationFactor,
       m_ProjectVertexMaximumNumberOfSteps};
    // on the calling thread unless the user vouched for the interpolator (SetHostWalkThreads); whatever Evaluate()
    // throws leaves through Update() like any other exception of the pipeline
    cuberille_detail::ParallelRanges(res.n_points, walk, m_HostWalkThreads);
    if (m_GenerateTriangleFaces)
      {
      ownCells.p = std::malloc(sizeof(uint64_t) * (res.n_cells * 6 + 1));
      if (!ownCells.p) itkExceptionMacro(<< "out of host memory for the mesh buffers");
      cuberille_detail::SplitQuads(points, cells, res.n_cells, static_cast<uint64_t *>(ownCells.p));
      cells = static_cast<uint64_t *>(ownCells.p);
      res.n_cells *= 2;
      res.verts_per_cell = 3;
      }
    }

  const double fillStart = cuberille_detail::WallSeconds();
  // Meshes whose containers are vectors (the reference's mesh type and every DefaultStaticMeshTraits mesh): points straight
  // into the vector's elements, all cells in one slab that lives and dies with the mesh (CellSlab above); any other
  // container -- DefaultDynamicMeshTraits' maps -- element by element through SetPoint / SetCell like the reference
  typedef typename OutputMeshType::PointsContainer PointsContainerType;
  typedef typename OutputMeshType::CellsContainer CellsContainerType;
  typedef typename OutputMeshType::CellType BaseCellType;
  cuberille_detail::FillPoints(mesh.GetPointer(), points, res.n_points,
                               cuberille_detail::FillTag<cuberille_detail::CastsToVectorOf<PointsContainerType, PointType>::Value>());
  typedef cuberille_detail::FillTag<cuberille_detail::CastsToVectorOf<CellsContainerType, BaseCellType *>::Value> CellFillTag;
  if (res.verts_per_cell == 3) cuberille_detail::FillCells<OutputMeshType, TriangleCellType, 3>(mesh.GetPointer(), cells, res.n_cells, CellFillTag());
  else cuberille_detail::FillCells<OutputMeshType, QuadrilateralCellType, 4>(mesh.GetPointer(), cells, res.n_cells, CellFillTag());
  m_LastMeshFillSeconds = cuberille_detail::WallSeconds() - fillStart;
  if (m_ReleaseHostMeshAfterFill) (void)cuberille_release_host_mesh(m_Context);
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
void CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::WriteLastMeshAsVTKPolyData(const char *fileName, int threads)
{
  if (!m_Context) itkExceptionMacro(<< "WriteLastMeshAsVTKPolyData: no Update() has run on this filter");
  if (cuberille_mesh_write_vtk(m_Context, fileName, threads) != CUBERILLE_OK)
    itkExceptionMacro(<< "cuberille_mesh_write_vtk: " << cuberille_last_error(m_Context));
}

template <class TInputImage, class TOutputMesh, class TInterpolator>
double CuberilleImageToMeshFilter<TInputImage, TOutputMesh, TInterpolator>::MeasureHostToDeviceSeconds(unsigned long long bytes)
{
  this->AcquireContext(true);
  double s = 0.0;
  if (cuberille_debug_h2d_seconds(m_Context, static_cast<size_t>(bytes), &s) != CUBERILLE_OK)
    itkExceptionMacro(<< "cuberille_debug_h2d_seconds: " << cuberille_last_error(m_Context));
  return s;
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
