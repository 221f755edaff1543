// C-ABI host layer of the MI355X cuberille extractor (include/cuberille_hip.h).
//
// Plays the role of the body of itk::CuberilleImageToMeshFilter::GenerateData()
// (/root/reference/Source/itkCuberilleImageToMeshFilter.txx:59-216): resolves the
// parameters the way txx:75-95 does, then drives the HIP kernels of
// cuberille_kernels.hip on one stream.  No CPU fallback exists: without a gfx950
// device every computing entry point fails with CUBERILLE_ERR_NO_DEVICE.

#include "../../include/cuberille_hip.h"
#include "cuberille_internal.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <vector>

using namespace cuberille;

namespace {

// text of the last failed cuberille_create on this thread (contexts are independent; so are their creators)
thread_local std::string g_create_error;

// Failure drill (cuberille_debug_set_option "fail_alloc_at" = n): the n-th device allocation this THREAD makes from now on
// reports out-of-memory without touching the device; -1 = off.  Lets the tests walk every allocation-failure path.
thread_local long long g_fail_alloc_countdown = -1;

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (g_fail_alloc_countdown >= 0 && g_fail_alloc_countdown-- == 0) return hipErrorOutOfMemory;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    // grow with head-room so that repeated calls on similar volumes do not re-allocate
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  // When the buffer has to grow anyway, grow it to `cover` -- what the NEXT extraction on the context will ask for when it is
  // launched from this one's sizes (cuberille_step_begin: + 25 %) -- so that the second extraction of a series does not
  // free and allocate the mesh buffers again (176 ms of one 14 ms extraction at 2048^3); back to `bytes` if that is too much.
  hipError_t reserve_covering(size_t bytes, size_t cover) {
    if (bytes <= cap) return hipSuccess;
    if (cover > bytes) {
      if (g_fail_alloc_countdown < 0) {          // (the failure drill counts one allocation per buffer: keep it so)
        const size_t had = cap;
        cap = 0;
        if (p) { (void)hipFree(p); p = nullptr; }
        if (hipMalloc(&p, cover + 256) == hipSuccess) { cap = cover + 256; return hipSuccess; }
        (void)hipGetLastError();
        p = nullptr;
        (void)had;
      }
    }
    return reserve(bytes);
  }
};

// Host memory of the context's own (cuberille_mesh_host): anonymous pages, 2 MiB aligned and advised as huge pages where
// the system has them; kept and re-used across extractions, grown with head-room.
struct HostBuf {
  void *p = nullptr, *base = nullptr;
  size_t cap = 0, mapped = 0;
  bool reserve(size_t bytes) {
    if (bytes <= cap) return true;
    release();
    const size_t huge = 2u << 20;
    size_t want = bytes + bytes / 8;
    want = (want + huge - 1) & ~(huge - 1);
    void *m = mmap(nullptr, want + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return false;
    base = m;
    mapped = want + huge;
    p = (void *)(((uintptr_t)m + huge - 1) & ~(uintptr_t)(huge - 1));
    cap = want;
#ifdef MADV_HUGEPAGE
    (void)madvise(p, cap, MADV_HUGEPAGE);
#endif
    return true;
  }
  void release() {
    if (base) (void)munmap(base, mapped);
    p = base = nullptr;
    cap = mapped = 0;
  }
};

}  // namespace

struct cuberille_ctx {
  int device = 0;
  hipStream_t own = nullptr, stream = nullptr;
  std::string err;
  DevBuf voxOwn, bits, flatBits, occ, prefix, segPre, blockTot, blockBase, points, cells, cmap, headV, headQ, vqueue, escList;
  DevBuf gradImg, rgA, rgB, rgScratch;   // gradient_variant 1: the gradient image and what its passes go through
  DevBuf heldGrad;                       // cuberille_hold_gradient: the float gradient image of the first projecting extraction
  HeldGradient held{};                   // ... with its geometry (img == null: none yet); what every later walk follows
  bool holdGradient = false;             // ... asked for
  HostBuf hostPoints, hostCells;         // cuberille_mesh_host: the last mesh in host memory of the context's own
  bool hostMeshValid = false;            // ... holds the mesh of the last emit
  Totals *hostTotals = nullptr;          // pinned
  uint32_t *hostOcc = nullptr;           // pinned mirror of the per-slice occupancy of the last slab count
  size_t hostOccCap = 0;
  hipEvent_t ev[8] = {};
  Tuning tune;                           // development switches (cuberille_debug_set_option)
  // overlapped ingestion (cuberille_extract_host): pinned staging ring and a copy stream
  hipStream_t copyStream = nullptr;
  void *stage[2] = {nullptr, nullptr};
  size_t stageBytes = 0;
  hipEvent_t stageFree[2] = {}, chunkIn[2] = {};
  bool aliasBelowBuffer = false;         // soft condition of the last slab count (cuberille_slab_info)
  bool aliasMustResolve = false;         // ... and it is certain: the source slice lies in this slab's own halo
  int aliasZ = -1;                       // local slice whose Q1 source is unresolved (the first occupied counted slice), -1
  bool slabMode = false;                 // the last count was given a slab
  bool thinHalo = false;                 // ... with CUBERILLE_SLAB_THIN_HALO: walks that leave the buffer are put aside
  bool pointsStartedEarly = false;       // cuberille_emit_points ran ahead of cuberille_emit (two device intervals to add up)
  bool escapeChecked = false;            // THIN_HALO: the number of escaped walks of the current vertex phase has been read back
  // cuberille_step_begin / _end: what the previous extraction on this context produced sizes the blind launches
  bool warm = false;                     // cuberille_warm_up has run its toy extraction
  bool haveHistory = false;
  u64 histV = 0, histQ = 0;
  u32 histVW = 0;
  bool histDense = false;                // ... and whether a quarter or more of its words created vertices
  bool histShortWalks = false;           // ... and whether its walks took fewer than four passes per vertex on average
  int stepMode = 0;                      // 0: no step open; 1: launched blindly (sizes on the device); 2: sized by a host read;
                                         // 3: cuberille_step_classify has run, cuberille_step_count is next
  hipEvent_t voxelHaloEvent = nullptr;   // cuberille_step_count: the halo's VOXELS are complete behind this event of the
                                         // caller's (only the walk reads them; everything before it needs their bits alone)
  Totals *hostRows = nullptr;            // pinned: the gathered totals of all ranks, read back by cuberille_step_end
  size_t hostRowsCap = 0;
  u64 pointOffset = 0;                   // of the last emit
  const u64 *extIds = nullptr;           // cuberille_set_alias_plane: planes for the next emit (device pointers)
  const float *extPts = nullptr;
  // state of the last count
  bool counted = false, haveMesh = false, slabMesh = false;
  bool pointsEmitted = false;            // the offset-free part of the emit has been launched for the current count
  bool stagesTimed = false;              // the per-stage events of the running count/emit pair are being recorded
  bool lightTiming = false;              // a few million voxels at most: ONE event pair around the extraction (every event
                                         // between two kernels costs the stream about as much as such a volume's kernels)
  bool oneCall = false;                  // inside cuberille_extract_device: no host turn between count and emit, so three
                                         // events do (start, end of the pass, end): each one more idles the stream ~10 us
  Grid g{};
  Geo geo{};
  Params prm{};
  int pixel_type = 0;
  Workspace w{};
  size_t nwords = 0, nseg = 0;
  Totals tot{};
  cuberille_result res{};
};

namespace {

int fail(cuberille_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}

#define HIP_TRY(c, call)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail((c), CUBERILLE_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));    \
  } while (0)

bool is_gfx950(int dev) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
  return std::strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

size_t pixel_size(int pt) {
  switch (pt) {
    case CUBERILLE_PIX_U8: case CUBERILLE_PIX_I8: return 1;
    case CUBERILLE_PIX_U16: case CUBERILLE_PIX_I16: return 2;
    case CUBERILLE_PIX_U32: case CUBERILLE_PIX_I32: case CUBERILLE_PIX_F32: return 4;
    case CUBERILLE_PIX_F64: case CUBERILLE_PIX_I64: case CUBERILLE_PIX_U64: return 8;
  }
  return 0;
}

// 3x3 inverse by cofactors (the parity tests hand the same matrix to the CPU checker, which
// uses the same formula; for the identity direction of every shipped volume it is exact)
void invert3(const double m[9], double inv[9]) {
  const double c00 = m[4] * m[8] - m[5] * m[7];
  const double c01 = m[5] * m[6] - m[3] * m[8];
  const double c02 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  inv[0] = c00 / det;
  inv[1] = (m[2] * m[7] - m[1] * m[8]) / det;
  inv[2] = (m[1] * m[5] - m[2] * m[4]) / det;
  inv[3] = c01 / det;
  inv[4] = (m[0] * m[8] - m[2] * m[6]) / det;
  inv[5] = (m[2] * m[3] - m[0] * m[5]) / det;
  inv[6] = c02 / det;
  inv[7] = (m[1] * m[6] - m[0] * m[7]) / det;
  inv[8] = (m[0] * m[4] - m[1] * m[3]) / det;
}

int validate(cuberille_ctx *c, const cuberille_image_desc *img, const void *vox, const cuberille_params *prm) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  if (!img || !vox || !prm) return fail(c, CUBERILLE_ERR_ARGUMENT, "null image, voxel or parameter pointer");
  if (pixel_size(img->pixel_type) == 0) return fail(c, CUBERILLE_ERR_ARGUMENT, "unknown pixel type");
  for (int i = 0; i < 3; i++) {
    if (img->dims[i] < 1) return fail(c, CUBERILLE_ERR_ARGUMENT, "image dimensions must be >= 1");
    if (img->dims[i] > 0x7fffffffLL) return fail(c, CUBERILLE_ERR_LIMIT, "image dimension exceeds 2^31-1");
    if (!(img->spacing[i] > 0.0)) return fail(c, CUBERILLE_ERR_ARGUMENT, "spacing must be > 0");
    if (img->index_start[i] < -(1LL << 30) || img->index_start[i] > (1LL << 30))
      return fail(c, CUBERILLE_ERR_LIMIT, "the buffered region's start index must lie within +-2^30");
  }
  if (prm->projection_variant < CUBERILLE_PROJECT_DEFAULT || prm->projection_variant > CUBERILLE_PROJECT_LINESEARCH)
    return fail(c, CUBERILLE_ERR_ARGUMENT, "unknown projection variant");
  if (prm->gradient_variant < CUBERILLE_GRADIENT_CENTRAL || prm->gradient_variant > CUBERILLE_GRADIENT_RECURSIVE_GAUSSIAN)
    return fail(c, CUBERILLE_ERR_ARGUMENT, "unknown gradient variant");
  if (prm->gradient_variant == CUBERILLE_GRADIENT_RECURSIVE_GAUSSIAN && prm->project_vertices && c->holdGradient)
    return fail(c, CUBERILLE_ERR_ARGUMENT, "cuberille_hold_gradient holds the central-difference gradient image the reference ships: not offered with the recursive-Gaussian one");
  if (prm->gradient_variant == CUBERILLE_GRADIENT_RECURSIVE_GAUSSIAN && prm->project_vertices)
    for (int i = 0; i < 3; i++)
      if (img->dims[i] < 4)   // (ITK's recursive filter throws for shorter lines)
        return fail(c, CUBERILLE_ERR_ARGUMENT, "the recursive-Gaussian gradient needs at least 4 voxels along every axis");
  // the iso value is an InputPixelType in the reference (h:180-181): for the integer pixel types it must convert
  // without leaving the type's range (a fraction is cut off like a C cast does)
  double lo = 0.0, hi = 0.0;
  switch (img->pixel_type) {
    case CUBERILLE_PIX_U8: hi = 255.0; break;
    case CUBERILLE_PIX_I8: lo = -128.0; hi = 127.0; break;
    case CUBERILLE_PIX_U16: hi = 65535.0; break;
    case CUBERILLE_PIX_I16: lo = -32768.0; hi = 32767.0; break;
    case CUBERILLE_PIX_U32: hi = 4294967295.0; break;
    case CUBERILLE_PIX_I32: lo = -2147483648.0; hi = 2147483647.0; break;
    default: lo = hi = 0.0; break;
  }
  if (hi != lo && !(prm->iso_value > lo - 1.0 && prm->iso_value < hi + 1.0))
    return fail(c, CUBERILLE_ERR_ARGUMENT, "iso value is not representable in the pixel type");
  return CUBERILLE_OK;
}

}  // namespace

extern "C" {

int cuberille_abi_version(void) { return CUBERILLE_ABI_VERSION; }

int cuberille_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int d = 0; d < n; d++) if (is_gfx950(d)) ok++;
  return ok;
}

const char *cuberille_last_error(const cuberille_ctx *ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int cuberille_create(cuberille_ctx **out, int device_id) {
  if (!out) return fail(nullptr, CUBERILLE_ERR_ARGUMENT, "null output pointer");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(nullptr, CUBERILLE_ERR_NO_DEVICE, "no HIP device: the cuberille hot path has no CPU fallback");
  if (device_id < 0 || device_id >= n) return fail(nullptr, CUBERILLE_ERR_ARGUMENT, "device id out of range");
  if (!is_gfx950(device_id))
    return fail(nullptr, CUBERILLE_ERR_NO_DEVICE, "device is not gfx950 (MI355X); this library carries gfx950 code only");
  cuberille_ctx *c = new (std::nothrow) cuberille_ctx;
  if (!c) return fail(nullptr, CUBERILLE_ERR_ARGUMENT, "out of host memory");
  c->device = device_id;
  hipError_t e = hipSetDevice(device_id);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipHostMalloc((void **)&c->hostTotals, sizeof(Totals), hipHostMallocDefault);
  for (int i = 0; i < 8 && e == hipSuccess; i++) e = hipEventCreate(&c->ev[i]);
  if (e != hipSuccess) {
    g_create_error = std::string("cuberille_create: ") + hipGetErrorString(e);
    cuberille_destroy(c);
    return CUBERILLE_ERR_HIP;
  }
  c->stream = c->own;
  *out = c;
  return CUBERILLE_OK;
}

void cuberille_destroy(cuberille_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->own) (void)hipStreamSynchronize(c->own);
  if (c->copyStream) (void)hipStreamSynchronize(c->copyStream);
  DevBuf *bufs[] = {&c->voxOwn, &c->bits, &c->flatBits, &c->occ, &c->prefix, &c->segPre, &c->blockTot, &c->blockBase,
                    &c->points, &c->cells, &c->cmap, &c->headV, &c->headQ, &c->vqueue, &c->escList,
                    &c->gradImg, &c->rgA, &c->rgB, &c->rgScratch, &c->heldGrad};
  for (DevBuf *b : bufs) b->release();
  c->hostPoints.release();
  c->hostCells.release();
  if (c->hostTotals) (void)hipHostFree(c->hostTotals);
  if (c->hostOcc) (void)hipHostFree(c->hostOcc);
  if (c->hostRows) (void)hipHostFree(c->hostRows);
  for (int i = 0; i < 2; i++) {
    if (c->stage[i]) (void)hipHostFree(c->stage[i]);
    if (c->stageFree[i]) (void)hipEventDestroy(c->stageFree[i]);
    if (c->chunkIn[i]) (void)hipEventDestroy(c->chunkIn[i]);
  }
  for (int i = 0; i < 8; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->copyStream) (void)hipStreamDestroy(c->copyStream);
  if (c->own) (void)hipStreamDestroy(c->own);
  delete c;
}

int cuberille_set_stream(cuberille_ctx *c, void *hip_stream) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own;
  return CUBERILLE_OK;
}

}  // extern "C"

namespace {

// How far, in slices, the projection can carry a vertex away from the slice it was created in, plus the cell's
// upper neighbour, the gradient ring and rounding: what a slab must hold beyond its owned range on each side.
// The walk moves at most step * sum(relax^k, k = 0 .. max_steps+1) in physical space (txx:449-470); the row of
// PhysicalPointToIndex for z turns that into slices.
long long projection_reach(const Geo &geo, const Params &p) {
  if (!p.project) return 0;
  const double n = (double)p.max_steps + 2.0;
  double travel;
  if (p.relax >= 1.0) travel = p.step * n;
  else if (p.relax <= 0.0) travel = p.step;
  else travel = p.step * (1.0 - std::pow(p.relax, n)) / (1.0 - p.relax);
  const double rowNorm = std::sqrt(geo.p2i[6] * geo.p2i[6] + geo.p2i[7] * geo.p2i[7] + geo.p2i[8] * geo.p2i[8]);
  // Where the walk STARTS: the reference moves the corner's index position to physical space and takes half a spacing off
  // every PHYSICAL axis (txx:266-270) -- half a voxel back along every index axis only while the direction matrix is the
  // identity.  Under a tilted direction the start lies row_z(PhysicalPointToIndex) . spacing/2 slices below the corner's
  // index instead of 1/2: with spacing (3, 1.7, 0.25) up to ten slices away from the lattice corner (found by
  // tests/fuzz_campaign.py, round 5: slabs cut to the old figure clamped four walks of a 19-slice volume).
  const double startOff = (geo.p2i[6] * geo.spacing[0] + geo.p2i[7] * geo.spacing[1] + geo.p2i[8] * geo.spacing[2]) * 0.5;
  const double astray = std::fabs(startOff - 0.5);
  const double slices = std::ceil(travel * rowNorm) + (astray < 1e-9 ? 0.0 : std::ceil(astray));
  if (!(slices < 1e9)) return 1000000000LL;
  return (long long)slices + 3;
}

void resolve(const cuberille_image_desc *img, const cuberille_params *prm, Geo &geo, Params &p) {
  // geometry and parameters (txx:75-85)
  double maxSpacing = img->spacing[0];
  for (int i = 0; i < 3; i++) {
    geo.spacing[i] = img->spacing[i];
    geo.origin[i] = img->origin[i];
    geo.gcoef[i] = (float)(0.5 * (1.0 / img->spacing[i]));
    geo.istart[i] = (int)img->index_start[i];
    if (img->spacing[i] > maxSpacing) maxSpacing = img->spacing[i];
  }
  for (int i = 0; i < 9; i++) geo.dir[i] = img->direction[i];
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < 3; k++) geo.i2p[r * 3 + k] = geo.dir[r * 3 + k] * geo.spacing[k];
  invert3(geo.i2p, geo.p2i);
  p.iso = prm->iso_value;
  p.isoInt = (long long)prm->iso_value_int;
  if (img->pixel_type == CUBERILLE_PIX_I64) p.iso = (double)prm->iso_value_int;          // what the walk compares with
  if (img->pixel_type == CUBERILLE_PIX_U64) p.iso = (double)(uint64_t)prm->iso_value_int;
  p.thr = prm->distance_threshold;
  p.step = prm->step_length < 0.0 ? maxSpacing * 0.25 : prm->step_length;
  p.relax = prm->relaxation;
  p.max_steps = prm->max_steps;
  p.triangles = prm->generate_triangles != 0;
  p.project = prm->project_vertices != 0;
  p.q1 = prm->emulate_empty_slice_aliasing != 0;
  p.variant = prm->projection_variant;
  p.gradVariant = prm->gradient_variant;
}

// RecursiveGaussianImageFilter::SetUp of ITK 3.x (Deriche's fourth-order recursive Gaussian): the 20 coefficients of one
// separable pass -- order 0 smoothing, order 1 first derivative with NormalizeAcrossScale (txx:490) -- for `sigma` in
// physical units on an axis of the given spacing.  Evaluated on the host in double, once per extraction; the kernels
// (k_rg_pass) only run the recurrences.  Layout = cuberille::DericheCoef.
void deriche_setup(double sigma, double spacing, int order, double out[20]) {
  static const double A1[2] = {1.3530, -0.6724}, B1[2] = {1.8151, -3.4327}, W1 = 0.6681, L1 = -1.3932;
  static const double A2[2] = {-0.3531, 0.6724}, B2[2] = {0.0902, 0.6100}, W2 = 2.0787, L2 = -1.3732;
  const double sigmad = sigma / spacing;
  const double cos1 = std::cos(W1 / sigmad), cos2 = std::cos(W2 / sigmad), sin1 = std::sin(W1 / sigmad), sin2 = std::sin(W2 / sigmad);
  const double exp1 = std::exp(L1 / sigmad), exp2 = std::exp(L2 / sigmad);
  double D4 = exp1 * exp1 * exp2 * exp2;
  double D3 = -2 * cos1 * exp1 * exp2 * exp2;
  D3 += -2 * cos2 * exp2 * exp1 * exp1;
  double D2 = 4 * cos2 * cos1 * exp1 * exp2;
  D2 += exp1 * exp1 + exp2 * exp2;
  const double D1 = -2 * (exp2 * cos2 + exp1 * cos1);
  const double SD = 1.0 + D1 + D2 + D3 + D4;
  const double DD = D1 + 2 * D2 + 3 * D3 + 4 * D4;
  const double a1 = A1[order], b1 = B1[order], a2 = A2[order], b2 = B2[order];
  double N0 = a1 + a2;
  double N1 = exp2 * (b2 * sin2 - (a2 + 2 * a1) * cos2);
  N1 += exp1 * (b1 * sin1 - (a1 + 2 * a2) * cos1);
  double N2 = (a1 + a2) * cos2 * cos1;
  N2 -= b1 * cos2 * sin1 + b2 * cos1 * sin2;
  N2 *= 2 * exp1 * exp2;
  N2 += a2 * exp1 * exp1 + a1 * exp2 * exp2;
  double N3 = exp2 * exp1 * exp1 * (b2 * sin2 - a2 * cos2);
  N3 += exp1 * exp2 * exp2 * (b1 * sin1 - a1 * cos1);
  const double SN = N0 + N1 + N2 + N3;
  const double DN = N1 + 2 * N2 + 3 * N3;
  double M1, M2, M3, M4;
  if (order == 0) {
    const double alpha0 = 2 * SN / SD - N0;
    N0 /= alpha0; N1 /= alpha0; N2 /= alpha0; N3 /= alpha0;
    M1 = N1 - D1 * N0; M2 = N2 - D2 * N0; M3 = N3 - D3 * N0; M4 = -D4 * N0;
  } else {
    double alpha1 = 2 * (SN * DD - DN * SD) / (SD * SD);
    alpha1 *= 1.0;                                // (spacing is positive here: no sign flip)
    N0 *= sigma / alpha1; N1 *= sigma / alpha1; N2 *= sigma / alpha1; N3 *= sigma / alpha1;
    M1 = -(N1 - D1 * N0); M2 = -(N2 - D2 * N0); M3 = -(N3 - D3 * N0); M4 = D4 * N0;
  }
  const double sn = N0 + N1 + N2 + N3, sm = M1 + M2 + M3 + M4, sd = 1.0 + D1 + D2 + D3 + D4;
  const double v[20] = {N0, N1, N2, N3, D1, D2, D3, D4, M1, M2, M3, M4,
                        D1 * sn / sd, D2 * sn / sd, D3 * sn / sd, D4 * sn / sd, D1 * sm / sd, D2 * sm / sd, D3 * sm / sd, D4 * sm / sd};
  for (int i = 0; i < 20; i++) out[i] = v[i];
}

// First half of a count: layout, parameters, workspace, zeroed state.  The caller then thresholds the slices
// (all at once, or z-range by z-range as they arrive) and calls count_finish.
int count_prepare(cuberille_ctx *c, const cuberille_image_desc *img, const void *dev_voxels, const cuberille_params *prm,
                  const cuberille_slab *slab) {
  c->counted = false;
  c->pointsEmitted = false;
  c->haveMesh = false;
  c->hostMeshValid = false;
  c->stepMode = 0;
  c->voxelHaloEvent = nullptr;
  c->aliasBelowBuffer = false;
  c->aliasMustResolve = false;
  c->aliasZ = -1;
  HIP_TRY(c, hipSetDevice(c->device));

  // ---- layout -----------------------------------------------------------------------------
  Grid g{};
  g.nx = (int)img->dims[0]; g.ny = (int)img->dims[1]; g.nzb = (int)img->dims[2];
  g.W = (g.nx + 63) / 64;
  g.lastpos = (g.nx - 1) & 63;
  g.wShift = g.yShift = -1;
  for (int b = 0; b < 31; b++) {
    if (g.W == (1 << b)) g.wShift = b;
    if (g.ny == (1 << b)) g.yShift = b;
  }
  Geo geo{};
  Params p{};
  resolve(img, prm, geo, p);
  const bool whole = !slab || (slab->global_nz == 0 && slab->z_begin == 0 && slab->own_z0 == 0 && slab->own_z1 == 0);   // (all-zero slab = whole volume)
  if (whole) {
    g.gnz = g.nzb; g.zglob0 = 0; g.oz0 = 0; g.oz1 = g.nzb;
  } else {
    if (slab->global_nz < 1 || slab->z_begin < 0 || slab->z_begin + g.nzb > slab->global_nz ||
        slab->own_z0 < slab->z_begin || slab->own_z1 > slab->z_begin + g.nzb || slab->own_z0 >= slab->own_z1)
      return fail(c, CUBERILLE_ERR_ARGUMENT, "slab ranges are inconsistent with the buffer");
    // the owned range needs 2 slices below (ids of corners created one slice down depend on the
    // slice below that) and 1 above; with the projection on, as far as a walk can reach (both unless the
    // volume ends there)
    // (a THIN_HALO slab promises the topology's slices only; walks that want more are put aside, not clamped)
    if (p.project && p.gradVariant != CUBERILLE_GRADIENT_CENTRAL)
      return fail(c, CUBERILLE_ERR_ARGUMENT, "the recursive-Gaussian gradient filters whole lines of the volume: not offered on slabs");
    if (p.project && c->holdGradient)
      return fail(c, CUBERILLE_ERR_ARGUMENT, "a held gradient image (cuberille_hold_gradient) belongs to a whole volume: not offered on slabs");
    const bool thin = (slab->flags & CUBERILLE_SLAB_THIN_HALO) != 0;
    if (thin && p.project && p.variant != CUBERILLE_PROJECT_DEFAULT)
      return fail(c, CUBERILLE_ERR_ARGUMENT, "a THIN_HALO slab is only offered with the default projection branch");
    long long halo = thin ? 0 : projection_reach(geo, p);
    const long long lo = halo > 2 ? halo : 2, hi = halo > 1 ? halo : 1;
    const long long needLo = slab->own_z0 >= lo ? slab->own_z0 - lo : 0;
    const long long needHi = slab->own_z1 + hi < slab->global_nz ? slab->own_z1 + hi : slab->global_nz;
    if (slab->z_begin > needLo || slab->z_begin + g.nzb < needHi) {
      char msg[256];
      snprintf(msg, sizeof msg, "slab buffer must hold %lld halo slices below and %lld above the owned range for these "
               "parameters (cuberille_required_halo); it holds %lld and %lld", lo, hi,
               (long long)(slab->own_z0 - slab->z_begin), (long long)(slab->z_begin + g.nzb - slab->own_z1));
      return fail(c, CUBERILLE_ERR_HALO, msg);
    }
    g.gnz = slab->global_nz; g.zglob0 = slab->z_begin;
    g.oz0 = (int)(slab->own_z0 - slab->z_begin);
    g.oz1 = (int)(slab->own_z1 - slab->z_begin);
  }
  g.cz0 = g.oz0 > 0 ? g.oz0 - 1 : 0;
  g.cmapLinear = c->tune.cmap_linear;
  const size_t nrowsAll = (size_t)g.ny * g.nzb;
  const size_t nwordsAll = nrowsAll * g.W;
  const size_t nwords = (size_t)(g.oz1 - g.cz0) * g.ny * g.W;
  const size_t nseg = (nwords + 63) / 64;
  const size_t nblk = (nwords + COUNT_WB - 1) / COUNT_WB;
  if (nseg > 0x7fffffffULL) return fail(c, CUBERILLE_ERR_LIMIT, "volume too large for one device scan");

  // ---- workspace ----------------------------------------------------------------------------------
  // (+ one slice past the buffer: a slab may be handed the source slice of quirk Q1 from the rank below)
  HIP_TRY(c, c->bits.reserve((nwordsAll + (size_t)g.ny * g.W) * sizeof(u64)));
  c->slabMode = !whole;
  c->thinHalo = !whole && (slab->flags & CUBERILLE_SLAB_THIN_HALO) != 0 && p.project;
  c->pointsStartedEarly = false;
  c->escapeChecked = false;
  c->extIds = nullptr;
  c->extPts = nullptr;
  // the totals and the per-slice occupancy share one allocation: one memset zeroes both
  static_assert(sizeof(Totals) % 16 == 0, "the occupancy words follow the totals");
  HIP_TRY(c, c->occ.reserve(sizeof(Totals) + (size_t)g.nzb * sizeof(u32)));
  HIP_TRY(c, c->prefix.reserve((nwords + 4) * sizeof(u32)));   // (+ the tail of a 16-byte read at the last words: locate_word_wave)
  HIP_TRY(c, c->segPre.reserve(nseg * sizeof(u64)));
  HIP_TRY(c, c->blockTot.reserve((nblk + 2 * (nblk / 8192 + 1)) * sizeof(u64)));     // (+ the sums of its chunks of 8192: k_block_partial)
  HIP_TRY(c, c->blockBase.reserve(nblk * 2 * sizeof(u64)));
  Workspace w{};
  w.flatBits = nullptr;
  if (g.nx % 64 != 0) {   // ragged rows: thresholded as one flat stream first, then cut into rows
    if (c->flatBits.reserve((nwordsAll + 32) * sizeof(u64)) == hipSuccess) w.flatBits = (u64 *)c->flatBits.p;
    else (void)hipGetLastError();
  }
  w.vqueue = nullptr;
  if (nwords < 0xffffffffULL && !c->tune.no_vqueue && c->vqueue.reserve(nwords * sizeof(u32)) == hipSuccess)
    w.vqueue = (u32 *)c->vqueue.p;
  else (void)hipGetLastError();
  w.vox = dev_voxels;
  w.bits = (u64 *)c->bits.p; w.sliceOcc = (u32 *)((char *)c->occ.p + sizeof(Totals));
  w.prefix = (u32 *)c->prefix.p;
  w.segPre = (u64 *)c->segPre.p; w.blockTot = (u64 *)c->blockTot.p; w.blockBase = (u64 *)c->blockBase.p;
  w.totals = (Totals *)c->occ.p;

  hipStream_t s = c->stream;
  HIP_TRY(c, hipMemsetAsync(w.totals, 0, sizeof(Totals) + (size_t)g.nzb * sizeof(u32), s));
  c->stagesTimed = c->tune.stage_timing != 0;
  c->lightTiming = !c->stagesTimed && (u64)g.nx * (u64)g.ny * (u64)g.nzb <= (4ull << 20);
  HIP_TRY(c, hipEventRecord(c->ev[0], s));
  c->g = g; c->geo = geo; c->prm = p; c->pixel_type = img->pixel_type; c->w = w;
  c->nwords = nwords; c->nseg = nseg;
  return CUBERILLE_OK;
}

// Second half: count + scan of the thresholded volume (launches only).
int count_launch(cuberille_ctx *c, const Gate &gate) {
  hipStream_t s = c->stream;
  if (c->stagesTimed) HIP_TRY(c, hipEventRecord(c->ev[1], s));
  HIP_TRY(c, launch_occupancy(c->pixel_type, c->w, c->g, c->tune, s));
  // (the LDS-tiled form pays where most words carry surface -- 2048^3 noise -- and costs where few do: it stages every
  //  row, a sparse block's untiled form skips whole words; the previous extraction's density decides)
  int tiled = c->tune.count_variant >= 0 ? c->tune.count_variant : (c->haveHistory && c->histDense ? 3 : 0);
  // A launch whose blocks are all resident at once is a block's latency, whatever the field: eight dependent trips to memory per
  // thread for the faces (2600 cycles each, profiles/microbench/r5_count_phase_stamps.log) and two or three more for the corner
  // logic from memory; the LDS tile (one block per workgroup) makes that one coalesced copy -- 512^3 sphere 0.0352 -> 0.0309 ms,
  // 512^3 Marschner-Lobb 0.0466 -> 0.0388; with several rounds of blocks (768^3: 3456) the form that skips empty words wins again.
  if (c->tune.count_variant < 0 && tiled == 0 && (c->nwords + COUNT_WB - 1) / COUNT_WB <= 1280 && (c->nwords + COUNT_WB - 1) / COUNT_WB > 64)
    tiled = 2;
  if (c->tune.count_variant < 0 && !c->haveHistory && c->nwords >= (1u << 22) && c->g.wShift >= 0) {
    // no previous extraction to go by (round-4 review: a one-shot caller of a dense field paid 2.3 ms for a 1.2 ms count):
    // a sample of THIS volume's bit volume picks the form -- one small launch and one more wait, on a context's first
    // extraction only, which waits for its counts anyway.  (Totals::iters carries the sample: the walk, which counts its
    // passes there, is a long way off; zeroed again behind the read.)
    u64 *slot = &c->w.totals->iters;
    HIP_TRY(c, launch_density_probe(c->w, c->g, c->nwords, slot, s));
    HIP_TRY(c, hipMemcpyAsync(&c->hostTotals->iters, slot, sizeof(u64), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemsetAsync(slot, 0, sizeof(u64), s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const u64 v = c->hostTotals->iters;
    const u64 mixed = v & 0xffffffffull, sampled = v >> 32;
    if (sampled && mixed * 4 >= sampled) tiled = 3;
  }
  HIP_TRY(c, launch_count(c->w, c->g, c->nwords, c->prm.q1, gate, tiled, c->tune.count_no_fold, s));
  if (!c->lightTiming) HIP_TRY(c, hipEventRecord(c->ev[2], s));
  return CUBERILLE_OK;
}

// the totals (and a slab's per-slice occupancy) on their way to pinned memory
int totals_to_host(cuberille_ctx *c) {
  hipStream_t s = c->stream;
  const Grid &g = c->g;
  HIP_TRY(c, hipMemcpyAsync(c->hostTotals, c->w.totals, sizeof(Totals), hipMemcpyDeviceToHost, s));
  if (c->slabMode) {
    // the slab status the multi-GPU driver asks for next rides in the same synchronisation
    if (c->hostOccCap < (size_t)g.nzb) {
      if (c->hostOcc) (void)hipHostFree(c->hostOcc);
      c->hostOcc = nullptr;
      c->hostOccCap = 0;
      HIP_TRY(c, hipHostMalloc((void **)&c->hostOcc, (size_t)g.nzb * sizeof(uint32_t), hipHostMallocDefault));
      c->hostOccCap = (size_t)g.nzb;
    }
    HIP_TRY(c, hipMemcpyAsync(c->hostOcc, c->w.sliceOcc, (size_t)g.nzb * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  }
  return CUBERILLE_OK;
}

// ... and, once the stream has been waited for, taken over as the state of a finished count
void adopt_totals(cuberille_ctx *c, uint64_t *n_points, uint64_t *n_cells) {
  const Grid &g = c->g;
  c->tot = *c->hostTotals;
  // quirk Q1 reaching below this slab: certain when the source slice is in the halo (the emit then insists on
  // cuberille_recount), possible when the search ran off the buffer's bottom (the ranks below know)
  c->aliasMustResolve = (c->tot.err & ERRF_ALIAS_UNKNOWN) != 0;
  c->aliasBelowBuffer = (c->tot.err & (ERRF_ALIAS_BELOW_BUFFER | ERRF_ALIAS_UNKNOWN)) != 0;
  // the slice it concerns: only the FIRST occupied slice of the counted range can have its source outside of it
  if (c->slabMode && c->aliasZ < 0 && c->aliasBelowBuffer)
    for (int z = g.cz0; z < g.oz1; z++)
      if (c->hostOcc[(size_t)z]) { c->aliasZ = z; break; }
  c->counted = true;
  std::memset(&c->res, 0, sizeof(c->res));
  c->res.n_points = c->tot.totV - c->tot.V0;
  c->res.n_cells = (c->tot.totQ - c->tot.Q0) * (c->prm.triangles ? 2 : 1);
  c->res.verts_per_cell = c->prm.triangles ? 3 : 4;
  if (n_points) *n_points = c->res.n_points;
  if (n_cells) *n_cells = c->res.n_cells;
}

int classify_slab(cuberille_ctx *c, const cuberille_image_desc *img, const cuberille_slab *slab);

int count_finish(cuberille_ctx *c, uint64_t *n_points, uint64_t *n_cells) {
  int rc = count_launch(c, Gate{});
  if (rc) return rc;
  rc = totals_to_host(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  adopt_totals(c, n_points, n_cells);
  return CUBERILLE_OK;
}

}  // namespace

extern "C" {

int cuberille_required_halo(const cuberille_image_desc *img, const cuberille_params *prm, int64_t *below, int64_t *above) {
  if (!img || !prm) return CUBERILLE_ERR_ARGUMENT;
  for (int i = 0; i < 3; i++) if (!(img->spacing[i] > 0.0)) return CUBERILLE_ERR_ARGUMENT;
  Geo geo{};
  Params p{};
  resolve(img, prm, geo, p);
  const long long halo = projection_reach(geo, p);
  if (below) *below = halo > 2 ? halo : 2;
  if (above) *above = halo > 1 ? halo : 1;
  return CUBERILLE_OK;
}

int cuberille_minimum_halo(const cuberille_image_desc *img, const cuberille_params *prm, int64_t *below, int64_t *above) {
  if (!img || !prm) return CUBERILLE_ERR_ARGUMENT;
  for (int i = 0; i < 3; i++) if (!(img->spacing[i] > 0.0)) return CUBERILLE_ERR_ARGUMENT;
  Geo geo{};
  Params p{};
  resolve(img, prm, geo, p);
  long long lo = 2, hi = 1;                      // the topology: ghost slice + the one under it; one slice above
  if (p.project) {
    // a vertex starts at index-space z = cz + oz, oz = -(row z of PhysicalPointToIndex) . spacing / 2 (txx:266-270: the
    // half-spacing shift is taken per PHYSICAL axis; -1/2 for an axis-aligned image); its cell floor(cz + oz) reads
    // slices floor - 1 .. floor + 2 (gradient ring).  Vertices that matter sit on planes own_z0 .. own_z1.
    double oz = 0.0;
    for (int k = 0; k < 3; k++) oz -= geo.p2i[6 + k] * (geo.spacing[k] / 2.0);
    const long long b = 1 - (long long)std::floor(oz - 1e-6), a = (long long)std::floor(oz + 1e-6) + 3;
    if (b > lo) lo = b;
    if (a > hi) hi = a;
  }
  if (below) *below = lo;
  if (above) *above = hi;
  return CUBERILLE_OK;
}

int cuberille_count(cuberille_ctx *c, const cuberille_image_desc *img, const void *dev_voxels,
                    const cuberille_params *prm, const cuberille_slab *slab, uint64_t *n_points, uint64_t *n_cells) {
  int rc = validate(c, img, dev_voxels, prm);
  if (rc) return rc;
  rc = count_prepare(c, img, dev_voxels, prm, slab);
  if (rc) return rc;
  rc = classify_slab(c, img, slab);
  if (rc) return rc;
  return count_finish(c, n_points, n_cells);
}

}  // extern "C"

namespace {

// threshold the buffer of a prepared count: at once, or the owned slices now and the halo slices behind the caller's event
int classify_slab(cuberille_ctx *c, const cuberille_image_desc *img, const cuberille_slab *slab) {
  const Grid &g = c->g;
  hipStream_t s = c->stream;
  if (slab && slab->voxels_ready_event) HIP_TRY(c, hipStreamWaitEvent(s, (hipEvent_t)slab->voxels_ready_event, 0));
  if (slab && slab->halo_ready_event && (g.oz0 > 0 || g.oz1 < g.nzb)) {
    // the caller's halo exchange is still in flight: threshold the owned slices now, the halo
    // slices once the event it recorded behind the exchange has fired
    HIP_TRY(c, launch_classify(img->pixel_type, c->w, g, c->prm, g.oz0, g.oz1, c->tune, s));
    HIP_TRY(c, hipStreamWaitEvent(s, (hipEvent_t)slab->halo_ready_event, 0));
    HIP_TRY(c, launch_classify(img->pixel_type, c->w, g, c->prm, 0, g.oz0, c->tune, s));
    HIP_TRY(c, launch_classify(img->pixel_type, c->w, g, c->prm, g.oz1, g.nzb, c->tune, s));
  } else {
    HIP_TRY(c, launch_classify(img->pixel_type, c->w, g, c->prm, 0, g.nzb, c->tune, s));
  }
  return CUBERILLE_OK;
}

}  // namespace

extern "C" {

int cuberille_recount(cuberille_ctx *c, const void *dev_source_bits, uint64_t *n_points, uint64_t *n_cells) {
  if (!c || !dev_source_bits) return CUBERILLE_ERR_ARGUMENT;
  if (!c->counted || !c->slabMode)
    return fail(c, CUBERILLE_ERR_STATE, "cuberille_recount follows a successful cuberille_count on a slab");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t sliceWords = (size_t)c->g.ny * c->g.W;
  // the source slice goes one past the buffer's last slice; the count then finds it where alias_of points
  HIP_TRY(c, hipMemcpyAsync((u64 *)c->bits.p + sliceWords * (size_t)c->g.nzb, dev_source_bits, sliceWords * sizeof(u64),
                            hipMemcpyDeviceToDevice, s));
  c->g.extAlias = 1;
  c->counted = false;
  c->pointsEmitted = false;                  // the counts change: whatever cuberille_emit_points started is void
  c->pointsStartedEarly = false;
  c->escapeChecked = false;
  HIP_TRY(c, hipMemsetAsync(c->w.totals, 0, sizeof(Totals), s));
  HIP_TRY(c, hipEventRecord(c->ev[0], s));   // ms_pass of a recounted slab: this count alone, not the host time since the first
  return count_finish(c, n_points, n_cells);
}

int cuberille_slice_bits_device(cuberille_ctx *c, int64_t z_global, const uint64_t **dev_words, size_t *n_words) {
  if (!c || !dev_words || !n_words) return CUBERILLE_ERR_ARGUMENT;
  if (!c->counted && !c->haveMesh) return fail(c, CUBERILLE_ERR_STATE, "no classified volume on this context");
  const long long z = z_global - c->g.zglob0;
  if (z < 0 || z >= c->g.nzb) return fail(c, CUBERILLE_ERR_ARGUMENT, "slice outside the buffer");
  *n_words = (size_t)c->g.ny * c->g.W;
  *dev_words = (const uint64_t *)c->bits.p + *n_words * (size_t)z;
  return CUBERILLE_OK;
}

int cuberille_alias_plane_device(cuberille_ctx *c, int64_t z_global, uint64_t *dev_ids, float *dev_points) {
  if (!c || !dev_ids || !dev_points) return CUBERILLE_ERR_ARGUMENT;
  if (!c->haveMesh) return fail(c, CUBERILLE_ERR_STATE, "no mesh: the plane is built from the last emit");
  const long long z = z_global - c->g.zglob0;
  if (z < c->g.oz0 || z >= c->g.oz1) return fail(c, CUBERILLE_ERR_ARGUMENT, "slice outside the owned range");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, launch_alias_plane(c->w, c->g, (int)z, c->pointOffset, (u64 *)dev_ids, dev_points, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CUBERILLE_OK;
}

int cuberille_set_alias_plane(cuberille_ctx *c, const uint64_t *dev_ids, const float *dev_points) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  if (!c->counted || !c->g.extAlias)
    return fail(c, CUBERILLE_ERR_STATE, "cuberille_set_alias_plane follows cuberille_recount and precedes cuberille_emit");
  c->extIds = (const u64 *)dev_ids;
  c->extPts = dev_points;
  return CUBERILLE_OK;
}

}  // extern "C"

namespace {

// The part of the emit that needs no id offsets: buffers, head tables, vertex scatter, projection.  Runs once per count
// (cuberille_emit_points may have started it already, while the caller was gathering the counts of the other ranks).
// dyn (cuberille_step_begin): buffers and launches are sized for the cover values, the kernels read the real counts
// from the device and run only when they fit (Totals::go).
int emit_points_phase(cuberille_ctx *c, bool dyn = false, u64 coverV = 0, u64 coverQ = 0, u32 coverVW = 0) {
  if (c->pointsEmitted) return CUBERILLE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const u64 nV = dyn ? coverV : c->tot.totV;                 // ghost + owned
  const u64 nGhost = dyn ? 0 : c->tot.V0;
  const u64 totQ = dyn ? coverQ : c->tot.totQ;
  const u64 nQ = dyn ? coverQ : c->tot.totQ - c->tot.Q0;
  const u32 nVW = dyn ? coverVW : c->tot.nVertexWords;
  // room behind this rank's points for the positions of a plane of the rank below's vertices (quirk Q1 across slabs)
  const size_t planeCorners = (c->slabMode || c->g.extAlias) ? (size_t)(c->g.nx + 1) * (c->g.ny + 1) : 0;
  // (a first extraction also makes room for the blind launches of the one behind it: gate.coverV / coverQ of cuberille_step_begin)
  const u64 nextV = dyn ? nV : nV + nV / 4 + 4096, nextQ = dyn ? nQ : totQ + totQ / 4 + 4096;
  HIP_TRY(c, c->points.reserve_covering((size_t)(nV + planeCorners ? nV + planeCorners : 1) * 3 * sizeof(float),
                                        (size_t)(nextV + planeCorners) * 3 * sizeof(float)));
  HIP_TRY(c, c->cells.reserve_covering((size_t)(nQ ? nQ : 1) * (c->prm.triangles ? 6 : 4) * sizeof(u64),
                                       (size_t)nextQ * (c->prm.triangles ? 6 : 4) * sizeof(u64)));
  Workspace &w = c->w;
  w.points = (float *)c->points.p;
  w.cells = (u64 *)c->cells.p;
  // dense corner -> vertex map (4 B per lattice corner of the buffer); when it cannot be had
  // (more than 2^32 vertices, or no memory) the cell kernel recomputes ids instead
  w.cmap = nullptr;
  if (nV < 0xffffffffULL && !c->tune.no_cmap) {
    // (bricks over the lattice corners 0..nx, 0..ny, 0..nzb + 1: the last plane also takes the handed-over one)
    const size_t mapBytes = c->g.cmapLinear ? (size_t)(c->g.nx + 1) * (c->g.ny + 1) * (c->g.nzb + 2) * sizeof(u32)
                          : (((size_t)c->g.nx + 4) >> 2) * (((size_t)c->g.ny + 4) >> 2) * (((size_t)c->g.nzb + 3) >> 1) * 32 * sizeof(u32);
    if (c->cmap.reserve(mapBytes) == hipSuccess) w.cmap = (u32 *)c->cmap.p;
    else (void)hipGetLastError();
  }
  // head tables for the per-wave inverse mapping (4 B per 64 outputs)
  w.headV = w.headQ = nullptr;
  if (c->nwords < 0xffffffffULL && !c->tune.no_heads) {
    if (c->headQ.reserve_covering((size_t)(totQ / 64 + 2) * sizeof(u32), (size_t)(nextQ / 64 + 2) * sizeof(u32)) == hipSuccess)
      w.headQ = (u32 *)c->headQ.p;
    if (!w.vqueue && c->headV.reserve_covering((size_t)(nV / 64 + 2) * sizeof(u32), (size_t)(nextV / 64 + 2) * sizeof(u32)) == hipSuccess)
      w.headV = (u32 *)c->headV.p;
    (void)hipGetLastError();
  }
  // THIN_HALO: room for the vertices whose walk leaves the buffer (more than these: the step is redone with the deep halo)
  w.escList = nullptr;
  w.escCap = 0;
  if (c->thinHalo) {
    const size_t cap = nV < ESCAPE_LIST_CAP ? (size_t)nV + 1 : (size_t)ESCAPE_LIST_CAP;
    HIP_TRY(c, c->escList.reserve(cap * sizeof(u32)));
    w.escList = (u32 *)c->escList.p;
    w.escCap = (u32)cap;
  }
  // (the blind form needs every scratch table: the fallbacks without them size their launches from the counts)
  if (dyn && (!w.cmap || !w.headQ || !w.vqueue || c->tune.points_variant != 3))
    return fail(c, CUBERILLE_ERR_STATE, "internal: blind launch without the scratch tables");
  hipStream_t s = c->stream;
  if (!c->lightTiming && (!c->oneCall || c->stagesTimed)) HIP_TRY(c, hipEventRecord(c->ev[4], s));
  HIP_TRY(c, launch_heads(w, c->g, nV, totQ, dyn ? 1 : 0, s));
  HIP_TRY(c, launch_emit_points(w, c->g, c->geo, c->prm.q1, nV, nVW, c->tune, dyn ? 1 : 0, s));
  if (c->stagesTimed) HIP_TRY(c, hipEventRecord(c->ev[5], s));
  w.gradImg = nullptr;
  if (c->prm.project && c->prm.gradVariant == CUBERILLE_GRADIENT_RECURSIVE_GAUSSIAN && nV) {
    // the whole-image gradient pre-pass of txx:478-498 in its recursive-Gaussian form (the shipped central differences
    // are evaluated on the fly inside the walk and never materialised)
    const size_t nvox = (size_t)c->g.nx * c->g.ny * c->g.nzb;
    HIP_TRY(c, c->gradImg.reserve(nvox * 3 * sizeof(double)));
    HIP_TRY(c, c->rgScratch.reserve(nvox * sizeof(double)));
    HIP_TRY(c, c->rgA.reserve(nvox * sizeof(float)));
    HIP_TRY(c, c->rgB.reserve(nvox * sizeof(float)));
    w.gradImg = (double *)c->gradImg.p; w.rgScratch = (double *)c->rgScratch.p;
    w.rgA = (float *)c->rgA.p; w.rgB = (float *)c->rgB.p;
    double sigma = c->geo.spacing[0];                        // txx:489: m_MaxSpacing * 1.0
    for (int i = 1; i < 3; i++) if (c->geo.spacing[i] > sigma) sigma = c->geo.spacing[i];
    double coef[3][2][20];
    for (int ax = 0; ax < 3; ax++) {
      deriche_setup(sigma, c->geo.spacing[ax], 1, coef[ax][0]);
      deriche_setup(sigma, c->geo.spacing[ax], 0, coef[ax][1]);
    }
    HIP_TRY(c, launch_recursive_gaussian(c->pixel_type, w, c->g, c->geo, coef, s));
  }
  if (c->prm.project) {
    // (walks that end after two or three passes -- a noise field -- leave the kernel bound by its gathers, which a refill
    //  issues for the few lanes it fills: such fields refill only empty waves.  Scheduling only: results never depend on it.)
    Tuning tn = c->tune;
    // (long walks refill at 32 idle lanes: 1.161-1.166 ms against 1.178-1.192 at 16 on the headline field, three boxes, round 5;
    //  24 and 40 are no better, 48 loses 8 %)
    // (... and at 24 below 8 M vertices: 512^3 Marschner-Lobb 0.387 against 0.398 / 0.403 ms at 16 / 32, 768^3 0.702 against
    //  0.717 / 0.718)
    if (tn.proj_short < 0) tn.proj_short = c->haveHistory && c->histShortWalks ? 1 : 0;
    if (tn.proj_refill <= 0) tn.proj_refill = tn.proj_short ? 64 : (nV >= 8000000ull ? 32 : 24);
    if (c->voxelHaloEvent) {                 // the first reader of the halo's voxels
      HIP_TRY(c, hipStreamWaitEvent(s, c->voxelHaloEvent, 0));
      c->voxelHaloEvent = nullptr;
    }
    w.held = c->holdGradient && c->held.img ? &c->held : nullptr;
    HIP_TRY(c, launch_project(c->pixel_type, w, c->g, c->geo, c->prm, nV, nGhost, tn, c->thinHalo ? 1 : 0, dyn ? 1 : 0, s));
    if (c->holdGradient && !c->held.img) {
      // quirk Q3 on request: this is the context's first projecting extraction -- ComputeGradientImage() of txx:478-498
      // runs (the walk above evaluated the same taps on the fly) and its image stays for every extraction to come
      const size_t nvox = (size_t)c->g.nx * c->g.ny * c->g.nzb;
      HIP_TRY(c, c->heldGrad.reserve(nvox * 3 * sizeof(float)));
      HIP_TRY(c, launch_gradient_image(c->pixel_type, w, c->g, c->geo, (float *)c->heldGrad.p, s));
      c->held.img = (const float *)c->heldGrad.p;
      c->held.geo = c->geo;
      c->held.n[0] = c->g.nx; c->held.n[1] = c->g.ny; c->held.n[2] = c->g.nzb;
    }
  }
  if (c->stagesTimed) HIP_TRY(c, hipEventRecord(c->ev[6], s));
  c->pointsEmitted = true;
  return CUBERILLE_OK;
}

// After the last kernel of an extraction has completed and the totals are back in pinned memory: statistics, device times.
int finish_result(cuberille_ctx *c, cuberille_result *res) {
  c->tot.iters = c->hostTotals->iters;
  c->tot.stopSteps = c->hostTotals->stopSteps;
  c->tot.stopThr = c->hostTotals->stopThr;
  c->tot.nEscaped = c->hostTotals->nEscaped;
  c->tot.err = c->hostTotals->err;
  cuberille_result &r = c->res;
  r.ms_scan = 0.0f;                          // the prefix sums are part of the count stage (k_count + k_block_scan)
  if (c->stagesTimed) {                      // (the switch as it was when the count ran)
    HIP_TRY(c, hipEventElapsedTime(&r.ms_classify, c->ev[0], c->ev[1]));
    HIP_TRY(c, hipEventElapsedTime(&r.ms_count, c->ev[1], c->ev[2]));
    HIP_TRY(c, hipEventElapsedTime(&r.ms_emit_points, c->ev[4], c->ev[5]));
    HIP_TRY(c, hipEventElapsedTime(&r.ms_project, c->ev[5], c->ev[6]));
    HIP_TRY(c, hipEventElapsedTime(&r.ms_emit_cells, c->pointsStartedEarly ? c->ev[3] : c->ev[6], c->ev[7]));
  }
  float b = 0.f, b2 = 0.f;
  if (c->lightTiming) {
    // one event pair: the whole extraction as the stream saw it (a host turn between count and emit included, where
    // the caller took one); no pass figure
    HIP_TRY(c, hipEventElapsedTime(&b, c->ev[0], c->ev[7]));
    r.ms_pass = 0.f;
  } else if (c->oneCall && !c->stagesTimed) {
    // one call, no host turn in the middle: the pass, and everything behind it
    HIP_TRY(c, hipEventElapsedTime(&r.ms_pass, c->ev[0], c->ev[2]));
    HIP_TRY(c, hipEventElapsedTime(&b, c->ev[2], c->ev[7]));
  } else {
    HIP_TRY(c, hipEventElapsedTime(&r.ms_pass, c->ev[0], c->ev[2]));
    if (c->pointsStartedEarly) {
      HIP_TRY(c, hipEventElapsedTime(&b, c->ev[4], c->ev[6]));
      HIP_TRY(c, hipEventElapsedTime(&b2, c->ev[3], c->ev[7]));
    } else {
      HIP_TRY(c, hipEventElapsedTime(&b, c->ev[4], c->ev[7]));
    }
  }
  r.ms_total = r.ms_pass + b + b2;           // device time: the host's turn between count and emit is in none of the intervals
  r.proj_iterations = c->tot.iters;
  r.proj_stop_steps = r.proj_stop_threshold = 0;
  if (c->prm.project) {
    r.proj_stop_steps = c->tot.stopSteps;
    // the default branch ends every walk one way or the other (txx:456-472): only the rare way is counted on the device
    r.proj_stop_threshold = c->prm.variant == CUBERILLE_PROJECT_DEFAULT ? r.n_points - c->tot.stopSteps : c->tot.stopThr;
  }
  r.n_escaped = c->tot.nEscaped;
  // what this extraction produced sizes the blind launches of the next cuberille_step_begin on this context
  c->haveHistory = true;
  c->histV = c->tot.totV; c->histQ = c->tot.totQ; c->histVW = c->tot.nVertexWords;
  c->histDense = (u64)c->tot.nVertexWords * 4 >= (u64)c->nwords;
  c->histShortWalks = c->prm.project && c->tot.iters > 0 && c->tot.iters < 4 * c->tot.totV;
  c->stepMode = 0;
  c->haveMesh = true;
  c->counted = false;                        // the workspace now belongs to this mesh
  if (res) *res = r;
  return CUBERILLE_OK;
}

int emit_preconditions(cuberille_ctx *c, const char *who) {
  if (!c->counted) return fail(c, CUBERILLE_ERR_STATE, std::string(who) + " called before a successful cuberille_count");
  if (c->aliasMustResolve)
    return fail(c, CUBERILLE_ERR_HALO,
                "an empty slice makes the reference re-use vertices created below this slab's counted range: hand the "
                "source slice over with cuberille_recount (DESIGN.md Q1)");
  return CUBERILLE_OK;
}

}  // namespace

extern "C" {

int cuberille_emit_points(cuberille_ctx *c) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  int rc = emit_preconditions(c, "cuberille_emit_points");
  if (rc) return rc;
  if (c->pointsEmitted) return CUBERILLE_OK;
  rc = emit_points_phase(c);
  if (rc) return rc;
  // the caller turns to the other ranks now: this phase gets its own end mark, cuberille_emit starts a second interval
  if (!c->stagesTimed && !c->lightTiming && !c->oneCall) HIP_TRY(c, hipEventRecord(c->ev[6], c->stream));
  c->pointsStartedEarly = true;
  return CUBERILLE_OK;
}

int cuberille_escaped_count(cuberille_ctx *c, uint64_t *n_escaped) {
  if (!c || !n_escaped) return CUBERILLE_ERR_ARGUMENT;
  if (!c->counted || !c->pointsEmitted)
    return fail(c, CUBERILLE_ERR_STATE, "cuberille_escaped_count follows cuberille_emit_points");
  *n_escaped = 0;
  if (!c->thinHalo) return CUBERILLE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(c->hostTotals, c->w.totals, sizeof(Totals), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->tot.nEscaped = c->hostTotals->nEscaped;
  c->tot.err = c->hostTotals->err;
  c->escapeChecked = true;
  *n_escaped = (c->tot.err & ERRF_ESCAPE_OVERFLOW) ? UINT64_MAX : (uint64_t)c->tot.nEscaped;
  return CUBERILLE_OK;
}

int cuberille_reproject_escaped(cuberille_ctx *c, const void *dev_voxels, int64_t z_begin, int64_t nz) {
  if (!c || !dev_voxels) return CUBERILLE_ERR_ARGUMENT;
  if (!c->counted || !c->pointsEmitted || !c->thinHalo)
    return fail(c, CUBERILLE_ERR_STATE, "cuberille_reproject_escaped follows cuberille_emit_points on a THIN_HALO slab");
  if (c->tot.err & ERRF_ESCAPE_OVERFLOW)
    return fail(c, CUBERILLE_ERR_LIMIT, "more walks left the thin halo than the escape list holds: count the slab again with "
                                        "the full halo (cuberille_required_halo)");
  // the deeper buffer must hold what a slab without the flag would have had to
  const long long reach = projection_reach(c->geo, c->prm);
  const long long lo = reach > 2 ? reach : 2, hi = reach > 1 ? reach : 1;
  const long long own0 = c->g.zglob0 + c->g.oz0, own1 = c->g.zglob0 + c->g.oz1;
  const long long needLo = own0 >= lo ? own0 - lo : 0, needHi = own1 + hi < c->g.gnz ? own1 + hi : c->g.gnz;
  if (nz < 1 || z_begin < 0 || z_begin + nz > c->g.gnz || z_begin > needLo || z_begin + nz < needHi)
    return fail(c, CUBERILLE_ERR_HALO, "the deeper buffer does not hold the halo these parameters need (cuberille_required_halo)");
  const u64 n = c->tot.nEscaped;
  if (n == 0) return CUBERILLE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  Grid deep = c->g;
  deep.nzb = (int)nz;
  deep.zglob0 = z_begin;
  Workspace w = c->w;
  w.vox = dev_voxels;
  Tuning tn = c->tune;
  if (tn.proj_refill <= 0) tn.proj_refill = 16;
  HIP_TRY(c, launch_project(c->pixel_type, w, deep, c->geo, c->prm, n, c->tot.V0, tn, 2, 0, c->stream));
  // nobody waits any more (the list's length travelled by value): the device-side counter starts over
  HIP_TRY(c, hipMemsetAsync(&c->w.totals->nEscaped, 0, sizeof(u32), c->stream));
  c->tot.nEscaped = 0;
  return CUBERILLE_OK;
}

int cuberille_emit(cuberille_ctx *c, uint64_t point_id_offset, cuberille_result *res) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  int rc = emit_preconditions(c, "cuberille_emit");
  if (rc) return rc;
  // (a recount for the ghost slice needed the source's bits only: that slice emits no cells)
  const bool needPlane = c->g.extAlias && c->aliasZ >= c->g.oz0;
  if (needPlane && (!c->extIds || !c->extPts))
    return fail(c, CUBERILLE_ERR_STATE, "cuberille_emit after cuberille_recount needs cuberille_set_alias_plane");
  c->slabMesh = point_id_offset != 0 || c->tot.V0 != 0 || part_of_a_volume(c->g);
  c->pointOffset = point_id_offset;
  rc = emit_points_phase(c);
  if (rc) return rc;
  if (c->thinHalo) {
    // no cell is written while a vertex waits for slices this buffer lacks (the count and the vertex phase stand: the
    // caller fetches the deeper halo, calls cuberille_reproject_escaped and comes back)
    uint64_t nEsc = c->tot.nEscaped;
    if (!c->escapeChecked && (rc = cuberille_escaped_count(c, &nEsc)) != CUBERILLE_OK) return rc;
    if (nEsc)
      return fail(c, CUBERILLE_ERR_HALO, std::to_string(nEsc) + " walks left the thin halo: cuberille_reproject_escaped "
                                         "with the deeper buffer comes before cuberille_emit");
  }
  const u64 nV = c->tot.totV;
  const u64 nQ = c->tot.totQ - c->tot.Q0;
  const size_t planeCorners = needPlane ? (size_t)(c->g.nx + 1) * (c->g.ny + 1) : 0;   // positions of the rank below's vertices
  Workspace &w = c->w;
  hipStream_t s = c->stream;
  // the vertex phase was started ahead of this call (cuberille_emit_points): the device may have idled since, waiting
  // for the host's all-gather -- the cell phase is timed as an interval of its own
  if (c->pointsStartedEarly && !c->lightTiming && (!c->oneCall || c->stagesTimed)) HIP_TRY(c, hipEventRecord(c->ev[3], s));
  if (planeCorners)
    HIP_TRY(c, hipMemcpyAsync(w.points + 3 * nV, c->extPts, planeCorners * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  HIP_TRY(c, launch_emit_cells(w, c->g, c->prm.triangles, c->prm.q1, point_id_offset, nQ, needPlane ? c->extIds : nullptr,
                               nullptr, 0, 0, 0, s));
  HIP_TRY(c, hipEventRecord(c->ev[7], s));
  HIP_TRY(c, hipMemcpyAsync(c->hostTotals, w.totals, sizeof(Totals), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  return finish_result(c, res);
}

}  // extern "C"

namespace {

// Everything of a step behind the sweep, launched without waiting: count + scan (+ gate), and the part of the emit that
// needs no id offset.  The slab's bit volume and slice occupancy are complete on the stream when this is called.
int step_launch(cuberille_ctx *c, const void **dev_row, size_t *row_bytes) {
  int rc;
  // blind launches need: the sizes of a previous extraction on this context, the default projection branch and every
  // scratch table (the vertex-word queue is set up by count_prepare; the others are checked below)
  const bool blind = c->haveHistory && c->w.vqueue && !c->tune.no_cmap && !c->tune.no_heads && c->tune.points_variant == 3 &&
                     (!c->prm.project || (c->prm.variant == CUBERILLE_PROJECT_DEFAULT && c->prm.gradVariant == 0 && !c->holdGradient)) &&
                     c->histV + c->histV / 4 < 0xfffff000ULL;
  if (blind) {
    Gate gate{};
    gate.on = 1;
    gate.coverV = c->histV + c->histV / 4 + 4096;
    gate.coverQ = c->histQ + c->histQ / 4 + 4096;
    const u64 vw = (u64)c->histVW + c->histVW / 4 + 1024;
    gate.coverVW = (u32)(vw < c->nwords ? vw : c->nwords);
    rc = count_launch(c, gate);
    if (rc) return rc;
    rc = emit_points_phase(c, true, gate.coverV, gate.coverQ, gate.coverVW);
    if (rc == CUBERILLE_OK) {
      c->stepMode = 1;
    } else {
      // a table could not be had: the count is in flight all the same, read it and go on by the exact sizes
      c->pointsEmitted = false;
      rc = totals_to_host(c);
      if (rc) return rc;
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      adopt_totals(c, nullptr, nullptr);
      // (Totals::go was set by the gate: a count that fits it may have let nothing run, the exact launches below do not ask)
    }
  } else {
    rc = count_finish(c, nullptr, nullptr);
    if (rc) return rc;
  }
  if (c->stepMode != 1) {
    // sized by a host read (the first extraction on a context, a fallback configuration): the vertex phase unless a
    // quirk-Q1 flag says that the counts may still change
    // (a source slice in this slab's own halo -- aliasMustResolve -- means a recount for certain; "nothing in my buffer
    //  below my first occupied slice" is an assumption the count has made already and the rows of the ranks below usually
    //  confirm: the vertex phase runs on it, the cell pass decides from the rows -- row_flags)
    c->stepMode = 2;
    if (!c->aliasMustResolve) {
      rc = emit_points_phase(c);
      if (rc) return rc;
    }
  }
  if (c->pointsEmitted) {
    if (!c->stagesTimed && !c->lightTiming && !c->oneCall) HIP_TRY(c, hipEventRecord(c->ev[6], c->stream));
    c->pointsStartedEarly = true;
  }
  *dev_row = c->w.totals;
  *row_bytes = sizeof(Totals);
  return CUBERILLE_OK;
}

}  // namespace

extern "C" {

// ---- one step without a host round trip between count and emit (the multi-GPU steady state) -------------------------
int cuberille_step_begin(cuberille_ctx *c, const cuberille_image_desc *img, const void *dev_voxels, const cuberille_params *prm,
                         const cuberille_slab *slab, const void **dev_row, size_t *row_bytes) {
  int rc = validate(c, img, dev_voxels, prm);
  if (rc) return rc;
  if (!dev_row || !row_bytes) return fail(c, CUBERILLE_ERR_ARGUMENT, "null row pointer");
  rc = count_prepare(c, img, dev_voxels, prm, slab);
  if (rc) return rc;
  rc = classify_slab(c, img, slab);
  if (rc) return rc;
  return step_launch(c, dev_row, row_bytes);
}

int cuberille_step_classify(cuberille_ctx *c, const cuberille_image_desc *img, const void *dev_voxels, const cuberille_params *prm,
                            const cuberille_slab *slab, uint64_t **dev_bits, size_t *words_per_slice) {
  int rc = validate(c, img, dev_voxels, prm);
  if (rc) return rc;
  if (!dev_bits || !words_per_slice) return fail(c, CUBERILLE_ERR_ARGUMENT, "null bit-plane pointer");
  rc = count_prepare(c, img, dev_voxels, prm, slab);
  if (rc) return rc;
  if (slab && slab->voxels_ready_event) HIP_TRY(c, hipStreamWaitEvent(c->stream, (hipEvent_t)slab->voxels_ready_event, 0));
  HIP_TRY(c, launch_classify(img->pixel_type, c->w, c->g, c->prm, c->g.oz0, c->g.oz1, c->tune, c->stream));
  c->stepMode = 3;
  *dev_bits = (uint64_t *)c->bits.p;
  *words_per_slice = (size_t)c->g.ny * c->g.W;
  return CUBERILLE_OK;
}

int cuberille_step_count(cuberille_ctx *c, void *halo_bits_event, void *halo_voxels_event, const void **dev_row, size_t *row_bytes) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  if (!dev_row || !row_bytes) return fail(c, CUBERILLE_ERR_ARGUMENT, "null row pointer");
  if (c->stepMode != 3) return fail(c, CUBERILLE_ERR_STATE, "cuberille_step_count follows cuberille_step_classify");
  HIP_TRY(c, hipSetDevice(c->device));
  c->stepMode = 0;
  if (halo_bits_event) HIP_TRY(c, hipStreamWaitEvent(c->stream, (hipEvent_t)halo_bits_event, 0));
  // which of the halo slices hold an inside voxel (quirk Q1 looks at it): from the planes that came in
  HIP_TRY(c, launch_occupancy_range(c->w, c->g, 0, c->g.oz0, c->stream));
  HIP_TRY(c, launch_occupancy_range(c->w, c->g, c->g.oz1, c->g.nzb, c->stream));
  c->voxelHaloEvent = (hipEvent_t)halo_voxels_event;
  const int rc = step_launch(c, dev_row, row_bytes);
  if (rc) c->voxelHaloEvent = nullptr;
  return rc;
}

}  // extern "C"

namespace {

// base: added to the offset summed from the rows (cuberille_extract_device on a slab: the caller's point_id_offset)
int step_end_impl(cuberille_ctx *c, const void *dev_rows, int n_ranks, int rank, u64 base, cuberille_result *res) {
  if (!c || !dev_rows || n_ranks < 1 || rank < 0 || rank >= n_ranks) return c ? fail(c, CUBERILLE_ERR_ARGUMENT, "bad rows or rank") : CUBERILLE_ERR_ARGUMENT;
  if (c->stepMode == 0) return fail(c, CUBERILLE_ERR_STATE, "cuberille_step_end follows cuberille_step_begin");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const bool blind = c->stepMode == 1;
  if (c->hostRowsCap < (size_t)n_ranks) {
    if (c->hostRows) (void)hipHostFree(c->hostRows);
    c->hostRows = nullptr;
    c->hostRowsCap = 0;
    HIP_TRY(c, hipHostMalloc((void **)&c->hostRows, (size_t)n_ranks * sizeof(Totals), hipHostMallocDefault));
    c->hostRowsCap = (size_t)n_ranks;
  }
  if (!c->lightTiming && (!c->oneCall || c->stagesTimed)) HIP_TRY(c, hipEventRecord(c->ev[3], s));
  // the cells, unless a flag stands somewhere (the kernel looks at the rows itself: every rank decides alike).  Sized
  // by the cover values (blind) or by this rank's counts; a rank whose vertex phase did not run launches nothing.
  if (blind || c->pointsEmitted) {
    const u64 nQ = blind ? c->histQ + c->histQ / 4 + 4096 : c->tot.totQ - c->tot.Q0;
    HIP_TRY(c, launch_emit_cells(c->w, c->g, c->prm.triangles, c->prm.q1, base, nQ, nullptr, (const Totals *)dev_rows, n_ranks, rank,
                                 blind ? 1 : 0, s));
  }
  HIP_TRY(c, hipEventRecord(c->ev[7], s));
  {
    const int rc = totals_to_host(c);
    if (rc) return rc;
  }
  HIP_TRY(c, hipMemcpyAsync(c->hostRows, dev_rows, (size_t)n_ranks * sizeof(Totals), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));                  // the one wait of the step
  u32 flags = 0;
  u64 off = base;
  for (int r = 0; r < n_ranks; r++) {
    flags |= row_flags(c->hostRows, r);                      // (the rule the cell pass applied on the device)
    if (r < rank) off += c->hostRows[r].totV - c->hostRows[r].V0;
  }
  const u32 mine = c->hostTotals->err;
  if (blind) {
    const bool ran = c->hostTotals->go != 0;
    adopt_totals(c, nullptr, nullptr);
    c->pointsEmitted = ran;                             // (a gate that said no: nothing ran, nothing is valid)
    if (!ran) c->pointsStartedEarly = false;
  } else {
    c->tot.nEscaped = c->hostTotals->nEscaped;
    c->tot.err = c->hostTotals->err;
  }
  c->escapeChecked = c->pointsEmitted;
  c->stepMode = 0;
  if (flags) {
    // nothing was written; the count stands and the synchronous calls take over from it (slab_info, recount,
    // emit_points, escaped_count / reproject_escaped, emit)
    (void)mine;
    c->err = "cuberille_step_end: a flag stands on some rank (quirk Q1 across slabs, capacity, or a walk left a thin halo)";
    if (res) *res = c->res;                             // the counts (what cuberille_count would have returned)
    return CUBERILLE_RETRY;
  }
  c->slabMesh = off != 0 || c->tot.V0 != 0 || part_of_a_volume(c->g);
  c->pointOffset = off;
  return finish_result(c, res);
}

}  // namespace

extern "C" {

int cuberille_step_end(cuberille_ctx *c, const void *dev_rows, int n_ranks, int rank, cuberille_result *res) {
  return step_end_impl(c, dev_rows, n_ranks, rank, 0, res);
}

int cuberille_failed_row(void *host_row, size_t capacity, size_t *row_bytes) {
  if (row_bytes) *row_bytes = sizeof(Totals);
  if (!host_row || capacity < sizeof(Totals)) return CUBERILLE_ERR_ARGUMENT;
  Totals t{};
  t.aliasZ = t.topZ = t.top2Z = -1;
  t.err = ERRF_RANK_FAILED;        // counts of zero: the ranks above add nothing to their offsets, and write nothing anyway
  std::memcpy(host_row, &t, sizeof t);
  return CUBERILLE_OK;
}

int cuberille_extract_device(cuberille_ctx *c, const cuberille_image_desc *img, const void *dev_voxels,
                             const cuberille_params *prm, const cuberille_slab *slab, cuberille_result *res) {
  // the one-wait step with this context as the only rank: the first extraction on a context reads its counts back
  // before it sizes the emit, the following ones launch everything blindly from the sizes of the one before and wait
  // once (every volume the reference ships is in the regime where the waits ARE the extraction time)
  const void *row = nullptr;
  size_t rowBytes = 0;
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  struct OneCall {
    cuberille_ctx *c;
    explicit OneCall(cuberille_ctx *ctx) : c(ctx) { c->oneCall = true; }
    ~OneCall() { c->oneCall = false; }
  } guard(c);
  int rc = cuberille_step_begin(c, img, dev_voxels, prm, slab, &row, &rowBytes);
  if (rc) return rc;
  rc = step_end_impl(c, row, 1, 0, slab ? slab->point_id_offset : 0, res);
  if (rc != CUBERILLE_RETRY) return rc;
  // counts beyond the guess, or a slab that needs its neighbours (quirk Q1, an escaped walk): the count stands
  return cuberille_emit(c, slab ? slab->point_id_offset : 0, res);
}

}  // extern "C"

namespace {

// The pinned staging ring of the chunked copies (two slots of `bytes`, their events, the copy stream), (re)made in one
// place: the recorded size only ever describes two live slots.
int ensure_staging(cuberille_ctx *c, size_t bytes) {
  if (!c->copyStream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
  for (int i = 0; i < 2; i++) {
    if (!c->stageFree[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->stageFree[i], hipEventDisableTiming));
    if (!c->chunkIn[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->chunkIn[i], hipEventDisableTiming));
  }
  if (c->stageBytes >= bytes && c->stage[0] && c->stage[1]) return CUBERILLE_OK;
  c->stageBytes = 0;
  for (int i = 0; i < 2; i++)
    if (c->stage[i]) { (void)hipHostFree(c->stage[i]); c->stage[i] = nullptr; }
  for (int i = 0; i < 2; i++) {
    if (g_fail_alloc_countdown >= 0 && g_fail_alloc_countdown-- == 0)
      return fail(c, CUBERILLE_ERR_HIP, "hipHostMalloc(staging slot): out of memory (failure drill)");
    HIP_TRY(c, hipHostMalloc(&c->stage[i], bytes, hipHostMallocDefault));
  }
  c->stageBytes = bytes;
  return CUBERILLE_OK;
}

// Host threads that copy pageable caller memory into the pinned staging ring, one fixed share of every chunk
// each (a single memcpy stream cannot feed a PCIe Gen5 link; a handful can).
struct StagePool {
  std::vector<std::thread> workers;
  std::atomic<long long> freeUpTo{-1};      // chunks whose staging slot may be overwritten: all i <= freeUpTo
  std::vector<std::atomic<int>> done;       // per chunk: workers that have copied their share
  explicit StagePool(size_t nchunks) : done(nchunks) { for (auto &d : done) d.store(0); }
};

}  // namespace

extern "C" {

int cuberille_extract_host(cuberille_ctx *c, const cuberille_image_desc *img, const void *host_voxels,
                           const cuberille_params *prm, cuberille_result *res) {
  int rc = validate(c, img, host_voxels, prm);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t psz = pixel_size(img->pixel_type);
  const size_t sliceBytes = (size_t)img->dims[0] * img->dims[1] * psz;
  const size_t nz = (size_t)img->dims[2];
  const size_t bytes = sliceBytes * nz;
  HIP_TRY(c, c->voxOwn.reserve(bytes));
  // below a GiB: one plain copy (the runtime stages pageable memory itself, at link rate once the copy is large; the
  // chunk pipeline below needs some tens of chunks to amortise its start -- measured 34 ms against 11 ms at 512^3 f32);
  // the extraction follows on the stream
  const size_t kChunk = 32u << 20;
  if (bytes < (1ull << 30) || sliceBytes > kChunk) {
    HIP_TRY(c, hipMemcpyAsync(c->voxOwn.p, host_voxels, bytes, hipMemcpyHostToDevice, c->stream));
    return cuberille_extract_device(c, img, c->voxOwn.p, prm, nullptr, res);
  }
  // large volumes: z-chunks through a pinned double buffer on a copy stream; chunk i is thresholded on the
  // context's stream while chunk i+1 crosses the link and the host threads stage chunk i+2
  rc = ensure_staging(c, kChunk);
  if (rc) return rc;
  rc = count_prepare(c, img, c->voxOwn.p, prm, nullptr);
  if (rc) return rc;
  const size_t slicesPerChunk = kChunk / sliceBytes;
  const size_t nchunks = (nz + slicesPerChunk - 1) / slicesPerChunk;
  unsigned hw = std::thread::hardware_concurrency();
  const int nT = (int)(hw >= 16 ? 8 : (hw >= 4 ? hw / 2 : 1));
  StagePool pool(nchunks);
  const char *src = (const char *)host_voxels;
  auto chunkRange = [&](size_t i, size_t &z0, size_t &z1) { z0 = i * slicesPerChunk; z1 = z0 + slicesPerChunk < nz ? z0 + slicesPerChunk : nz; };
  std::atomic<bool> abort{false};
  for (int t = 0; t < nT; t++) {
    pool.workers.emplace_back([&, t] {
      for (size_t i = 0; i < nchunks && !abort.load(std::memory_order_relaxed); i++) {
        while (pool.freeUpTo.load(std::memory_order_acquire) < (long long)i) {
          if (abort.load(std::memory_order_relaxed)) return;
          std::this_thread::yield();
        }
        size_t z0, z1;
        chunkRange(i, z0, z1);
        const size_t cb = (z1 - z0) * sliceBytes;
        const size_t a = cb * t / nT & ~(size_t)63, b = (t == nT - 1) ? cb : (cb * (t + 1) / nT & ~(size_t)63);
        std::memcpy((char *)c->stage[i & 1] + a, src + z0 * sliceBytes + a, b - a);
        pool.done[i].fetch_add(1, std::memory_order_release);
      }
    });
  }
  auto joinAll = [&] { for (auto &w : pool.workers) if (w.joinable()) w.join(); };
  hipError_t e = hipSuccess;
  pool.freeUpTo.store(1, std::memory_order_release);          // both slots start free
  for (size_t i = 0; i < nchunks && e == hipSuccess; i++) {
    while (pool.done[i].load(std::memory_order_acquire) < nT) std::this_thread::yield();
    size_t z0, z1;
    chunkRange(i, z0, z1);
    e = hipMemcpyAsync((char *)c->voxOwn.p + z0 * sliceBytes, c->stage[i & 1], (z1 - z0) * sliceBytes, hipMemcpyHostToDevice,
                       c->copyStream);
    if (e == hipSuccess) e = hipEventRecord(c->chunkIn[i & 1], c->copyStream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->chunkIn[i & 1], 0);
    if (e == hipSuccess) e = launch_classify(img->pixel_type, c->w, c->g, c->prm, (int)z0, (int)z1, c->tune, c->stream);
    // slot (i & 1) is free for chunk i + 2 once this chunk has crossed the link
    if (e == hipSuccess && i + 2 < nchunks) {
      e = hipEventSynchronize(c->chunkIn[i & 1]);
      pool.freeUpTo.store((long long)i + 2, std::memory_order_release);
    }
  }
  if (e != hipSuccess) abort.store(true);
  joinAll();
  if (e != hipSuccess) return fail(c, CUBERILLE_ERR_HIP, std::string("overlapped upload: ") + hipGetErrorString(e));
  uint64_t np = 0, nc = 0;
  rc = count_finish(c, &np, &nc);
  if (rc) return rc;
  return cuberille_emit(c, 0, res);
}

int cuberille_extract_stream(cuberille_ctx *c, const cuberille_image_desc *img, cuberille_chunk_source source, void *user,
                             const cuberille_params *prm, cuberille_result *res) {
  if (c && !source) return fail(c, CUBERILLE_ERR_ARGUMENT, "null chunk source");
  int rc = validate(c, img, (const void *)source, prm);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t psz = pixel_size(img->pixel_type);
  const size_t sliceBytes = (size_t)img->dims[0] * img->dims[1] * psz;
  const size_t nz = (size_t)img->dims[2];
  HIP_TRY(c, c->voxOwn.reserve(sliceBytes * nz));
  // chunks of about 32 MiB, whole slices, at least one
  const size_t slicesPerChunk = sliceBytes >= (32u << 20) ? 1 : (32u << 20) / sliceBytes;
  const size_t chunkBytes = slicesPerChunk * sliceBytes;
  const size_t nchunks = (nz + slicesPerChunk - 1) / slicesPerChunk;
  rc = ensure_staging(c, chunkBytes);
  if (rc) return rc;
  rc = count_prepare(c, img, c->voxOwn.p, prm, nullptr);
  if (rc) return rc;
  hipError_t e = hipSuccess;
  int gaveUp = 0;
  for (size_t i = 0; i < nchunks && e == hipSuccess; i++) {
    const size_t z0 = i * slicesPerChunk, z1 = z0 + slicesPerChunk < nz ? z0 + slicesPerChunk : nz;
    // the slot is free once chunk i - 2 has crossed the link
    if (i >= 2) e = hipEventSynchronize(c->chunkIn[i & 1]);
    if (e != hipSuccess) break;
    gaveUp = source(user, c->stage[i & 1], (int64_t)z0, (int64_t)z1);
    if (gaveUp) break;
    e = hipMemcpyAsync((char *)c->voxOwn.p + z0 * sliceBytes, c->stage[i & 1], (z1 - z0) * sliceBytes, hipMemcpyHostToDevice,
                       c->copyStream);
    if (e == hipSuccess) e = hipEventRecord(c->chunkIn[i & 1], c->copyStream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->chunkIn[i & 1], 0);
    if (e == hipSuccess) e = launch_classify(img->pixel_type, c->w, c->g, c->prm, (int)z0, (int)z1, c->tune, c->stream);
  }
  if (gaveUp || e != hipSuccess) {
    // let what is in flight finish before the staging slots are used again
    (void)hipStreamSynchronize(c->copyStream);
    (void)hipStreamSynchronize(c->stream);
    if (gaveUp) return fail(c, CUBERILLE_ERR_SOURCE, "the chunk source gave up with status " + std::to_string(gaveUp));
    return fail(c, CUBERILLE_ERR_HIP, std::string("streamed upload: ") + hipGetErrorString(e));
  }
  uint64_t np = 0, nc = 0;
  rc = count_finish(c, &np, &nc);
  if (rc) return rc;
  return cuberille_emit(c, 0, res);
}

int cuberille_warm_up(cuberille_ctx *c, const cuberille_image_desc *img, const cuberille_params *prm) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  cuberille_params dflt{};
  dflt.iso_value = 1.0; dflt.generate_triangles = 1; dflt.project_vertices = 1; dflt.distance_threshold = 0.5;
  dflt.step_length = -1.0; dflt.relaxation = 0.95; dflt.max_steps = 50; dflt.emulate_empty_slice_aliasing = 1;
  dflt.iso_value_int = 1;
  // A count, a mesh or an open step on the context: its workspace is in use -- the toy extraction would replace the state,
  // and DevBuf::reserve frees before it allocates, which would leave the workspace pointers of the live count (c->w, the
  // bit volume) dangling under cuberille_emit / _recount / _slice_bits_device / _alias_plane_device / _debug_bits.
  // Such a context has run kernels already; the next extraction grows what it needs itself.
  const bool live = c->counted || c->haveMesh || c->stepMode != 0;
  if (live) c->warm = true;
  if (!c->warm) {
    // one tiny extraction: the first launch of any kernel loads the library's code objects, the first copies and events
    // set up the runtime's queues.  8 x 8 x 8 voxels with a 4 x 4 x 4 block inside; nothing of it stays on the context.
    unsigned char tiny[512];
    for (int z = 0; z < 8; z++)
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) tiny[(z * 8 + y) * 8 + x] = (x >= 2 && x < 6 && y >= 2 && y < 6 && z >= 2 && z < 6) ? 200 : 0;
    cuberille_image_desc d{};
    d.pixel_type = CUBERILLE_PIX_U8;
    for (int i = 0; i < 3; i++) { d.dims[i] = 8; d.spacing[i] = 1.0; d.direction[i * 4] = 1.0; }
    cuberille_params p = dflt;
    p.iso_value = 100.0;
    cuberille_result r{};
    for (int i = 0; i < 2; i++) {            // twice: the second one takes the blind launches of the one-wait step
      const int rc = cuberille_extract_host(c, &d, tiny, &p, &r);
      if (rc) return rc;
    }
    // the runtime sets up its staging for copies from and to PAGEABLE memory at the first copy that needs it (measured
    // through the reference's driver: 7.2 ms inside the first hipMemcpyAsync of nucleon.mha's 69 KB, profiles/
    // r4_cold_update.log): one round trip of a size that takes its staging buffers, one of a size it pins in place
    {
      std::vector<char> host(8u << 20, 1);
      HIP_TRY(c, c->voxOwn.reserve(host.size()));
      const size_t sizes[2] = {256u << 10, host.size()};
      for (size_t n : sizes) {
        HIP_TRY(c, hipMemcpyAsync(c->voxOwn.p, host.data(), n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(host.data(), c->voxOwn.p, n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
      }
    }
    c->haveHistory = false;                  // (sizes of a toy volume: the first real extraction reads its own counts)
    c->haveMesh = false;
    c->counted = false;
    c->warm = true;
  }
  if (!img) return CUBERILLE_OK;
  if (pixel_size(img->pixel_type) == 0) return fail(c, CUBERILLE_ERR_ARGUMENT, "unknown pixel type");
  for (int i = 0; i < 3; i++)
    if (img->dims[i] < 1 || img->dims[i] > 0x7fffffffLL) return fail(c, CUBERILLE_ERR_ARGUMENT, "image dimensions out of range");
  (void)prm;
  if (live) return CUBERILLE_OK;
  // the buffers whose size follows from the description (count_prepare, emit_points_phase, cuberille_extract_host); a
  // reservation that fails here is asked for again, and reported, by the extraction
  const size_t nx = (size_t)img->dims[0], ny = (size_t)img->dims[1], nz = (size_t)img->dims[2];
  const size_t W = (nx + 63) / 64, nwords = nz * ny * W, nseg = (nwords + 63) / 64, nblk = (nwords + COUNT_WB - 1) / COUNT_WB;
  const size_t bytes = nx * ny * nz * pixel_size(img->pixel_type);
  bool ok = c->voxOwn.reserve(bytes) == hipSuccess;
  ok = ok && c->bits.reserve((nwords + ny * W) * sizeof(u64)) == hipSuccess;
  ok = ok && c->occ.reserve(sizeof(Totals) + nz * sizeof(u32)) == hipSuccess;
  ok = ok && c->prefix.reserve((nwords + 4) * sizeof(u32)) == hipSuccess;
  ok = ok && c->segPre.reserve(nseg * sizeof(u64)) == hipSuccess;
  ok = ok && c->blockTot.reserve((nblk + 2 * (nblk / 8192 + 1)) * sizeof(u64)) == hipSuccess;
  ok = ok && c->blockBase.reserve(nblk * 2 * sizeof(u64)) == hipSuccess;
  if (ok && nx % 64 != 0) ok = c->flatBits.reserve((nwords + 32) * sizeof(u64)) == hipSuccess;
  if (ok && nwords < 0xffffffffULL && !c->tune.no_vqueue) ok = c->vqueue.reserve(nwords * sizeof(u32)) == hipSuccess;
  if (ok && !c->tune.no_cmap) {
    const size_t mapBytes = c->tune.cmap_linear ? (nx + 1) * (ny + 1) * (nz + 2) * sizeof(u32)
                          : ((nx + 4) >> 2) * ((ny + 4) >> 2) * ((nz + 3) >> 1) * 32 * sizeof(u32);
    ok = c->cmap.reserve(mapBytes) == hipSuccess;
  }
  if (!ok) (void)hipGetLastError();
  // the pinned staging ring: a chunked upload (volumes of a GiB and more) and the download of a mesh of more than 128 MiB
  // go through it; a volume of 64 MiB can carry such a mesh
  if (ok && bytes >= (64ull << 20)) (void)ensure_staging(c, 32u << 20);
  return CUBERILLE_OK;
}

int cuberille_slice_counts(cuberille_ctx *c, uint64_t *points, uint64_t *quads, size_t n_slices) {
  if (!c || (!points && !quads)) return CUBERILLE_ERR_ARGUMENT;
  if (!c->counted && !c->haveMesh) return fail(c, CUBERILLE_ERR_STATE, "no count on this context");
  const int own = c->g.oz1 - c->g.oz0, counted = c->g.oz1 - c->g.cz0;
  if (n_slices != (size_t)own) return fail(c, CUBERILLE_ERR_ARGUMENT, "one entry per owned slice, please");
  HIP_TRY(c, hipSetDevice(c->device));
  // (scratch: the cell buffer is free between a count and its emit, but a mesh may be in it; a small buffer of its own)
  DevBuf tmp;
  HIP_TRY(c, tmp.reserve((size_t)(counted + 1) * 2 * sizeof(u64)));
  std::vector<u64> host((size_t)(counted + 1) * 2);
  hipError_t e = launch_slice_prefix(c->w, c->g, (u64 *)tmp.p, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(host.data(), tmp.p, host.size() * sizeof(u64), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  tmp.release();
  if (e != hipSuccess) return fail(c, CUBERILLE_ERR_HIP, std::string("slice counts: ") + hipGetErrorString(e));
  const int skip = c->g.oz0 - c->g.cz0;                       // the ghost slice
  for (int z = 0; z < own; z++) {
    if (points) points[z] = host[2 * (size_t)(z + skip + 1)] - host[2 * (size_t)(z + skip)];
    if (quads) quads[z] = host[2 * (size_t)(z + skip + 1) + 1] - host[2 * (size_t)(z + skip) + 1];
  }
  return CUBERILLE_OK;
}

int cuberille_slab_info(cuberille_ctx *c, cuberille_slab_status *out) {
  if (!c || !out) return CUBERILLE_ERR_ARGUMENT;
  if ((!c->counted && !c->haveMesh) || !c->slabMode || !c->hostOcc)
    return fail(c, CUBERILLE_ERR_STATE, "no counted slab on this context");
  out->alias_source_below_buffer = c->aliasBelowBuffer ? 1 : 0;
  out->reserved = 0;
  out->lowest_occupied_z = out->highest_occupied_z = out->second_highest_occupied_z = -1;
  for (int z = c->g.oz0; z < c->g.oz1; z++)
    if (c->hostOcc[(size_t)z]) {
      if (out->lowest_occupied_z < 0) out->lowest_occupied_z = c->g.zglob0 + z;
      out->second_highest_occupied_z = out->highest_occupied_z;
      out->highest_occupied_z = c->g.zglob0 + z;
    }
  out->alias_z = c->aliasZ >= 0 ? c->g.zglob0 + c->aliasZ : -1;
  return CUBERILLE_OK;
}

static bool set_opt(Tuning &t, const char *name, long long v) {
#define OPT(field) if (!std::strcmp(name, #field)) { t.field = (int)v; return true; }
  OPT(no_cmap) OPT(no_heads) OPT(no_vqueue) OPT(no_stream_classify) OPT(classify_variant) OPT(classify_grid)
  OPT(points_variant) OPT(points_no_split) OPT(count_variant) OPT(cmap_linear) OPT(proj_chunk) OPT(proj_waves) OPT(proj_refill) OPT(proj_xcd) OPT(proj_literal) OPT(stage_timing) OPT(classify_keep_tail) OPT(proj_chunk64_below) OPT(points_split) OPT(count_no_fold) OPT(proj_short) OPT(proj_ident)
#undef OPT
  return false;
}

int cuberille_debug_h2d_seconds(cuberille_ctx *c, size_t bytes, double *seconds) {
  if (!c || !seconds || !bytes) return CUBERILLE_ERR_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t piece = 64u << 20;
  void *host = nullptr, *dev = nullptr;
  HIP_TRY(c, hipHostMalloc(&host, piece, hipHostMallocDefault));
  if (hipMalloc(&dev, piece) != hipSuccess) { (void)hipHostFree(host); return fail(c, CUBERILLE_ERR_HIP, "hipMalloc failed"); }
  std::memset(host, 1, piece);
  hipEvent_t a = nullptr, b = nullptr;
  hipError_t e = hipEventCreate(&a);
  if (e == hipSuccess) e = hipEventCreate(&b);
  if (e == hipSuccess) e = hipMemcpyAsync(dev, host, piece, hipMemcpyHostToDevice, c->stream);   // warm-up
  if (e == hipSuccess) e = hipEventRecord(a, c->stream);
  for (size_t done = 0; done < bytes && e == hipSuccess; done += piece)
    e = hipMemcpyAsync(dev, host, bytes - done < piece ? bytes - done : piece, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipEventRecord(b, c->stream);
  if (e == hipSuccess) e = hipEventSynchronize(b);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  (void)hipFree(dev);
  (void)hipHostFree(host);
  if (e != hipSuccess) return fail(c, CUBERILLE_ERR_HIP, std::string("link measurement: ") + hipGetErrorString(e));
  *seconds = 1e-3 * ms;
  return CUBERILLE_OK;
}

int cuberille_debug_set_option(cuberille_ctx *c, const char *name, int64_t value) {
  if (!c || !name) return CUBERILLE_ERR_ARGUMENT;
  if (!std::strcmp(name, "defaults")) { c->tune = Tuning(); g_fail_alloc_countdown = -1; return CUBERILLE_OK; }
  if (!std::strcmp(name, "fail_alloc_at")) { g_fail_alloc_countdown = (long long)value; return CUBERILLE_OK; }
  if (!set_opt(c->tune, name, (long long)value)) return fail(c, CUBERILLE_ERR_ARGUMENT, std::string("unknown option ") + name);
  return CUBERILLE_OK;
}

int cuberille_mesh_device(const cuberille_ctx *c, const float **d_points, const uint64_t **d_cells) {
  if (!c || !c->haveMesh) return CUBERILLE_ERR_STATE;
  if (d_points) *d_points = (const float *)c->points.p + 3 * c->tot.V0;
  if (d_cells) *d_cells = (const uint64_t *)c->cells.p;
  return CUBERILLE_OK;
}

}  // extern "C"

namespace {

// Device -> pageable host memory: 32 MiB chunks land in the pinned staging slots on the copy stream while a few host
// threads move the previous chunk to its destination.  What this buys is the FIRST touch of a freshly allocated
// destination (the usual case: one result array per extraction), which the page faults bound: 668 MB in 48 ms against
// 65 ms for one plain hipMemcpy on the same box; into memory that has been touched before both run at the link's
// 52-55 GB/s (12.9 against 12.2 ms).  Small copies take the plain way.
int download_pipelined(cuberille_ctx *c, void *dst, const void *src, size_t bytes) {
  const size_t kChunk = 32u << 20;
  if (bytes < 4 * kChunk) {
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CUBERILLE_OK;
  }
  {
    const int rc = ensure_staging(c, kChunk);
    if (rc) return rc;
  }
  // the mesh was written on the context's stream
  HIP_TRY(c, hipEventRecord(c->stageFree[0], c->stream));
  HIP_TRY(c, hipStreamWaitEvent(c->copyStream, c->stageFree[0], 0));
  const size_t nchunks = (bytes + kChunk - 1) / kChunk;
  unsigned hw = std::thread::hardware_concurrency();
  const int nT = (int)(hw >= 16 ? 8 : (hw >= 4 ? hw / 2 : 1));
  std::vector<std::atomic<int>> landed(nchunks), moved(nchunks);
  for (size_t i = 0; i < nchunks; i++) { landed[i].store(0); moved[i].store(0); }
  std::atomic<bool> abort{false};
  std::vector<std::thread> workers;
  for (int t = 0; t < nT; t++) {
    workers.emplace_back([&, t] {
      for (size_t i = 0; i < nchunks; i++) {
        while (!landed[i].load(std::memory_order_acquire)) {
          if (abort.load(std::memory_order_relaxed)) return;
          std::this_thread::yield();
        }
        const size_t off = i * kChunk, cb = bytes - off < kChunk ? bytes - off : kChunk;
        const size_t a = cb * t / nT & ~(size_t)63, b = (t == nT - 1) ? cb : (cb * (t + 1) / nT & ~(size_t)63);
        std::memcpy((char *)dst + off + a, (const char *)c->stage[i & 1] + a, b - a);
        moved[i].fetch_add(1, std::memory_order_release);
      }
    });
  }
  auto issue = [&](size_t i) -> hipError_t {
    const size_t off = i * kChunk, cb = bytes - off < kChunk ? bytes - off : kChunk;
    hipError_t e = hipMemcpyAsync(c->stage[i & 1], (const char *)src + off, cb, hipMemcpyDeviceToHost, c->copyStream);
    if (e == hipSuccess) e = hipEventRecord(c->chunkIn[i & 1], c->copyStream);
    return e;
  };
  hipError_t e = issue(0);
  for (size_t i = 0; i < nchunks && e == hipSuccess; i++) {
    if (i + 1 < nchunks) {
      // slot (i + 1) & 1 held chunk i - 1: the host threads must be done with it
      if (i >= 1) while (moved[i - 1].load(std::memory_order_acquire) < nT) std::this_thread::yield();
      e = issue(i + 1);
      if (e != hipSuccess) break;
    }
    e = hipEventSynchronize(c->chunkIn[i & 1]);
    if (e == hipSuccess) landed[i].store(1, std::memory_order_release);
  }
  if (e != hipSuccess) abort.store(true);
  for (auto &w : workers) w.join();
  if (e != hipSuccess) {
    (void)hipStreamSynchronize(c->copyStream);
    return fail(c, CUBERILLE_ERR_HIP, std::string("mesh download: ") + hipGetErrorString(e));
  }
  return CUBERILLE_OK;
}

}  // namespace

extern "C" {

int cuberille_mesh_download(cuberille_ctx *c, float *points, uint64_t *cells) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  if (!c->haveMesh) return fail(c, CUBERILLE_ERR_STATE, "no mesh: call cuberille_extract_* or cuberille_emit first");
  HIP_TRY(c, hipSetDevice(c->device));
  const cuberille_result &r = c->res;
  if (points && r.n_points) {
    const int rc = download_pipelined(c, points, (const float *)c->points.p + 3 * c->tot.V0, r.n_points * 3 * sizeof(float));
    if (rc) return rc;
  }
  if (cells && r.n_cells) {
    const int rc = download_pipelined(c, cells, c->cells.p, r.n_cells * r.verts_per_cell * sizeof(uint64_t));
    if (rc) return rc;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CUBERILLE_OK;
}

int cuberille_mesh_host(cuberille_ctx *c, float **points, uint64_t **cells) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  if (!c->haveMesh) return fail(c, CUBERILLE_ERR_STATE, "no mesh: call cuberille_extract_* or cuberille_emit first");
  const cuberille_result &r = c->res;
  if (!c->hostMeshValid) {
    const size_t pb = (size_t)r.n_points * 3 * sizeof(float), cb = (size_t)r.n_cells * r.verts_per_cell * sizeof(uint64_t);
    if (!c->hostPoints.reserve(pb ? pb : 1) || !c->hostCells.reserve(cb ? cb : 1))
      return fail(c, CUBERILLE_ERR_HIP, "cuberille_mesh_host: out of host memory");
    const int rc = cuberille_mesh_download(c, (float *)c->hostPoints.p, (uint64_t *)c->hostCells.p);
    if (rc) return rc;
    c->hostMeshValid = true;
  }
  if (points) *points = (float *)c->hostPoints.p;
  if (cells) *cells = (uint64_t *)c->hostCells.p;
  return CUBERILLE_OK;
}

int cuberille_hold_gradient(cuberille_ctx *c, int hold) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  if (c->stepMode != 0) return fail(c, CUBERILLE_ERR_STATE, "a step is open on this context");
  c->holdGradient = hold != 0;
  if (!hold && c->held.img) {
    (void)hipSetDevice(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));     // (a walk may still be reading it)
    c->heldGrad.release();
    c->held = HeldGradient{};
  }
  return CUBERILLE_OK;
}

int cuberille_gradient_held(cuberille_ctx *c, int64_t dims[3]) {
  if (!c) return 0;
  if (dims) for (int i = 0; i < 3; i++) dims[i] = c->held.img ? c->held.n[i] : 0;
  return c->held.img ? 1 : 0;
}

int cuberille_release_host_mesh(cuberille_ctx *c) {
  if (!c) return CUBERILLE_ERR_ARGUMENT;
  c->hostPoints.release();
  c->hostCells.release();
  c->hostMeshValid = false;
  return CUBERILLE_OK;
}

int cuberille_mesh_write_vtk(cuberille_ctx *c, const char *path, int n_threads) {
  if (!c || !path) return CUBERILLE_ERR_ARGUMENT;
  if (!c->haveMesh) return fail(c, CUBERILLE_ERR_STATE, "no mesh: call cuberille_extract_* or cuberille_emit first");
  if (c->slabMesh) return fail(c, CUBERILLE_ERR_STATE, "a slab mesh is not self-contained: concatenate the rank buffers and call cuberille_write_vtk_buffers");
  const cuberille_result &r = c->res;
  float *pts = nullptr;
  uint64_t *cells = nullptr;
  const int rc = cuberille_mesh_host(c, &pts, &cells);
  if (rc != CUBERILLE_OK) return rc;
  const int wr = cuberille_write_vtk_buffers(path, pts, r.n_points, cells, r.n_cells, r.verts_per_cell, n_threads);
  if (wr != CUBERILLE_OK) return fail(c, wr, std::string("cannot write ") + path);
  return CUBERILLE_OK;
}

int cuberille_debug_bits(cuberille_ctx *c, uint64_t *words, size_t n_words) {
  if (!c || !words) return CUBERILLE_ERR_ARGUMENT;
  if (!c->counted && !c->haveMesh) return fail(c, CUBERILLE_ERR_STATE, "no classified volume on this context");
  const size_t have = (size_t)c->g.ny * c->g.nzb * c->g.W;
  if (n_words > have) return fail(c, CUBERILLE_ERR_ARGUMENT, "more words requested than the bit volume holds");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(words, c->bits.p, n_words * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CUBERILLE_OK;
}

int cuberille_slice_occupancy(cuberille_ctx *c, uint32_t *occupied, size_t n_slices) {
  if (!c || !occupied) return CUBERILLE_ERR_ARGUMENT;
  if (!c->counted && !c->haveMesh) return fail(c, CUBERILLE_ERR_STATE, "no classified volume on this context");
  if (n_slices > (size_t)c->g.nzb) return fail(c, CUBERILLE_ERR_ARGUMENT, "more slices requested than the buffer holds");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(occupied, c->w.sliceOcc, n_slices * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CUBERILLE_OK;
}

}  // extern "C"
