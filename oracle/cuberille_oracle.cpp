/*
 * cuberille_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see cuberille_oracle.h).
 *
 * Line-by-line restatement of /root/reference/Source/itkCuberilleImageToMeshFilter.txx
 * (GenerateData 59-216, SetVerticesFromFace 218-233, GetVertexLookupIndex 235-254,
 * AddVertex 256-276, AddQuadFace 278-332, ProjectVertexToIsoSurface default branch
 * 439-474, ComputeGradientImage 478-498) and of the lookup classes of
 * itkCuberilleImageToMeshFilter.h:243-313, WITHOUT ITK.  Every ITK call the reference
 * makes is replaced by a named function below that states the ITK 3.x behaviour this
 * build adopts ("[ITK] contract" I1..I12).  ITK is a
 * third-party dependency of the reference that is not vendored under /root/reference
 * and is not installed in this image (CMakeLists.txt:7 FIND_PACKAGE(ITK REQUIRED), no
 * version pinned; 2010-era ITK 3.16-3.20 by its API use), so these pieces restate its
 * published algorithm and cannot be checked against it here.
 *
 * Deliberately keeps the reference's data structures (two std::map lookups, a
 * whole-image float gradient image, one heap allocation per cell) so that timing it
 * is timing the reference's design ("port" baseline in bench.py).
 *
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off matters: results must not depend on FMA fusion.
 */
#include "cuberille_oracle.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <thread>
#include <type_traits>
#include <vector>

namespace {

typedef int64_t idx_t;

struct Geometry {
  idx_t n[3];
  idx_t start[3];  // index of the first buffered pixel (ImageRegion::GetIndex): ITK's transforms and interpolators work on
                   // INDICES, buffer position + start; everything below takes and returns buffer positions
  double spacing[3], origin[3], dir[9];
  double i2p[9];   // Direction * diag(spacing)           (ImageBase::m_IndexToPhysicalPoint)
  double p2i[9];   // inverse of i2p                      (ImageBase::m_PhysicalPointToIndex)
};

// 3x3 inverse by cofactors.  ITK uses vnl's SVD-based inverse; for the identity
// direction / axis-aligned spacing of every shipped volume both are exact.
// The product library uses the same formula (csrc/cuberille_api.hip, invert3) so that the
// matrix entries handed to both sides are the same doubles.
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

Geometry make_geometry(const oracle_image *img) {
  Geometry g;
  for (int i = 0; i < 3; i++) {
    g.n[i] = img->dims[i];
    g.start[i] = img->index_start[i];
    g.spacing[i] = img->spacing[i];
    g.origin[i] = img->origin[i];
  }
  for (int i = 0; i < 9; i++) g.dir[i] = img->direction[i];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) g.i2p[r * 3 + c] = g.dir[r * 3 + c] * g.spacing[c];
  invert3(g.i2p, g.p2i);
  return g;
}

inline idx_t clampi(idx_t v, idx_t lo, idx_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// floor'ed continuous index -> pixel index clamped into [0,end]; NaN -> 0 so that a
// vertex that has gone NaN (quirk Q4) never reads out of bounds.
inline idx_t to_index_clamped(double b, idx_t end) {
  if (!(b >= 0.0)) return 0;
  if (b >= (double)end) return end;
  return (idx_t)b;
}

template <class T>
struct Image {
  Geometry g;
  const T *px;
  inline T at(idx_t x, idx_t y, idx_t z) const { return px[(z * g.n[1] + y) * g.n[0] + x]; }
  // I2: ConstShapedNeighborhoodIterator's default ZeroFluxNeumannBoundaryCondition
  // returns the nearest in-image pixel for an out-of-image neighbour (txx:99,167).
  inline T at_clamped(idx_t x, idx_t y, idx_t z) const {
    return at(clampi(x, 0, g.n[0] - 1), clampi(y, 0, g.n[1] - 1), clampi(z, 0, g.n[2] - 1));
  }
};

// I3: ImageBase::TransformIndexToPhysicalPoint, result cast to the mesh coordinate
// type float (txx:266; itk::Mesh default traits use float coordinates, I11).
void index_to_point(const Geometry &g, const idx_t idx[3], float p[3]) {
  for (int r = 0; r < 3; r++) {
    double sum = 0.0;
    for (int c = 0; c < 3; c++) sum += g.i2p[r * 3 + c] * (double)(idx[c] + g.start[c]);
    p[r] = (float)(sum + g.origin[r]);
  }
}

// I4: ImageBase::TransformPhysicalPointToContinuousIndex (double).
void point_to_cindex(const Geometry &g, const double p[3], double ci[3]) {
  double cv[3];
  for (int k = 0; k < 3; k++) cv[k] = p[k] - g.origin[k];
  for (int r = 0; r < 3; r++) {
    double sum = 0.0;
    for (int c = 0; c < 3; c++) sum += g.p2i[r * 3 + c] * cv[c];
    ci[r] = sum;
  }
}

// Weights and neighbour indices shared by I5 and I7 (ITK 3.x N-d linear
// interpolation: LinearInterpolateImageFunction::EvaluateAtContinuousIndex).
struct Cell8 {
  idx_t lo[3], hi[3];
  double d[3];
};
inline void make_cell(const Geometry &g, const double ci[3], Cell8 &c) {
  for (int k = 0; k < 3; k++) {
    const double b = std::floor(ci[k]);
    c.d[k] = ci[k] - b;
    // (the neighbours clamp into [StartIndex, EndIndex] of the buffered region; stored as buffer positions)
    c.lo[k] = to_index_clamped(b - (double)g.start[k], g.n[k] - 1);
    c.hi[k] = to_index_clamped(b + 1.0 - (double)g.start[k], g.n[k] - 1);
  }
}

// I5: scalar linear interpolation, 8-neighbour weighted sum in double in counter
// order 0..7 (bit0 = upper x, bit1 = upper y, bit2 = upper z); zero-weight
// neighbours skipped; stops early once the accumulated weight is exactly 1.
template <class T>
double interpolate(const Image<T> &im, const double point[3]) {
  double ci[3];
  point_to_cindex(im.g, point, ci);
  Cell8 c;
  make_cell(im.g, ci, c);
  double value = 0.0, total = 0.0;
  for (unsigned counter = 0; counter < 8; counter++) {
    double overlap = 1.0;
    idx_t ni[3];
    for (int k = 0; k < 3; k++) {
      if (counter & (1u << k)) { ni[k] = c.hi[k]; overlap *= c.d[k]; }
      else                     { ni[k] = c.lo[k]; overlap *= 1.0 - c.d[k]; }
    }
    if (overlap) {
      value += overlap * (double)im.at(ni[0], ni[1], ni[2]);
      total += overlap;
    }
    if (total == 1.0) break;
  }
  return value;
}

// I6: itk::GradientImageFilter at one pixel: per axis a 3-tap derivative operator
// with coefficients (-c, 0, +c), c = float(0.5 * (1/spacing)), applied as an inner
// product accumulated in float in tap order (-1, 0, +1) over ZeroFluxNeumann-clamped
// neighbours; UseImageSpacing on; UseImageDirection on (TransformLocalVectorToPhysicalVector).
template <class T>
void gradient_at_index(const Image<T> &im, idx_t x, idx_t y, idx_t z, float out[3]) {
  float local[3];
  for (int a = 0; a < 3; a++) {
    const float c = (float)(0.5 * (1.0 / im.g.spacing[a]));
    const idx_t dx = (a == 0), dy = (a == 1), dz = (a == 2);
    const float fm = (float)im.at_clamped(x - dx, y - dy, z - dz);
    const float f0 = (float)im.at(x, y, z);
    const float fp = (float)im.at_clamped(x + dx, y + dy, z + dz);
    float sum = 0.0f;
    sum += (-c) * fm;
    sum += 0.0f * f0;
    sum += c * fp;
    local[a] = sum;
  }
  for (int r = 0; r < 3; r++) {
    float sum = 0.0f;
    for (int c = 0; c < 3; c++) sum = (float)((double)sum + im.g.dir[r * 3 + c] * (double)local[c]);
    out[r] = sum;
  }
}

// I7: VectorLinearInterpolateImageFunction over the float gradient image; per
// component accumulated in double, narrowed to CovariantVector<float,3> on
// assignment (txx:451).
void interpolate_gradient(const Geometry &g, const float *grad, const double point[3], float out[3]) {
  double ci[3];
  point_to_cindex(g, point, ci);
  Cell8 c;
  make_cell(g, ci, c);
  double acc[3] = {0.0, 0.0, 0.0}, total = 0.0;
  for (unsigned counter = 0; counter < 8; counter++) {
    double overlap = 1.0;
    idx_t ni[3];
    for (int k = 0; k < 3; k++) {
      if (counter & (1u << k)) { ni[k] = c.hi[k]; overlap *= c.d[k]; }
      else                     { ni[k] = c.lo[k]; overlap *= 1.0 - c.d[k]; }
    }
    if (overlap) {
      const float *gp = grad + 3 * ((ni[2] * g.n[1] + ni[1]) * g.n[0] + ni[0]);
      for (int k = 0; k < 3; k++) acc[k] += overlap * (double)gp[k];
      total += overlap;
    }
    if (total == 1.0) break;
  }
  for (int k = 0; k < 3; k++) out[k] = (float)acc[k];
}

// I8: CovariantVector<float,3>::Normalize -- norm in double, no zero guard (quirk Q4).
inline void normalize(float v[3]) {
  double sum = 0.0;
  for (int k = 0; k < 3; k++) { const double e = (double)v[k]; sum += e * e; }
  const double norm = std::sqrt(sum);
  for (int k = 0; k < 3; k++) v[k] = (float)((double)v[k] / norm);
}

// ---------------------------------------------------------------------------------------------------------------
// USE_GRADIENT_RECURSIVE_GAUSSIAN (h:21,163-164; txx:488-491), compiled out upstream: the gradient image comes from
// itk::GradientRecursiveGaussianImageFilter with sigma = max spacing and NormalizeAcrossScale on.  The reference holds
// three lines of it; everything else is ITK (3.x era), restated here from its published algorithm -- Deriche's
// fourth-order recursive approximation of the Gaussian and its first derivative as ITK's
// RecursiveGaussianImageFilter / RecursiveSeparableImageFilter run it (coefficient tables, the causal and anti-causal
// recurrences with their constant-extension start-up, "boundary" coefficients, the float images between the passes,
// the division by the spacing and the direction matrix at the end).  PARITY UNPINNED: no fixture of the reference
// covers it and ITK cannot be built here; tests/test_oracle.py holds this to a second restatement in numpy.
// ---------------------------------------------------------------------------------------------------------------
struct Deriche {
  double N0, N1, N2, N3, D1, D2, D3, D4, M1, M2, M3, M4, BN1, BN2, BN3, BN4, BM1, BM2, BM3, BM4;
};

// RecursiveGaussianImageFilter::SetUp for order 0 (smoothing) or 1 (first derivative); sigma in physical units
Deriche deriche_setup(double sigma, double spacing, int order, bool normalizeAcrossScale) {
  static const double A1[3] = {1.3530, -0.6724, -1.3563}, B1[3] = {1.8151, -3.4327, 5.2318}, W1 = 0.6681, L1 = -1.3932;
  static const double A2[3] = {-0.3531, 0.6724, 0.3446}, B2[3] = {0.0902, 0.6100, -2.2355}, W2 = 2.0787, L2 = -1.3732;
  double direction = 1.0;
  if (spacing < 0.0) { direction = -1.0; spacing = -spacing; }
  const double sigmad = sigma / spacing;
  Deriche c;
  // ComputeDCoefficients
  const double Cos1 = std::cos(W1 / sigmad), Cos2 = std::cos(W2 / sigmad), Exp1 = std::exp(L1 / sigmad), Exp2 = std::exp(L2 / sigmad);
  c.D4 = Exp1 * Exp1 * Exp2 * Exp2;
  c.D3 = -2 * Cos1 * Exp1 * Exp2 * Exp2;
  c.D3 += -2 * Cos2 * Exp2 * Exp1 * Exp1;
  c.D2 = 4 * Cos2 * Cos1 * Exp1 * Exp2;
  c.D2 += Exp1 * Exp1 + Exp2 * Exp2;
  c.D1 = -2 * (Exp2 * Cos2 + Exp1 * Cos1);
  const double SD = 1.0 + c.D1 + c.D2 + c.D3 + c.D4;
  const double DD = c.D1 + 2 * c.D2 + 3 * c.D3 + 4 * c.D4;
  // ComputeNCoefficients
  const double a1 = A1[order], b1 = B1[order], a2 = A2[order], b2 = B2[order];
  const double Sin1 = std::sin(W1 / sigmad), Sin2 = std::sin(W2 / sigmad);
  c.N0 = a1 + a2;
  c.N1 = Exp2 * (b2 * Sin2 - (a2 + 2 * a1) * Cos2);
  c.N1 += Exp1 * (b1 * Sin1 - (a1 + 2 * a2) * Cos1);
  c.N2 = (a1 + a2) * Cos2 * Cos1;
  c.N2 -= b1 * Cos2 * Sin1 + b2 * Cos1 * Sin2;
  c.N2 *= 2 * Exp1 * Exp2;
  c.N2 += a2 * Exp1 * Exp1 + a1 * Exp2 * Exp2;
  c.N3 = Exp2 * Exp1 * Exp1 * (b2 * Sin2 - a2 * Cos2);
  c.N3 += Exp1 * Exp2 * Exp2 * (b1 * Sin1 - a1 * Cos1);
  const double SN = c.N0 + c.N1 + c.N2 + c.N3;
  const double DN = c.N1 + 2 * c.N2 + 3 * c.N3;
  bool symmetric;
  if (order == 0) {
    const double alpha0 = 2 * SN / SD - c.N0;
    c.N0 /= alpha0; c.N1 /= alpha0; c.N2 /= alpha0; c.N3 /= alpha0;
    symmetric = true;
  } else {
    const double across = normalizeAcrossScale ? sigma : 1.0;
    double alpha1 = 2 * (SN * DD - DN * SD) / (SD * SD);
    alpha1 *= direction;                          // a negative spacing negates the derivative's response
    c.N0 *= across / alpha1; c.N1 *= across / alpha1; c.N2 *= across / alpha1; c.N3 *= across / alpha1;
    symmetric = false;
  }
  // ComputeRemainingCoefficients
  if (symmetric) {
    c.M1 = c.N1 - c.D1 * c.N0; c.M2 = c.N2 - c.D2 * c.N0; c.M3 = c.N3 - c.D3 * c.N0; c.M4 = -c.D4 * c.N0;
  } else {
    c.M1 = -(c.N1 - c.D1 * c.N0); c.M2 = -(c.N2 - c.D2 * c.N0); c.M3 = -(c.N3 - c.D3 * c.N0); c.M4 = c.D4 * c.N0;
  }
  const double SN2 = c.N0 + c.N1 + c.N2 + c.N3, SM = c.M1 + c.M2 + c.M3 + c.M4, SD2 = 1.0 + c.D1 + c.D2 + c.D3 + c.D4;
  c.BN1 = c.D1 * SN2 / SD2; c.BN2 = c.D2 * SN2 / SD2; c.BN3 = c.D3 * SN2 / SD2; c.BN4 = c.D4 * SN2 / SD2;
  c.BM1 = c.D1 * SM / SD2; c.BM2 = c.D2 * SM / SD2; c.BM3 = c.D3 * SM / SD2; c.BM4 = c.D4 * SM / SD2;
  return c;
}

// RecursiveSeparableImageFilter::FilterDataArray: one line of ln >= 4 samples, in double
void deriche_line(const Deriche &c, const double *data, double *outs, double *scratch, idx_t ln) {
  const double outV1 = data[0];                     // assumed to go on from the border to infinity
  scratch[0] = outV1 * c.N0 + outV1 * c.N1 + outV1 * c.N2 + outV1 * c.N3;
  scratch[1] = data[1] * c.N0 + outV1 * c.N1 + outV1 * c.N2 + outV1 * c.N3;
  scratch[2] = data[2] * c.N0 + data[1] * c.N1 + outV1 * c.N2 + outV1 * c.N3;
  scratch[3] = data[3] * c.N0 + data[2] * c.N1 + data[1] * c.N2 + outV1 * c.N3;
  scratch[0] -= outV1 * c.BN1 + outV1 * c.BN2 + outV1 * c.BN3 + outV1 * c.BN4;
  scratch[1] -= scratch[0] * c.D1 + outV1 * c.BN2 + outV1 * c.BN3 + outV1 * c.BN4;
  scratch[2] -= scratch[1] * c.D1 + scratch[0] * c.D2 + outV1 * c.BN3 + outV1 * c.BN4;
  scratch[3] -= scratch[2] * c.D1 + scratch[1] * c.D2 + scratch[0] * c.D3 + outV1 * c.BN4;
  for (idx_t i = 4; i < ln; i++) {
    scratch[i] = data[i] * c.N0 + data[i - 1] * c.N1 + data[i - 2] * c.N2 + data[i - 3] * c.N3;
    scratch[i] -= scratch[i - 1] * c.D1 + scratch[i - 2] * c.D2 + scratch[i - 3] * c.D3 + scratch[i - 4] * c.D4;
  }
  for (idx_t i = 0; i < ln; i++) outs[i] = scratch[i];
  const double outV2 = data[ln - 1];
  scratch[ln - 1] = outV2 * c.M1 + outV2 * c.M2 + outV2 * c.M3 + outV2 * c.M4;
  scratch[ln - 2] = data[ln - 1] * c.M1 + outV2 * c.M2 + outV2 * c.M3 + outV2 * c.M4;
  scratch[ln - 3] = data[ln - 2] * c.M1 + data[ln - 1] * c.M2 + outV2 * c.M3 + outV2 * c.M4;
  scratch[ln - 4] = data[ln - 3] * c.M1 + data[ln - 2] * c.M2 + data[ln - 1] * c.M3 + outV2 * c.M4;
  scratch[ln - 1] -= outV2 * c.BM1 + outV2 * c.BM2 + outV2 * c.BM3 + outV2 * c.BM4;
  scratch[ln - 2] -= scratch[ln - 1] * c.D1 + outV2 * c.BM2 + outV2 * c.BM3 + outV2 * c.BM4;
  scratch[ln - 3] -= scratch[ln - 2] * c.D1 + scratch[ln - 1] * c.D2 + outV2 * c.BM3 + outV2 * c.BM4;
  scratch[ln - 4] -= scratch[ln - 3] * c.D1 + scratch[ln - 2] * c.D2 + scratch[ln - 1] * c.D3 + outV2 * c.BM4;
  for (idx_t i = ln - 4; i > 0; i--) {
    scratch[i - 1] = data[i] * c.M1 + data[i + 1] * c.M2 + data[i + 2] * c.M3 + data[i + 3] * c.M4;
    scratch[i - 1] -= scratch[i] * c.D1 + scratch[i + 1] * c.D2 + scratch[i + 2] * c.D3 + scratch[i + 3] * c.D4;
  }
  for (idx_t i = 0; i < ln; i++) outs[i] += scratch[i];
}

// one separable pass along `axis` over a whole volume: `get(i)` reads the input pixel (as a double), the result is
// stored as float (the filter's internal images are float)
template <class Get>
void deriche_pass(const Geometry &g, const Deriche &c, int axis, Get get, float *out) {
  const idx_t n[3] = {g.n[0], g.n[1], g.n[2]};
  const idx_t stride[3] = {1, n[0], n[0] * n[1]};
  const idx_t ln = n[axis];
  const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
  std::vector<double> data(ln), outs(ln), scratch(ln);
  for (idx_t v = 0; v < n[a2]; v++)
    for (idx_t u = 0; u < n[a1]; u++) {
      const idx_t base = u * stride[a1] + v * stride[a2];
      for (idx_t i = 0; i < ln; i++) data[i] = get(base + i * stride[axis]);
      deriche_line(c, data.data(), outs.data(), scratch.data(), ln);
      for (idx_t i = 0; i < ln; i++) out[base + i * stride[axis]] = (float)outs[i];
    }
}

// h:243-313: (x,y) -> point id, ordered by y then x (h:262-263).
struct LookupNode {
  unsigned long x, y;
  bool operator<(const LookupNode &o) const { return (y < o.y) || ((y == o.y) && (x < o.x)); }
};
struct VertexLookupMap {
  std::map<LookupNode, uint64_t> m;
  void Clear() { m.clear(); }
  void AddVertex(unsigned long x, unsigned long y, uint64_t id) { m.insert({LookupNode{x, y}, id}); }
  bool GetVertex(unsigned long x, unsigned long y, uint64_t &id) {
    auto it = m.find(LookupNode{x, y});
    if (it == m.end()) return false;
    id = it->second;
    return true;
  }
};

struct CellObj { uint64_t ids[4]; };   // stands for one heap-allocated itk cell

template <class T>
struct Filter {
  Image<T> im;
  Image<T> gim;                      // the image the gradient interpolator was created on: `im` itself on a filter's first
                                     // projecting Update(); on later ones STILL that first image (quirk Q3, txx:484: the
                                     // interpolator is only ever created while it is null)
  oracle_params prm;
  T iso;
  double step_length;
  std::vector<float> grad;           // ComputeGradientImage output, 3 floats per pixel
  std::vector<double> gradD;         // ... of the recursive-Gaussian variant: CovariantVector<double,3> per pixel
  std::vector<float> points;
  std::vector<uint64_t> cells_flat;
  std::vector<std::unique_ptr<CellObj>> cells_heap;
  uint64_t iters = 0, stop_thr = 0, stop_steps = 0;

  // GradientRecursiveGaussianImageFilter::GenerateData (ITK 3.x): per component `dim` the derivative filter along
  // `dim` first, then the smoothing filters along the other axes in ascending order, float images in between; the
  // result divided by the spacing of `dim`; the direction matrix applied to the vector at the end
  void ComputeGradientImageRecursiveGaussian() {
    const Geometry &g = gim.g;
    const size_t N = (size_t)g.n[0] * g.n[1] * g.n[2];
    double sigma = g.spacing[0];                              // txx:489: m_MaxSpacing * 1.0
    for (int i = 1; i < 3; i++) sigma = sigma > g.spacing[i] ? sigma : g.spacing[i];
    gradD.assign(3 * N, 0.0);
    std::vector<float> a(N), b(N);
    for (int dim = 0; dim < 3; dim++) {
      const Deriche dc = deriche_setup(sigma, g.spacing[dim], 1, true);       // txx:490: NormalizeAcrossScale
      const T *px = gim.px;
      deriche_pass(g, dc, dim, [&](idx_t i) { return (double)px[i]; }, a.data());
      float *src = a.data(), *dst = b.data();
      for (int ax = 0; ax < 3; ax++) {
        if (ax == dim) continue;
        const Deriche sc = deriche_setup(sigma, g.spacing[ax], 0, true);
        const float *in = src;
        deriche_pass(g, sc, ax, [&](idx_t i) { return (double)in[i]; }, dst);
        std::swap(src, dst);
      }
      for (size_t i = 0; i < N; i++) gradD[3 * i + dim] = (double)src[i] / g.spacing[dim];
    }
    for (size_t i = 0; i < N; i++) {                          // TransformLocalVectorToPhysicalVector
      const double l[3] = {gradD[3 * i], gradD[3 * i + 1], gradD[3 * i + 2]};
      for (int r = 0; r < 3; r++) {
        double sum = 0.0;
        for (int c = 0; c < 3; c++) sum += g.dir[r * 3 + c] * l[c];
        gradD[3 * i + r] = sum;
      }
    }
  }

  // the normal at a point (txx:451-452 and its twins in the other branches), as a double vector for the step that
  // follows.  Default gradient: CovariantVector<float,3> interpolated and normalised as I7 / I8 say; the recursive-
  // Gaussian variant's GradientPixelType is CovariantVector<double,3>: interpolation and Normalize() stay in double
  void NormalAt(const double p[3], double nd[3]) {
    if (prm.gradient_variant == 1) {
      double ci[3];
      point_to_cindex(gim.g, p, ci);
      Cell8 c;
      make_cell(gim.g, ci, c);
      double acc[3] = {0.0, 0.0, 0.0}, total = 0.0;
      for (unsigned counter = 0; counter < 8; counter++) {
        double overlap = 1.0;
        idx_t ni[3];
        for (int k = 0; k < 3; k++) {
          if (counter & (1u << k)) { ni[k] = c.hi[k]; overlap *= c.d[k]; }
          else                     { ni[k] = c.lo[k]; overlap *= 1.0 - c.d[k]; }
        }
        if (overlap) {
          const double *gp = &gradD[3 * ((ni[2] * gim.g.n[1] + ni[1]) * gim.g.n[0] + ni[0])];
          for (int k = 0; k < 3; k++) acc[k] += overlap * gp[k];
          total += overlap;
        }
        if (total == 1.0) break;
      }
      double sum = 0.0;
      for (int k = 0; k < 3; k++) sum += acc[k] * acc[k];
      const double norm = std::sqrt(sum);
      for (int k = 0; k < 3; k++) nd[k] = acc[k] / norm;
    } else {
      float normal[3];
      interpolate_gradient(gim.g, grad.data(), p, normal);
      normalize(normal);
      for (int k = 0; k < 3; k++) nd[k] = (double)normal[k];
    }
  }

  // txx:478-498 (whole image, threads over z like ITK's ThreadedGenerateData)
  void ComputeGradientImage() {
    if (prm.gradient_variant == 1) { ComputeGradientImageRecursiveGaussian(); return; }
    const Geometry &g = gim.g;
    grad.resize((size_t)3 * g.n[0] * g.n[1] * g.n[2]);
    int nt = prm.gradient_threads > 0 ? prm.gradient_threads : 1;
    if (nt > g.n[2]) nt = (int)g.n[2];
    auto work = [&](idx_t z0, idx_t z1) {
      for (idx_t z = z0; z < z1; z++)
        for (idx_t y = 0; y < g.n[1]; y++)
          for (idx_t x = 0; x < g.n[0]; x++)
            gradient_at_index(gim, x, y, z, &grad[3 * ((z * g.n[1] + y) * g.n[0] + x)]);
    };
    if (nt <= 1) { work(0, g.n[2]); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back(work, g.n[2] * t / nt, g.n[2] * (t + 1) / nt);
    for (auto &t : th) t.join();
  }

  // txx:439-474
  void ProjectVertexToIsoSurface(float vertex[3]) {
    bool done = false;
    double sign = 1.0;
    double step = step_length;
    unsigned int numberOfSteps = 0;
    double normal[3];
    while (!done) {
      iters++;
      const double p[3] = {(double)vertex[0], (double)vertex[1], (double)vertex[2]};
      NormalAt(p, normal);                                                    // txx:451-452
      const double value = interpolate(im, p);                                // txx:455
      done |= std::fabs(value - (double)iso) < prm.distance_threshold;        // txx:456
      if (done) { stop_thr++; break; }                                        // txx:460
      sign = (value < (double)iso) ? +1.0 : -1.0;                             // txx:463
      for (int i = 0; i < 3; i++)                                             // txx:464-467 (I9)
        vertex[i] = (float)((double)vertex[i] + (normal[i] * sign * step));
      step *= prm.relaxation;                                                 // txx:468
      done |= numberOfSteps++ > prm.max_steps;                                // txx:469
      if (done) stop_steps++;
    }
  }

  // txx:340-397, the USE_ADVANCED_PROJECTION branch (compiled out upstream, h:22): every pass steps BOTH ways along
  // the normal and keeps the end that is closer to the iso value; stops within the threshold, out of steps, or after
  // five passes that chose the other side than the first pass did
  void ProjectVertexAdvanced(float vertex[3]) {
    bool done = false;
    double step = step_length;
    unsigned int numberOfSteps = 0;
    float temp[2][3];
    double value[2], diff[2];
    unsigned int i, swaps = 0;
    int previousi = -1;
    while (!done) {
      iters++;
      const double p[3] = {(double)vertex[0], (double)vertex[1], (double)vertex[2]};
      double normal[3];
      NormalAt(p, normal);                                                    // txx:356-357
      for (int k = 0; k < 3; k++) {                                           // txx:360-364
        temp[0][k] = (float)((double)vertex[k] + (normal[k] * +1.0 * step));
        temp[1][k] = (float)((double)vertex[k] + (normal[k] * -1.0 * step));
      }
      step *= prm.relaxation;                                                 // txx:365
      for (int e = 0; e < 2; e++) {                                           // txx:368-371
        const double q[3] = {(double)temp[e][0], (double)temp[e][1], (double)temp[e][2]};
        value[e] = interpolate(im, q);
        diff[e] = std::fabs(value[e] - (double)iso);
      }
      i = (diff[0] <= diff[1]) ? 0 : 1;                                       // txx:372
      if (previousi < 0) previousi = (int)i;                                  // txx:373
      swaps += (unsigned int)(previousi != (int)i);                           // txx:374
      for (int k = 0; k < 3; k++) vertex[k] = temp[i][k];                     // txx:375
      done |= diff[i] < prm.distance_threshold;                               // txx:378
      if (done) { stop_thr++; break; }
      done |= numberOfSteps++ > prm.max_steps;                                // txx:385
      if (done) { stop_steps++; break; }
      done |= (swaps >= 5);                                                   // txx:392
      if (done) break;
    }
  }

  // txx:398-437, the USE_LINESEARCH_PROJECTION branch (compiled out upstream, h:23): one normal, MaximumNumberOfSteps/2 - 1
  // samples on either side of the vertex out to the step length, the sample closest to the iso value wins.
  // Upstream leaves bestVertex uninitialised when no sample beats the initial 10000 (or there is no sample at all);
  // here the vertex then stays where it is.
  void ProjectVertexLineSearch(float vertex[3]) {
    float bestVertex[3] = {vertex[0], vertex[1], vertex[2]};
    double normal[3];
    double bestMetric = 10000;
    const double p[3] = {(double)vertex[0], (double)vertex[1], (double)vertex[2]};
    NormalAt(p, normal);                                                      // txx:408-409
    for (double sign = -1.0; sign <= 1.0; sign += 2.0) {                      // txx:412
      for (unsigned int j = 1; j < prm.max_steps / 2; j++) {                  // txx:415
        iters++;
        const double d = (double)j / ((double)prm.max_steps / 2.0);           // txx:418
        float temp[3];
        for (int k = 0; k < 3; k++)                                           // txx:419-422
          temp[k] = (float)((double)vertex[k] + (normal[k] * sign * step_length * d));
        const double q[3] = {(double)temp[0], (double)temp[1], (double)temp[2]};
        const double metric = std::fabs(interpolate(im, q) - (double)iso);    // txx:424-425
        if (metric < bestMetric) {                                            // txx:430-434
          bestMetric = metric;
          for (int k = 0; k < 3; k++) bestVertex[k] = temp[k];
        }
      }
    }
    for (int k = 0; k < 3; k++) vertex[k] = bestVertex[k];                    // txx:437
  }

  // txx:256-276
  void AddVertex(uint64_t &id, const idx_t index[3]) {
    float vertex[3];
    index_to_point(im.g, index, vertex);                                      // txx:266
    for (int k = 0; k < 3; k++)                                               // txx:268-270
      vertex[k] = (float)((double)vertex[k] - (im.g.spacing[k] / 2.0));
    if (prm.project_vertices) {                                               // txx:271-274
      if (prm.projection_variant == 1) ProjectVertexAdvanced(vertex);
      else if (prm.projection_variant == 2) ProjectVertexLineSearch(vertex);
      else ProjectVertexToIsoSurface(vertex);
    }
    if (points.size() < 3 * (id + 1)) points.resize(3 * (id + 1));
    std::memcpy(&points[3 * id], vertex, sizeof(vertex));                     // txx:275
    id++;
  }

  void store_cell(uint64_t &id, const uint64_t *ids, int n) {
    if (prm.faithful_cells) {
      std::unique_ptr<CellObj> c(new CellObj);                                // txx:311,318,327
      for (int k = 0; k < n; k++) c->ids[k] = ids[k];
      cells_heap.push_back(std::move(c));
    } else {
      for (int k = 0; k < n; k++) cells_flat.push_back(ids[k]);
    }
    id++;
  }

  // I10: Point::SquaredEuclideanDistanceTo on float points, accumulated in double.
  static double sqdist(const float *a, const float *b) {
    double sum = 0.0;
    for (int i = 0; i < 3; i++) { const double d = (double)b[i] - (double)a[i]; sum += d * d; }
    return sum;
  }

  // txx:278-332
  void AddQuadFace(uint64_t &id, const uint64_t face[4]) {
    if (prm.generate_triangles) {
      const float *v[4];
      for (int i = 0; i < 4; i++) v[i] = &points[3 * face[i]];                // txx:289-293
      uint64_t f1[3], f2[3];
      if (sqdist(v[0], v[2]) >= sqdist(v[1], v[3])) {                         // txx:298
        f1[0] = face[0]; f1[1] = face[1]; f1[2] = face[3];
        f2[0] = face[1]; f2[1] = face[2]; f2[2] = face[3];
      } else {
        f1[0] = face[0]; f1[1] = face[1]; f1[2] = face[2];
        f2[0] = face[0]; f2[1] = face[2]; f2[2] = face[3];
      }
      store_cell(id, f1, 3);
      store_cell(id, f2, 3);
    } else {
      store_cell(id, face, 4);
    }
  }

  // txx:218-233
  static void SetVerticesFromFace(unsigned face, bool *v) {
    static const int FV[6][4] = {{0, 4, 7, 3}, {0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {0, 3, 2, 1}, {4, 5, 6, 7}};
    for (int k = 0; k < 4; k++) v[FV[face][k]] = true;
  }
  // txx:235-254
  static void GetVertexLookupIndex(unsigned vertex, const idx_t index[3], idx_t result[3]) {
    static const int VO[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    for (int k = 0; k < 3; k++) result[k] = index[k] + VO[vertex][k];
  }

  // txx:59-216
  void GenerateData() {
    const Geometry &g = im.g;
    double maxSpacing = g.spacing[0];                                         // txx:75-79
    for (int i = 1; i < 3; i++) maxSpacing = maxSpacing > g.spacing[i] ? maxSpacing : g.spacing[i];
    step_length = prm.step_length;
    if (step_length < 0.0) step_length = maxSpacing * 0.25;                   // txx:82-85

    unsigned look, look0, look1;
    unsigned char numFaces;
    bool faceHasQuad[6], vertexHasQuad[8];
    uint64_t v[8], f[4];
    uint64_t nextVertexId = 0, nextCellId = 0;
    idx_t lastZ = -1;
    static const int offset[6][3] = {{-1, 0, 0}, {0, -1, 0}, {+1, 0, 0}, {0, +1, 0}, {0, 0, -1}, {0, 0, +1}}; // txx:122-127
    VertexLookupMap lookup[2];
    look0 = 1; look1 = 0;                                                     // txx:129

    // txx:136: raster order over the buffered region, x fastest (I1)
    for (idx_t z = 0; z < g.n[2]; z++)
      for (idx_t y = 0; y < g.n[1]; y++)
        for (idx_t x = 0; x < g.n[0]; x++) {
          const T center = im.at(x, y, z);
          if (center < iso) continue;                                         // txx:139-141
          numFaces = 0;
          for (int i = 0; i < 6; i++) faceHasQuad[i] = false;
          for (int i = 0; i < 8; i++) vertexHasQuad[i] = false;
          const idx_t index[3] = {x, y, z};
          if (z != lastZ) {                                                   // txx:156-161
            unsigned t = look0; look0 = look1; look1 = t;
            lookup[look1].Clear();
            lastZ = z;
          }
          for (unsigned i = 0; i < 6; i++) {                                  // txx:164-173
            faceHasQuad[i] = im.at_clamped(x + offset[i][0], y + offset[i][1], z + offset[i][2]) < iso;
            if (faceHasQuad[i]) { numFaces++; SetVerticesFromFace(i, vertexHasQuad); }
          }
          if (numFaces > 0) {
            for (unsigned i = 0; i < 8; i++) {                                // txx:179-194
              if (!vertexHasQuad[i]) continue;
              idx_t vindex[3];
              GetVertexLookupIndex(i, index, vindex);
              look = (i < 4) ? look0 : look1;
              if (!lookup[look].GetVertex((unsigned long)vindex[0], (unsigned long)vindex[1], v[i])) {
                v[i] = nextVertexId;
                AddVertex(nextVertexId, vindex);
                lookup[look].AddVertex((unsigned long)vindex[0], (unsigned long)vindex[1], v[i]);
              }
            }
            // txx:197-202
            if (faceHasQuad[0]) { f[0] = v[0]; f[1] = v[4]; f[2] = v[7]; f[3] = v[3]; AddQuadFace(nextCellId, f); }
            if (faceHasQuad[1]) { f[0] = v[0]; f[1] = v[1]; f[2] = v[5]; f[3] = v[4]; AddQuadFace(nextCellId, f); }
            if (faceHasQuad[2]) { f[0] = v[1]; f[1] = v[2]; f[2] = v[6]; f[3] = v[5]; AddQuadFace(nextCellId, f); }
            if (faceHasQuad[3]) { f[0] = v[2]; f[1] = v[3]; f[2] = v[7]; f[3] = v[6]; AddQuadFace(nextCellId, f); }
            if (faceHasQuad[4]) { f[0] = v[0]; f[1] = v[3]; f[2] = v[2]; f[3] = v[1]; AddQuadFace(nextCellId, f); }
            if (faceHasQuad[5]) { f[0] = v[4]; f[1] = v[5]; f[2] = v[6]; f[3] = v[7]; AddQuadFace(nextCellId, f); }
          }
        }
  }
};

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class T>
int run_typed(const oracle_image *img, const oracle_image *gradient_of, const oracle_params *prm, oracle_mesh *out) {
  Filter<T> f;
  f.im.g = make_geometry(img);
  f.im.px = (const T *)img->voxels;
  f.gim = f.im;
  if (gradient_of) {
    f.gim.g = make_geometry(gradient_of);
    f.gim.px = (const T *)gradient_of->voxels;
  }
  f.prm = *prm;
  // m_IsoSurfaceValue is an InputPixelType (h:180-181); the 64-bit integer types get theirs as an integer
  if (std::is_integral<T>::value && sizeof(T) == 8) f.iso = (T)prm->iso_value_int;
  else f.iso = (T)prm->iso_value;
  double t0 = now_s();
  if (prm->project_vertices) f.ComputeGradientImage();                        // txx:95,484
  double t1 = now_s();
  f.GenerateData();
  double t2 = now_s();
  const int vpc = prm->generate_triangles ? 3 : 4;
  out->verts_per_cell = vpc;
  out->n_points = f.points.size() / 3;
  out->points = (float *)std::malloc(sizeof(float) * (f.points.size() ? f.points.size() : 1));
  std::memcpy(out->points, f.points.data(), sizeof(float) * f.points.size());
  if (prm->faithful_cells) {
    out->n_cells = f.cells_heap.size();
    out->cells = (uint64_t *)std::malloc(sizeof(uint64_t) * (out->n_cells * vpc + 1));
    for (uint64_t c = 0; c < out->n_cells; c++)
      for (int k = 0; k < vpc; k++) out->cells[c * vpc + k] = f.cells_heap[c]->ids[k];
  } else {
    out->n_cells = f.cells_flat.size() / vpc;
    out->cells = (uint64_t *)std::malloc(sizeof(uint64_t) * (f.cells_flat.size() + 1));
    std::memcpy(out->cells, f.cells_flat.data(), sizeof(uint64_t) * f.cells_flat.size());
  }
  out->seconds_gradient = t1 - t0;
  out->seconds_sweep = t2 - t1;
  out->proj_iterations = f.iters;
  out->proj_stop_threshold = f.stop_thr;
  out->proj_stop_steps = f.stop_steps;
  return 0;
}

template <class F>
auto dispatch(int pixel_type, F &&fn, int &err) {
  err = 0;
  switch (pixel_type) {
    case ORACLE_PIX_U8:  return fn((uint8_t *)nullptr);
    case ORACLE_PIX_I8:  return fn((int8_t *)nullptr);
    case ORACLE_PIX_U16: return fn((uint16_t *)nullptr);
    case ORACLE_PIX_I16: return fn((int16_t *)nullptr);
    case ORACLE_PIX_U32: return fn((uint32_t *)nullptr);
    case ORACLE_PIX_I32: return fn((int32_t *)nullptr);
    case ORACLE_PIX_F32: return fn((float *)nullptr);
    case ORACLE_PIX_F64: return fn((double *)nullptr);
    case ORACLE_PIX_I64: return fn((int64_t *)nullptr);
    case ORACLE_PIX_U64: return fn((uint64_t *)nullptr);
  }
  err = 1;
  return fn((uint8_t *)nullptr);
}

bool valid_image(const oracle_image *img) {
  if (!img || !img->voxels) return false;
  if (img->pixel_type < 0 || img->pixel_type > ORACLE_PIX_U64) return false;
  for (int i = 0; i < 3; i++) if (img->dims[i] < 1 || !(img->spacing[i] > 0.0)) return false;
  for (int i = 0; i < 3; i++) if (img->index_start[i] < -(1LL << 30) || img->index_start[i] > (1LL << 30)) return false;
  return true;
}

}  // namespace

extern "C" {

int cuberille_oracle_run_after(const oracle_image *img, const oracle_image *first, const oracle_params *prm, oracle_mesh *out) {
  if (!valid_image(img) || !prm || !out) return 1;
  if (first && (!valid_image(first) || first->pixel_type != img->pixel_type)) return 1;   // one filter object, one TInputImage
  std::memset(out, 0, sizeof(*out));
  if (prm->gradient_variant < 0 || prm->gradient_variant > 1) return 1;
  if (prm->gradient_variant == 1 && prm->project_vertices)      // ITK: "the number of pixels along a direction must be >= 4"
    for (int i = 0; i < 3; i++) if ((first ? first : img)->dims[i] < 4) return 1;
  int err = 0;
  int rc = dispatch(img->pixel_type, [&](auto *tag) {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    return run_typed<T>(img, first, prm, out);
  }, err);
  return err ? err : rc;
}

int cuberille_oracle_run(const oracle_image *img, const oracle_params *prm, oracle_mesh *out) {
  return cuberille_oracle_run_after(img, nullptr, prm, out);
}

void cuberille_oracle_free(oracle_mesh *m) {
  if (!m) return;
  std::free(m->points);
  std::free(m->cells);
  m->points = nullptr;
  m->cells = nullptr;
}

double cuberille_oracle_interpolate(const oracle_image *img, const double point[3]) {
  if (!valid_image(img)) return NAN;
  int err = 0;
  return dispatch(img->pixel_type, [&](auto *tag) {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    Image<T> im;
    im.g = make_geometry(img);
    im.px = (const T *)img->voxels;
    return interpolate(im, point);
  }, err);
}

void cuberille_oracle_gradient_at_index(const oracle_image *img, const int64_t idx[3], float g[3]) {
  if (!valid_image(img)) return;
  int err = 0;
  dispatch(img->pixel_type, [&](auto *tag) {
    typedef typename std::remove_pointer<decltype(tag)>::type T;
    Image<T> im;
    im.g = make_geometry(img);
    im.px = (const T *)img->voxels;
    gradient_at_index(im, idx[0], idx[1], idx[2], g);
    return 0;
  }, err);
}

void cuberille_oracle_index_to_point(const oracle_image *img, const int64_t idx[3], float p[3]) {
  if (!valid_image(img)) return;
  Geometry g = make_geometry(img);
  index_to_point(g, idx, p);
}

}  // extern "C"
