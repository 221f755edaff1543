// HIP kernels of the cuberille hot path for gfx950 (MI355X, 64-lane wavefronts).
//
// Reference being re-implemented (not translated):
//   /root/reference/Source/itkCuberilleImageToMeshFilter.txx:136-206 (sweep),
//   256-332 (AddVertex/AddQuadFace), 439-474 (projection), 478-498 (gradient).
//
// Design (DESIGN.md sections 4-5): the order-dependent raster sweep with its two std::map lookups
// is replaced by a closed form over a packed inside-bit volume:
//   classify : one coalesced, nontemporal pass over the voxels, 16 B per lane, thresholds against the
//              iso value and packs 64 voxels per uint64 word (DPP OR inside the lane group); the words of a
//              1 MiB span are staged in LDS and leave as 16-byte write-through stores          -> bits
//   count    : per block of 2048 words: face masks for every word, then one lane per word that has a
//              face: 3-input boolean algebra (v_bitop3) over the 27 neighbour bit-rows gives the 8 "this
//              voxel creates corner i" masks; popcounts + a wavefront scan per 64-word segment + a scan
//              over the block's 32 segments; the words that create vertices go to a queue, in order (ballots);
//              where most words carry surface the bit rows are read from an LDS tile
//   scan     : k_block_scan, one workgroup: block totals -> absolute bases, grand totals (and, for a step
//              launched blindly, the go/no-go for the launches behind it)            -> prefix, segPre, blockBase
//   emit     : points: one lane per vertex-creating word writes descriptors into LDS, then one lane per
//              vertex; cells: one lane per output quad, located by a per-wave search of the prefix
//              arrays, ids staged through LDS; everything lands at its final, reference-order index;
//              corner ids through a dense lattice-corner map; a rank's id offset from the gathered rows of
//              all ranks when the step runs without the host in between
//   project  : refilling waves, the damped gradient walk with the gradient image evaluated on the
//              fly (never materialised); runs between the point and the cell pass so that the
//              shorter-diagonal triangle split is fused into the cell pass.  On a THIN_HALO slab a walk that
//              wants a slice the buffer lacks is put aside and walked again once the slice is there.
//              Plain lane-per-vertex kernels for the reference's compiled-out variants (the two other
//              projection branches, the recursive-Gaussian gradient image).
// Bit-exactness of the floating-point part against the CPU oracle relies on
// -ffp-contract=off (no FMA fusion; the explicit fma calls in the walk are the compiler's own f64
// sqrt / division sequences written out) and IEEE f64 arithmetic; see csrc/Makefile.
// Development switches live in cuberille::Tuning (cuberille_debug_set_option), not in the environment.

#include "cuberille_internal.h"
#include "../../include/cuberille_hip.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

namespace cuberille {

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ u64 lowmask(int b) { return (1ull << b) - 1ull; }   // b in 0..63
__device__ __forceinline__ int popc64(u64 v) { return __popcll(v); }

// the iso value in the pixel type (txx:140: m_IsoSurfaceValue IS an InputPixelType): a C cast of the double, except for
// the 64-bit integer types, whose value travels as an integer
template <class T>
__device__ __forceinline__ T iso_as(double isoD, long long isoI) {
  if constexpr (std::is_integral<T>::value && sizeof(T) == 8) return (T)isoI;
  else return (T)isoD;
}

// corner number i (txx:244-251) -> block position code e = x | y<<1 | z<<2, and back
__device__ __constant__ const int kCornerEnc[8] = {0, 1, 3, 2, 4, 5, 7, 6};
__device__ __constant__ const int kEncCorner[8] = {0, 1, 3, 2, 4, 5, 7, 6};
// face -> its four corners in the order of txx:197-202
__device__ __constant__ const int kFaceCorner[6][4] = {{0, 4, 7, 3}, {0, 1, 5, 4}, {1, 2, 6, 5},
                                                       {2, 3, 7, 6}, {0, 3, 2, 1}, {4, 5, 6, 7}};

// ---------------------------------------------------------------------------------------------
// K1: classify (threshold + bit-pack).  inside(u) := !(pixel(u) < iso)   (txx:139-141,167)
// ---------------------------------------------------------------------------------------------
template <class T>
struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  union { uint4 raw; T v[N]; };
};

// 3-input boolean function on the gfx950 v_bitop3_b32 (truth table TT: bit (a<<2 | b<<1 | c))
template <int TT>
__device__ __forceinline__ u32 bop3_32(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }

// inside-bits of the 16/sizeof(T) pixels of one 16-byte load: bit j = !(pixel j < iso)
template <class T>
__device__ __forceinline__ u32 inside_bits(const Vec16<T> &r, T iso) {
  if constexpr (sizeof(T) == 1) {
    // 1-byte pixels, SWAR on the packed dwords.  Per byte, unsigned x >= t: with xl, tl the low 7 bits, bit 7 of
    // ((xl | 0x80) - tl) says xl >= tl (no borrow crosses bytes), and x >= t is (x7 | that) when t < 128, (x7 & that)
    // when t >= 128; signed pixels are biased by 0x80 first.  Three instructions per dword on v_bitop3 ((x & 0x7f..) |
    // 0x80.., the subtraction, (x op d) & 0x80..), and the four bit-7s of a dword are gathered by ONE v_dot4_u32_u8
    // against the byte weights 1, 2, 4, 8 (16 .. 128 for the odd dwords), accumulating as it goes: a sum of 0x80 * weight
    // per inside voxel, i.e. the eight bits of two dwords shifted left by 7.  (The 32-bit multiply that used to gather
    // them issues at a quarter of the rate: 2048^3 uint8 went from 1.69 ms to the figure in DESIGN.md.)
    const u32 bias = std::is_signed<T>::value ? 0x80808080u : 0u;
    const u32 t = ((u32)(unsigned char)iso) ^ (bias & 0x80u);
    const u32 tl = (t & 0x7fu) * 0x01010101u;
    const bool thigh = (t & 0x80u) != 0;                // wave-uniform
    const u32 w[4] = {r.raw.x, r.raw.y, r.raw.z, r.raw.w};
    u32 acc[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const u32 x = w[j] ^ bias;
      const u32 d = bop3_32<0xEA>(x, 0x7f7f7f7fu, 0x80808080u) - tl;              // (x & 0x7f..) | 0x80..
      const u32 ge = thigh ? bop3_32<0x80>(x, d, 0x80808080u) : bop3_32<0xA8>(x, d, 0x80808080u);   // (x &| d) & 0x80..
      acc[j >> 1] = __builtin_amdgcn_udot4(ge, (j & 1) ? 0x80402010u : 0x08040201u, acc[j >> 1], false);
    }
    return (acc[0] >> 7) | (acc[1] << 1);
  } else {
    u32 m = 0;
#pragma unroll
    for (int j = 0; j < Vec16<T>::N; j++) m |= (!(r.v[j] < iso) ? 1u : 0u) << j;
    return m;
  }
}

// Fast path: nx % 64 == 0, so the volume is a flat array of 64-voxel words.  Each lane
// loads 16 B (VPL voxels), builds VPL bits; LPW = 64/VPL adjacent lanes OR their partial
// words together.  One wave turns U KiB of voxels into U*VPL words per trip.
template <class T, int U, bool NT>
__global__ __launch_bounds__(256) void k_classify_flat(const T *__restrict__ vox, u64 *__restrict__ bits,
                                                       u64 nchunks, double isoD, long long isoI, u32 *__restrict__ sliceOcc,
                                                       int lgWordsPerSlice, u64 wordBase) {
  constexpr int VPL = 16 / sizeof(T);
  constexpr int LPW = 64 / VPL;
  const T iso = iso_as<T>(isoD, isoI);
  const int lane = threadIdx.x & 63;
  const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
  const int sub = lane % LPW;                 // position of this lane inside its word
  for (u64 c = wave * U; c < nchunks; c += nwaves * U) {
    Vec16<T> r[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (c + u < nchunks) {
        const uint4 *src = reinterpret_cast<const uint4 *>(vox + ((c + u) * 64 + lane) * VPL);
        if (NT) {   // streamed once: keep it out of the way of the bit volume in L2 / Infinity Cache
          r[u].raw.x = __builtin_nontemporal_load(&src->x); r[u].raw.y = __builtin_nontemporal_load(&src->y);
          r[u].raw.z = __builtin_nontemporal_load(&src->z); r[u].raw.w = __builtin_nontemporal_load(&src->w);
        } else {
          r[u].raw = *src;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (c + u >= nchunks) break;            // wave-uniform
      const u32 m = inside_bits<T>(r[u], iso);
      u64 part = (u64)m << (sub * VPL);
#pragma unroll
      for (int s = 1; s < LPW; s <<= 1) part |= __shfl_xor(part, s, 64);
      if (sub == 0) {
        const u64 widx = (c + u) * VPL + lane / LPW;
        bits[widx] = part;
        // slice occupancy on the fly when a slice is 2^n words (else k_occupancy derives it afterwards)
        if (lgWordsPerSlice >= 0 && part) sliceOcc[(wordBase + widx) >> lgWordsPerSlice] = 1u;   // benign race: all store 1
      }
    }
  }
}

// OR over each group of LPW consecutive lanes on the DPP data path (row_shr steps inside the 16-lane rows, row_bcast
// across them; no LDS round trips); the result is valid in the group's LAST lane.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ u64 or_dpp64(u64 v) {
  const u32 lo = (u32)v, hi = (u32)(v >> 32);
  const u32 lo2 = lo | (u32)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, ROWMASK, 0xf, false);
  const u32 hi2 = hi | (u32)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, ROWMASK, 0xf, false);
  return (u64)lo2 | ((u64)hi2 << 32);
}
template <int LPW>
__device__ __forceinline__ u64 group_or(u64 part) {
  if (LPW >= 2) part = or_dpp64<0x111, 0xf>(part);     // row_shr:1
  if (LPW >= 4) part = or_dpp64<0x112, 0xf>(part);     // row_shr:2
  if (LPW >= 8) part = or_dpp64<0x114, 0xf>(part);     // row_shr:4
  if (LPW >= 16) part = or_dpp64<0x118, 0xf>(part);    // row_shr:8
  if (LPW >= 32) part = or_dpp64<0x142, 0xa>(part);    // row_bcast:15 into rows 1 and 3
  return part;
}

// Large volumes: the sweep is bound by HBM reads (6.7-7.1 TB/s for a kernel that only reads), and what costs it is
// (1) the 1-bit-per-voxel WRITE stream trickling out of L2 between the reads (measured, profiles/microbench: 0.78 ms
// with write-back 32-byte stores, 0.63 ms with the stores removed) and (2) too many loads in flight: the memory side
// runs best with about 32 KiB outstanding per CU (two workgroups of four waves with 4 KiB each: 0.62-0.66 ms) and
// loses 5-10 % with the 128-256 KiB a full grid keeps in flight.  Here a block owns whole spans of SPAN_WORDS words:
// its four waves threshold 4 KiB trips in turn into an LDS stage, then the 32 KiB of words leave as 16-byte
// write-through (sc1) stores in one burst; the grid is two workgroups per CU.
constexpr int SPAN_WORDS = 4096;

// SPAN: words per span -- SPAN_WORDS, or a quarter of it where the volume is fewer than two rounds of such spans (a 512^3 float32
// volume is ONE: every workgroup then ends in its 32 KiB store burst at the same moment, behind the last read; with four
// rounds of 8 KiB bursts the stores of a round leave beside the next round's reads)
template <class T, int SPAN = SPAN_WORDS>
__global__ __launch_bounds__(256) void k_classify_span(const T *__restrict__ vox, u64 *__restrict__ bits, u64 nspans,
                                                       double isoD, long long isoI, u32 *__restrict__ sliceOcc,
                                                       int lgWordsPerSlice) {
  constexpr int U = 4;
  constexpr int VPL = 16 / sizeof(T);
  constexpr int LPW = 64 / VPL;
  constexpr int TRIPS = SPAN / (4 * U * VPL);      // trips of U KiB per wave and span
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) u64 stage[SPAN];
  const T iso = iso_as<T>(isoD, isoI);
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane % LPW;
  const bool last = sub == LPW - 1;
  for (u64 sp = blockIdx.x; sp < nspans; sp += gridDim.x) {
    const u64 c0 = sp * (u64)(4 * TRIPS * U);            // first 1 KiB chunk of the span
    auto load = [&](Vec16<T> (&r)[U], int tl) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint4 *src = reinterpret_cast<const uint4 *>(vox + ((c0 + (u64)tl * U + u) * 64 + lane) * VPL);
        r[u].raw.x = __builtin_nontemporal_load(&src->x); r[u].raw.y = __builtin_nontemporal_load(&src->y);
        r[u].raw.z = __builtin_nontemporal_load(&src->z); r[u].raw.w = __builtin_nontemporal_load(&src->w);
      }
    };
    auto pack = [&](const Vec16<T> (&r)[U], int tl) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const u32 m = inside_bits<T>(r[u], iso);
        const u64 word = group_or<LPW>((u64)m << (sub * VPL));
        if (last) stage[(tl * U + u) * VPL + lane / LPW] = word;
      }
    };
#pragma unroll 1
    for (int i = 0; i < TRIPS; i++) {
      const int tl = i * 4 + wib;                        // the waves take the span's trips in turn
      Vec16<T> r[U];
      load(r, tl);
      pack(r, tl);
    }
    __syncthreads();
    const u64 w0 = sp * (u64)SPAN;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(bits + w0, 0, SPAN * 8, 0x00020000);
    for (int i = threadIdx.x * 2; i < SPAN; i += 512) {
      const u32x4 v = *reinterpret_cast<const u32x4 *>(&stage[i]);
      __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i * 8, 0, 16);          // aux 16 = sc1: write-through
      if (lgWordsPerSlice > 0) {
        if (v.x | v.y | v.z | v.w) sliceOcc[(w0 + i) >> lgWordsPerSlice] = 1u;  // both words lie in one slice
      } else if (lgWordsPerSlice == 0) {
        if (v.x | v.y) sliceOcc[w0 + i] = 1u;
        if (v.z | v.w) sliceOcc[w0 + i + 1] = 1u;
      }
    }
    __syncthreads();
  }
}

// The same sweep for rows that are NOT whole words (nx % 64 != 0 -- most real volumes -- or a pointer that is not 16-byte
// aligned): a workgroup still owns SPAN_WORDS consecutive OUTPUT words, i.e. a range of (row, word) pairs.  Rows follow
// each other in memory, so the voxels behind those words are one contiguous range too: it is thresholded exactly as above
// -- a flat stream of 16-byte vectors from the 16-byte boundary at or below its first voxel, 4 KiB trips per wave, DPP OR
// into 64-voxel words of the STREAM -- into the LDS stage; then every thread cuts two row words out of the staged stream
// (two LDS words, funnel-shifted, the row's last word masked to the voxels it has) and the span leaves as the same 16-byte
// write-through stores, occupancy folded in.  No flat scratch stream in memory, no second trip of the bits through L2, no
// repack and occupancy launches (round-4 review: 1000^3 f32 swept at 5.15 TB/s through those, against 6.7 for whole-word rows).
template <class T>
__global__ __launch_bounds__(256) void k_classify_span_rows(const T *__restrict__ vox, u64 *__restrict__ bits, u64 nspans,
                                                            u64 nwordsAll, int nx, int W, u32 ny, double isoD, long long isoI,
                                                            u32 *__restrict__ sliceOcc) {
  constexpr int U = 4;
  constexpr int VPL = 16 / sizeof(T);
  constexpr int LPW = 64 / VPL;
  // stream words of a span: SPAN_WORDS * 64 voxels at most, + the skew to the 16-byte boundary, rounded up to whole trips
  // (chunks past the stream's end repeat its last vector: never read back), + the word a funnel shift looks ahead
  constexpr int STAGE = SPAN_WORDS + (4 * U + 1) * VPL + 2;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) u64 stage[STAGE];
  const T iso = iso_as<T>(isoD, isoI);
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane % LPW;
  const bool last = sub == LPW - 1;
  const u32 qStep = 512u / (u32)W, rStep = 512u % (u32)W;
  for (u64 sp = blockIdx.x; sp < nspans; sp += gridDim.x) {
    // (the launcher takes this path for fewer than 2^32 words: rows and words of a span in 32-bit arithmetic)
    const u64 w0 = sp * (u64)SPAN_WORDS;
    const u64 left = nwordsAll - w0;
    const u32 nw = left < (u64)SPAN_WORDS ? (u32)left : (u32)SPAN_WORDS;
    const u32 r0 = (u32)w0 / (u32)W;
    const u32 k0 = (u32)w0 - r0 * (u32)W;
    const u32 wl = (u32)w0 + nw - 1, r1 = wl / (u32)W;
    const u32 k1 = wl - r1 * (u32)W;
    const u64 v0 = (u64)r0 * (u64)nx + (u64)k0 * 64;
    const u32 e1 = (k1 + 1) * 64u < (u32)nx ? (k1 + 1) * 64u : (u32)nx;
    const u32 nvox = (r1 - r0) * (u32)nx + e1 - k0 * 64u;                  // voxels behind this span's words
    const uintptr_t a0 = (uintptr_t)(vox + v0), ab = a0 & ~(uintptr_t)15;
    const u32 skew = (u32)((a0 - ab) / sizeof(T));
    const T *abase = reinterpret_cast<const T *>(ab);
    // (the first and the last vector may reach up to 15 bytes outside the range: the same 16-byte granule as valid voxels,
    //  so the loads cannot fault, and those bits are never used)
    const u32 nvec = (skew + nvox + VPL - 1) / VPL;
    const u32 n1k = (nvec + 63) / 64;                      // 1 KiB chunks of the stream
#pragma unroll 1
    for (u32 c = (u32)wib * U; c < n1k; c += 4 * U) {      // the waves take the stream's 4 KiB trips in turn
      Vec16<T> r[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        u32 vi = (c + u) * 64 + lane;
        vi = vi < nvec ? vi : nvec - 1;
        const uint4 *src = reinterpret_cast<const uint4 *>(abase + (size_t)vi * VPL);
        r[u].raw.x = __builtin_nontemporal_load(&src->x); r[u].raw.y = __builtin_nontemporal_load(&src->y);
        r[u].raw.z = __builtin_nontemporal_load(&src->z); r[u].raw.w = __builtin_nontemporal_load(&src->w);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const u32 m = inside_bits<T>(r[u], iso);
        const u64 word = group_or<LPW>((u64)m << (sub * VPL));
        if (last) stage[(c + u) * VPL + lane / LPW] = word;
      }
    }
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(bits + w0, 0, (int)(nw * 8u), 0x00020000);
    // slice of a row: a span seldom crosses a slice boundary (SPAN_WORDS / W rows); where slices are shorter than a span's
    // rows the division is taken per word
    const u32 s0 = r0 / ny;
    const u32 b1 = (s0 + 1) * ny;                          // first row of the next slice
    const bool manySlices = r1 - r0 + 1 > ny;
    u32 i = threadIdx.x * 2;
    u32 row = r0 + (k0 + i) / (u32)W;
    u32 k = (k0 + i) % (u32)W;
    auto cut = [&](u32 rw, u32 kk, bool live) -> u64 {
      if (!live) return 0ull;
      const u32 rel = skew + (rw - r0) * (u32)nx + kk * 64u - k0 * 64u;       // bit of the staged stream the word starts at
      const u32 j = rel >> 6, sh = rel & 63u;
      const u64 lo = stage[j], hi = stage[j + 1];
      u64 word = sh ? ((lo >> sh) | (hi << (64u - sh))) : lo;
      const int n = nx - (int)kk * 64;
      if (n < 64) word &= lowmask(n);
      return word;
    };
    auto mark = [&](u64 word, u32 rw) {
      if (!word) return;
      const u32 sl = manySlices ? rw / ny : s0 + (rw >= b1 ? 1u : 0u);
      sliceOcc[sl] = 1u;                                   // benign race: all store 1
    };
    for (; i < (u32)SPAN_WORDS; i += 512) {
      const u32 rowB = k + 1 < (u32)W ? row : row + 1;
      const u32 kB = k + 1 < (u32)W ? k + 1 : 0;
      const u64 a = cut(row, k, i < nw), b = cut(rowB, kB, i + 1 < nw);
      u32x4 v;
      v.x = (u32)a; v.y = (u32)(a >> 32); v.z = (u32)b; v.w = (u32)(b >> 32);
      __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i * 8, 0, 16);          // aux 16 = sc1: write-through; past nw: dropped
      mark(a, row);
      mark(b, rowB);
      k += rStep; row += qStep;
      if (k >= (u32)W) { k -= (u32)W; row++; }
    }
    __syncthreads();
  }
}

// Ragged rows (nx % 64 != 0), the fast way: k_classify_flat thresholds the slices as ONE flat stream of 16-byte
// vectors starting at the 16-byte boundary at or below the first voxel (`skew` voxels earlier); this kernel
// finishes the last, partial 1 KiB chunk ...
template <class T>
__global__ __launch_bounds__(64) void k_classify_tail(const T *__restrict__ abase, u64 *__restrict__ flat, u64 firstVec,
                                                      u64 nvec, double isoD, long long isoI) {
  constexpr int VPL = 16 / sizeof(T);
  constexpr int LPW = 64 / VPL;
  const T iso = iso_as<T>(isoD, isoI);
  const int lane = threadIdx.x & 63;
  const u64 v = firstVec + lane;
  u32 m = 0;
  if (v < nvec) {
    Vec16<T> r;
    r.raw = *reinterpret_cast<const uint4 *>(abase + v * VPL);
    m = inside_bits<T>(r, iso);
  }
  const int sub = lane % LPW;
  u64 part = (u64)m << (sub * VPL);
#pragma unroll
  for (int s = 1; s < LPW; s <<= 1) part |= __shfl_xor(part, s, 64);
  if (sub == 0) flat[(firstVec * VPL) / 64 + lane / LPW] = part;
}

// ... and this one cuts the flat bit stream into rows: word k of row r is 64 bits of the stream from bit
// skew + r*nx + 64k on (two flat words, funnel-shifted), masked to the voxels the row really has.
__global__ __launch_bounds__(256) void k_repack_rows(const u64 *__restrict__ flat, u64 *__restrict__ bits, int nx, int W,
                                                     u64 nrows, u64 skew) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 total = nrows * (u64)W;
  if (t >= total) return;
  u64 row;
  int k;
  if (total <= 0xffffffffull) { const u32 r = (u32)t / (u32)W; row = r; k = (int)((u32)t - r * (u32)W); }
  else { row = t / (u64)W; k = (int)(t % (u64)W); }
  const u64 sbit = skew + row * (u64)nx + (u64)k * 64;
  const int sh = (int)(sbit & 63);
  const u64 lo = flat[sbit >> 6], hi = flat[(sbit >> 6) + 1];
  u64 word = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
  const int n = nx - k * 64;
  if (n < 64) word &= lowmask(n);
  bits[t] = word;
}

// Generic path (any nx): one wave per (row, word); lane l tests voxel x = 64k + l; the
// wave ballot IS the packed word.
template <class T>
__global__ __launch_bounds__(256) void k_classify_rows(const T *__restrict__ vox, u64 *__restrict__ bits,
                                                       int nx, int W, u64 t0, u64 nrows, u64 rowsPerSlice, double isoD,
                                                       long long isoI, u32 *__restrict__ sliceOcc) {
  const T iso = iso_as<T>(isoD, isoI);
  const int lane = threadIdx.x & 63;
  const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
  const u64 total = nrows * (u64)W;
  for (u64 t = t0 + wave; t < total; t += nwaves) {
    const u64 row = t / W;
    const int k = (int)(t % W);
    const int x = k * 64 + lane;
    bool in = false;
    if (x < nx) in = !(vox[row * (u64)nx + x] < iso);
    const u64 word = __ballot(in);
    if (lane == 0) {
      bits[t] = word;
      if (word) sliceOcc[row / rowsPerSlice] = 1u;
    }
  }
}

// Per-slice occupancy (does the slice hold any inside voxel?) from the packed bits: one block per
// slice, stops at the first non-zero word it sees.
__global__ __launch_bounds__(256) void k_occupancy(const u64 *__restrict__ bits, size_t wordsPerSlice,
                                                   u32 *__restrict__ sliceOcc) {
  const u64 *p = bits + (size_t)blockIdx.x * wordsPerSlice;
  __shared__ int found;
  if (threadIdx.x == 0) found = 0;
  __syncthreads();
  for (size_t i0 = 0; i0 < wordsPerSlice; i0 += 1024) {
    u64 v = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const size_t i = i0 + j * 256 + threadIdx.x;
      if (i < wordsPerSlice) v |= p[i];
    }
    if (v) found = 1;
    __syncthreads();
    if (found) break;
  }
  if (threadIdx.x == 0) sliceOcc[blockIdx.x] = found ? 1u : 0u;
}

// Empty-slice aliasing (reference quirk Q1, txx:139-141 precede 156-161: the lookup planes are swapped only when
// an INSIDE voxel is met at a new z).  For a slice z that holds an inside voxel (every caller's does): when slice
// z-1 holds none and zp < z-1 is the previous occupied slice, bottom-plane corners of slice z are looked up among
// the top-plane corners of zp.  Returns zp or -1.  The usual answer costs one cached load.
// unknown (an ERRF_* bit, raised by the count kernel): the source lies in the buffer but below the counted range
// [cz0, ..) -- its ids belong to the rank below -- or the search fell off the bottom of a slab buffer that does not
// start at the volume's first slice (the multi-GPU driver knows whether anything is occupied down there).  Either
// way the rank below can hand the source slice over (Grid::extAlias, cuberille_recount).
// The ghost slice (cz0 when cz0 < oz0: counted for its vertex ids only, it emits no cells) needs nothing but the
// source's inside BITS -- which bottom corners it does not create -- so a source anywhere in the buffer serves it;
// only a source below the buffer has to be handed over, and then the bits alone.
__device__ __forceinline__ int alias_of(const u32 *__restrict__ occ, const Grid &g, int q1, int z, u32 &unknown) {
  unknown = 0;
  if (!q1 || z <= 0 || occ[z - 1]) return -1;
  int p = z - 2;
  while (p >= 0 && !occ[p]) p--;
  if (p >= 0 && (p >= g.cz0 || z < g.oz0)) return p;
  if (p < 0 && g.zglob0 == 0) return -1;
  if (g.extAlias) return g.nzb;        // the source slice came from a rank below: it sits one past the buffer
  unknown = p >= 0 ? ERRF_ALIAS_UNKNOWN : ERRF_ALIAS_BELOW_BUFFER;
  return -1;
}

// Inclusive prefix sum over the 64 lanes of a wave on the DPP data path (no LDS round trips, unlike __shfl_up):
// doubling steps inside each row of 16 lanes, then the row totals carried across with row_bcast 15 / 31.
__device__ __forceinline__ u32 wave_inclusive_sum(u32 v) {
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}

// ---------------------------------------------------------------------------------------------
// Word classification: everything a 64-voxel word needs, from the 27 neighbour bit-rows.
// ---------------------------------------------------------------------------------------------
// 3-input boolean function of 64-bit operands on the gfx950 v_bitop3_b32 (truth table TT: bit (a<<2 | b<<1 | c))
template <int TT>
__device__ __forceinline__ u64 bop3(u64 a, u64 b, u64 c) {
  const u32 lo = __builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, TT);
  const u32 hi = __builtin_amdgcn_bitop3_b32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), TT);
  return (u64)lo | ((u64)hi << 32);
}
constexpr int TT_AND3 = 0x80, TT_OR3 = 0xfe, TT_A_ANDN_B = 0x30;   // a&b&c, a|b|c, a&~b

struct Rows3 { u64 m, c, p; };   // bit x = inside(x-1), inside(x), inside(x+1), border-clamped (I2)

__device__ __forceinline__ u64 valid_mask(const Grid &g, int k) {
  return (k == g.W - 1 && g.lastpos != 63) ? ((2ull << g.lastpos) - 1ull) : ~0ull;
}

// flat word index (counted range) -> word k of row y of local slice z.  Power-of-two rows take shifts;
// otherwise one 32-bit division pair when the index fits (64-bit division costs more than the whole
// face test of a word).
__device__ __forceinline__ void word_coords(const Grid &g, size_t gi, int &k, int &y, int &z) {
  if (g.wShift >= 0 && g.yShift >= 0) {
    k = (int)(gi & (size_t)(g.W - 1));
    const size_t row = gi >> g.wShift;
    y = (int)(row & (size_t)(g.ny - 1));
    z = g.cz0 + (int)(row >> g.yShift);
  } else if (gi <= 0xffffffffull) {
    const u32 row = (u32)gi / (u32)g.W;
    k = (int)((u32)gi - row * (u32)g.W);
    const u32 zz = row / (u32)g.ny;
    y = (int)(row - zz * (u32)g.ny);
    z = g.cz0 + (int)zz;
  } else {
    const size_t row = gi / g.W;
    k = (int)(gi % g.W);
    y = (int)(row % g.ny);
    z = g.cz0 + (int)(row / g.ny);
  }
}

// A word inside the bit volume and the (clamped) word offsets of its neighbours: the neighbour rows are the word's
// own address plus an offset that is 0 at the image border (ZeroFluxNeumann: the clamped neighbour is the row itself).
struct WordPos {
  const u64 *q;                 // &bits[(z*ny + y)*W + k]
  bool global;                  // q points into the bit volume itself (not into a staged copy): see centre_row
  int k, y, z;
  long long km, kp;             // -1 / +1, or 0 at the row ends
  long long yo[3], zo[3];       // word offsets of rows y-1, y, y+1 and slices z-1, z, z+1 (clamped)
};

__device__ __forceinline__ WordPos word_pos(const u64 *__restrict__ bits, const Grid &g, int y, int z, int k) {
  WordPos w;
  w.q = bits + ((size_t)z * g.ny + y) * g.W + k;
  w.global = true;
  w.k = k; w.y = y; w.z = z;
  w.km = k > 0 ? -1 : 0;
  w.kp = k < g.W - 1 ? 1 : 0;
  const long long rs = g.W, ss = (long long)g.W * g.ny;
  w.yo[0] = y > 0 ? -rs : 0;          w.yo[1] = 0;  w.yo[2] = y < g.ny - 1 ? rs : 0;
  w.zo[0] = z > 0 ? -ss : 0;          w.zo[1] = 0;  w.zo[2] = z < g.nzb - 1 ? ss : 0;
  return w;
}

// The same word inside a staged copy of the bit rows (the count kernel's LDS tile): `q` points at the word's copy, rows
// are W words apart as in memory, the copies of the slices z-1 / z+1 sit `below` / `above` words from it.
__device__ __forceinline__ WordPos word_pos_tile(const u64 *q, long long below, long long above, const Grid &g, int y, int z, int k) {
  WordPos w;
  w.q = q;
  w.global = false;
  w.k = k; w.y = y; w.z = z;
  w.km = k > 0 ? -1 : 0;
  w.kp = k < g.W - 1 ? 1 : 0;
  const long long rs = g.W;
  w.yo[0] = y > 0 ? -rs : 0;          w.yo[1] = 0;  w.yo[2] = y < g.ny - 1 ? rs : 0;
  w.zo[0] = z > 0 ? below : 0;        w.zo[1] = 0;  w.zo[2] = z < g.nzb - 1 ? above : 0;
  return w;
}

// A word of the bit volume together with the two bits of its row neighbours that border it -- bit 63 of the word before,
// bit 0 of the word behind -- in ONE 16-byte load: the four dwords from the upper half of the word before on (the kernels
// that look at words are bound by the number of their vector-memory instructions, and this was three of them).  The
// first word of a row starts the load at itself instead (its left neighbour is never looked at, and the very first word
// of the volume has nothing before it); the word behind the last one of the buffer is the spare slice count_prepare
// reserves.  4-byte aligned, like every dword of the volume.
__device__ __forceinline__ void centre_row(const u64 *r, int k, u64 &c, u32 &prevHi, u32 &nextLo) {
  // (ONE vector load of four dwords that is only 4-byte aligned: global_load_dwordx4 takes any dword address; as four
  //  scalar loads the compiler, which knows the 8-byte phase, cuts them into dword + dwordx2 + dword)
  typedef u32 Dwords4 __attribute__((ext_vector_type(4)));
  typedef Dwords4 __attribute__((aligned(4))) Dwords4A4;
  const u32 *p = reinterpret_cast<const u32 *>(r) - (k > 0 ? 1 : 0);
  const Dwords4 v = *reinterpret_cast<const Dwords4A4 *>(p);
  const u32 v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
  if (k > 0) { prevHi = v0; c = (u64)v1 | ((u64)v2 << 32); nextLo = v3; }
  else { prevHi = 0u; c = (u64)v0 | ((u64)v1 << 32); nextLo = v2; }
}

// the row at word offset `off` from the word (a staged copy keeps its three separate reads: LDS)
__device__ __forceinline__ Rows3 load_row(const WordPos &w, const Grid &g, long long off) {
  const u64 *r = w.q + off;
  Rows3 o;
  if (w.global) {
    u64 c;
    u32 prevHi, nextLo;
    centre_row(r, w.k, c, prevHi, nextLo);
    o.c = c;
    o.m = (c << 1) | (w.k > 0 ? (u64)(prevHi >> 31) : (c & 1ull));
    o.p = (c >> 1) | (w.k < g.W - 1 ? ((u64)(nextLo & 1u) << 63) : (c & (1ull << g.lastpos)));
    return o;
  }
  const u64 c = r[0], wp = r[w.km], wn = r[w.kp];
  o.c = c;
  o.m = (c << 1) | (w.k > 0 ? (wp >> 63) : (c & 1ull));
  o.p = (c >> 1) | (w.k < g.W - 1 ? (wn << 63) : (c & (1ull << g.lastpos)));
  return o;
}

struct WordInfo {
  u64 F[6];   // F[f] bit x: voxel x emits a quad on face f                  (txx:164-173)
  u64 C[8];   // C[i] bit x: voxel x is the first to need its corner i, i.e. the reference
              //             would call AddVertex for it here                 (txx:179-194)
};

struct Neigh {
  u64 raw[3][3][3];  // [dz+1][dy+1][dx+1] bit x: inside(x+dx, y+dy, z+dz), every coordinate clamped into the image
  u64 exm, exp_;     // bit x: voxel x-1 / x+1 exists (all ones except at the two ends of a row)
  u32 exy[3], exz[3];  // all ones when row y+dy / slice z+dz exists, else 0 (only read on waves that touch a y/z border)
};

// block member E (code x|y<<1|z<<2) of the 2x2x2 block around corner D of voxel x sits at offset D - E per axis.
// It activates the corner when it exists, is inside, and at least one of its three face neighbours inside the
// block is outside: two 3-input operations per 32-bit half.  The x existence is
// applied by the caller to whole groups of terms (it only differs from all-ones at the row ends).
template <bool YZ, int D, int E>
__device__ __forceinline__ u64 act(const Neigh &n) {
  constexpr int sx = (D & 1) - (E & 1), sy = ((D >> 1) & 1) - ((E >> 1) & 1), sz = (D >> 2) - (E >> 2);
  constexpr int nx_ = (D & 1) - ((E ^ 1) & 1), ny_ = ((D >> 1) & 1) - (((E ^ 2) >> 1) & 1), nz_ = (D >> 2) - ((E ^ 4) >> 2);
  const u64 t = bop3<TT_AND3>(n.raw[sz + 1][sy + 1][nx_ + 1], n.raw[sz + 1][ny_ + 1][sx + 1], n.raw[nz_ + 1][sy + 1][sx + 1]);
  u64 a = bop3<TT_A_ANDN_B>(n.raw[sz + 1][sy + 1][sx + 1], t, t);
  if (YZ && (sy != 0 || sz != 0)) {
    const u32 ex = n.exy[sy + 1] & n.exz[sz + 1];
    a &= (u64)ex | ((u64)ex << 32);
  }
  return a;
}
// OR of act<D,E'> over E' = E..7 with E' & 1 == PAR (members with a larger code come earlier in raster order:
// smaller coordinates).  PAR == D & 1: members in the voxel's own x column (offset 0 in x); the others sit at
// x-1 (D even) or x+1 (D odd).
template <bool YZ, int D, int E, int PAR>
__device__ __forceinline__ u64 act_or(const Neigh &n) {
  if constexpr (E > 7) return 0ull;
  else if constexpr ((E & 1) != PAR) return act_or<YZ, D, E + 1, PAR>(n);
  else if constexpr (E + 2 > 7) return act<YZ, D, E>(n);
  else if constexpr (E + 4 > 7) return act<YZ, D, E>(n) | act<YZ, D, E + 2>(n);
  else return bop3<TT_OR3>(act<YZ, D, E>(n), act<YZ, D, E + 2>(n), act_or<YZ, D, E + 4, PAR>(n));
}
template <bool YZ, int I>
__device__ __forceinline__ u64 created(const Neigh &n) {
  constexpr int D = (I == 2) ? 3 : (I == 3) ? 2 : (I == 6) ? 7 : (I == 7) ? 6 : I;
  const u64 own = act<YZ, D, D>(n);
  if constexpr (D == 7) return own;
  else {
    const u64 same = act_or<YZ, D, D + 1, (D & 1)>(n);
    const u64 other = act_or<YZ, D, D + 1, 1 - (D & 1)>(n) & ((D & 1) ? n.exp_ : n.exm);
    return bop3<0x10>(own, same, other);                 // own & ~(same | other)
  }
}

// YZ: some lane of the wave sits on a y or z border of the buffer (then rows off the image must not activate
// anything); interior waves skip those masks.
// Loads: the nine words of the neighbourhood's rows, and the centre row's two x neighbours.  The x neighbours of the
// other eight rows each contribute ONE bit (the carry into voxel 0 / 63), and those two voxels only matter when they
// emit a face themselves (a voxel without a face creates no corner): the sixteen loads are issued for those lanes
// only.  Returns the word's any-face mask; the caller ANDs it into the created masks, which also wipes whatever
// the missing carries left in bits 0 and 63.
// bit x = inside(x-1) / inside(x+1) of a row word c, on 32-bit halves (v_alignbit + v_lshl_or: five instructions for
// both shifted rows instead of two 64-bit shifts, two selects and two ORs); carryM / carryP are the neighbour words'
// edge bits (0 or 1)
__device__ __forceinline__ void shifted_rows(u64 c, u32 carryM, u32 carryP, u64 &m, u64 &p) {
  const u32 lo = (u32)c, hi = (u32)(c >> 32);
  const u32 mlo = (lo << 1) | carryM, mhi = __builtin_amdgcn_alignbit(hi, lo, 31);
  const u32 plo = __builtin_amdgcn_alignbit(hi, lo, 1), phi = (hi >> 1) | (carryP << 31);
  m = (u64)mlo | ((u64)mhi << 32);
  p = (u64)plo | ((u64)phi << 32);
}

template <bool YZ>
__device__ __forceinline__ u64 load_neigh(const WordPos &w, const Grid &g, Neigh &n) {
  const u64 valid = valid_mask(g, w.k);
  n.exm = (w.k == 0) ? ~1ull : ~0ull;
  n.exp_ = (w.k == g.W - 1) ? (valid >> 1) : ~0ull;
  if (YZ) {
    n.exy[0] = w.y > 0 ? ~0u : 0u;            n.exy[1] = ~0u;  n.exy[2] = w.y < g.ny - 1 ? ~0u : 0u;
    n.exz[0] = w.z > 0 ? ~0u : 0u;            n.exz[1] = ~0u;  n.exz[2] = w.z < g.nzb - 1 ? ~0u : 0u;
  }
  u64 c[3][3];
  u32 wpHi11, wnLo11;
  const u32 *q32 = reinterpret_cast<const u32 *>(w.q);
#pragma unroll
  for (int dz = 0; dz < 3; dz++)
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
      if (dz != 1 || dy != 1) c[dz][dy] = w.q[w.yo[dy] + w.zo[dz]];
  // the centre row's edge bits: always (they decide whether voxel 0 / 63 emits a face); with the word itself in one load
  if (w.global) {
    centre_row(w.q, w.k, c[1][1], wpHi11, wnLo11);
  } else {
    c[1][1] = w.q[0];
    wpHi11 = q32[2 * w.km + 1];
    wnLo11 = q32[2 * w.kp];
  }
  const bool first = w.k == 0, last = w.k == g.W - 1;
  const bool ragged = g.lastpos != 63;           // uniform: the last word of a row is partly filled
  const u64 lastbit = 1ull << g.lastpos;
  auto row = [&](u64 cc, u32 wpHi, u32 wnLo, u64 &m, u64 &p) {
    // border clamp (I2): off the row's ends the neighbour is the end voxel itself
    const u32 carryM = first ? ((u32)cc & 1u) : (wpHi >> 31);
    const u32 carryP = last ? (ragged ? 0u : (u32)(cc >> 63)) : (wnLo & 1u);
    shifted_rows(cc, carryM, carryP, m, p);
    if (ragged && last) p |= cc & lastbit;
  };
  u64 m11, p11;
  row(c[1][1], wpHi11, wnLo11, m11, p11);
  const u64 anyFace = c[1][1] & ~(m11 & p11 & c[1][0] & c[1][2] & c[0][1] & c[2][1]);
  const bool needM = (anyFace & 1ull) && !first, needP = (anyFace >> 63) && !last;
  u32 wpHi[3][3], wnLo[3][3];
#pragma unroll
  for (int dz = 0; dz < 3; dz++)
#pragma unroll
    for (int dy = 0; dy < 3; dy++) { wpHi[dz][dy] = 0; wnLo[dz][dy] = 0; }
  if (needM) {
#pragma unroll
    for (int dz = 0; dz < 3; dz++)
#pragma unroll
      for (int dy = 0; dy < 3; dy++)
        if (dz != 1 || dy != 1) wpHi[dz][dy] = q32[2 * (w.yo[dy] + w.zo[dz] - 1) + 1];
  }
  if (needP) {
#pragma unroll
    for (int dz = 0; dz < 3; dz++)
#pragma unroll
      for (int dy = 0; dy < 3; dy++)
        if (dz != 1 || dy != 1) wnLo[dz][dy] = q32[2 * (w.yo[dy] + w.zo[dz] + 1)];
  }
#pragma unroll
  for (int dz = 0; dz < 3; dz++)
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
      u64 m, p;
      if (dz == 1 && dy == 1) { m = m11; p = p11; }
      else row(c[dz][dy], wpHi[dz][dy], wnLo[dz][dy], m, p);
      n.raw[dz][dy][0] = m; n.raw[dz][dy][1] = c[dz][dy]; n.raw[dz][dy][2] = p;
    }
  return anyFace;
}

// AE[i] (i = 0..3) bit x: bottom corner i of voxel x already exists as a top-plane corner of the
// aliased source slice zp (any existing inside voxel of slice zp touching that (x,y) corner).
__device__ __forceinline__ void alias_exists(const WordPos &w, const Grid &g, int zp, u64 AE[4]) {
  const u64 valid = valid_mask(g, w.k);
  const u64 exX[3] = {(w.k == 0) ? ~1ull : ~0ull, valid, (w.k == g.W - 1) ? (valid >> 1) : ~0ull};
  const u64 exY[3] = {(w.y > 0) ? ~0ull : 0ull, ~0ull, (w.y < g.ny - 1) ? ~0ull : 0ull};
  const long long toZp = (long long)(zp - w.z) * g.W * g.ny;
  Rows3 r[3];
#pragma unroll
  for (int dy = 0; dy < 3; dy++) r[dy] = load_row(w, g, toZp + w.yo[dy]);
  // corner offsets (dx,dy) of corners 0..3: (0,0) (1,0) (1,1) (0,1); voxels (x+dx-ex, y+dy-ey)
  auto at = [&](int sx, int sy) -> u64 {
    const Rows3 &rr = r[sy + 1];
    const u64 b = sx < 0 ? rr.m : (sx == 0 ? rr.c : rr.p);
    return b & exX[sx + 1] & exY[sy + 1];
  };
  AE[0] = at(0, 0) | at(-1, 0) | at(0, -1) | at(-1, -1);
  AE[1] = at(1, 0) | at(0, 0) | at(1, -1) | at(0, -1);
  AE[2] = at(1, 1) | at(0, 1) | at(1, 0) | at(0, 0);
  AE[3] = at(0, 1) | at(-1, 1) | at(0, 0) | at(-1, 0);
}

// The eight "voxel x creates its corner i" masks of one word (and, FACES, its six face masks).
// `unknown`: ERRF_* bits of alias_of for the count kernel to raise; the other callers ignore them.
// wp: where the word's rows are read from (the bit volume, or a staged copy of it); bits: the bit volume itself, for
// the aliased source slice of quirk Q1 (any slice of the buffer: never staged)
template <bool FACES>
__device__ __forceinline__ void classify_word_at(const WordPos &wp, const u64 *__restrict__ bits, const u32 *__restrict__ occ,
                                                 const Grid &g, int q1, WordInfo &w, u32 &unknown) {
  const int y = wp.y, z = wp.z;
  const bool yzBorder = y == 0 || y == g.ny - 1 || z == 0 || z == g.nzb - 1;
  const bool anyBorder = __ballot(yzBorder) != 0ull;      // uniform over the lanes that are here
  Neigh n;
  u64 anyFace;
  if (anyBorder) {
    anyFace = load_neigh<true>(wp, g, n);
    w.C[0] = created<true, 0>(n); w.C[1] = created<true, 1>(n); w.C[2] = created<true, 2>(n); w.C[3] = created<true, 3>(n);
    w.C[4] = created<true, 4>(n); w.C[5] = created<true, 5>(n); w.C[6] = created<true, 6>(n); w.C[7] = created<true, 7>(n);
  } else {
    anyFace = load_neigh<false>(wp, g, n);
    w.C[0] = created<false, 0>(n); w.C[1] = created<false, 1>(n); w.C[2] = created<false, 2>(n); w.C[3] = created<false, 3>(n);
    w.C[4] = created<false, 4>(n); w.C[5] = created<false, 5>(n); w.C[6] = created<false, 6>(n); w.C[7] = created<false, 7>(n);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) w.C[i] &= anyFace;          // a voxel creates corners only on faces it emits
  if (FACES) {
    const u64 I = n.raw[1][1][1];
    w.F[0] = I & ~n.raw[1][1][0];   // -x   (offsets of txx:122-127)
    w.F[1] = I & ~n.raw[1][0][1];   // -y
    w.F[2] = I & ~n.raw[1][1][2];   // +x
    w.F[3] = I & ~n.raw[1][2][1];   // +y
    w.F[4] = I & ~n.raw[0][1][1];   // -z
    w.F[5] = I & ~n.raw[2][1][1];   // +z
  }
  const int zp = alias_of(occ, g, q1, z, unknown);
  if (zp >= 0) {
    u64 AE[4];
    alias_exists(word_pos(bits, g, y, z, wp.k), g, zp, AE);
#pragma unroll
    for (int i = 0; i < 4; i++) w.C[i] &= ~AE[i];
  }
}

template <bool FACES>
__device__ __forceinline__ void classify_word(const u64 *__restrict__ bits, const u32 *__restrict__ occ, const Grid &g, int q1,
                                              int y, int z, int k, WordInfo &w, u32 &unknown) {
  classify_word_at<FACES>(word_pos(bits, g, y, z, k), bits, occ, g, q1, w, unknown);
}

// the six face masks of a word only (7 bit-rows instead of 27)
__device__ __forceinline__ void faces_at(const WordPos &w, const Grid &g, u64 F[6]) {
  const Rows3 c = load_row(w, g, 0);
  const u64 ym = w.q[w.yo[0]], yp = w.q[w.yo[2]], zm = w.q[w.zo[0]], zp = w.q[w.zo[2]];
  F[0] = c.c & ~c.m; F[1] = c.c & ~ym; F[2] = c.c & ~c.p; F[3] = c.c & ~yp; F[4] = c.c & ~zm; F[5] = c.c & ~zp;
}
__device__ __forceinline__ void faces_word(const u64 *__restrict__ bits, const Grid &g, int y, int z, int k, u64 F[6]) {
  faces_at(word_pos(bits, g, y, z, k), g, F);
}

// ---------------------------------------------------------------------------------------------
// The corner logic per LATTICE corner (the dense form of the count, k_count_dense).
// Everything about the lattice corner (X, Y, Z) is a function of the 2x2x2 block of voxels around it; voxel x sees that
// corner as its corner (0,dy,dz) when X = x and as its corner (1,dy,dz) when X = x + 1.  The per-voxel form above
// evaluates the block once for each of the two (36 activation terms per word); here a word evaluates the four corner
// rows (dy,dz) of the lattice positions X = 64k .. 64k+63 ONCE and reads both answers off them.  With b_E the inside bit
// of block member E (code ex | ey<<1 | ez<<2: the voxel at X - ex, rows y+dy-ey, z+dz-ez; a larger code comes earlier in
// raster order) and T_E = b_E & ~(b_E^1 & b_E^2 & b_E^4) "member E activates the corner" (it is inside and has a face
// inside the block), D = (0,dy,dz):
//   P(X) = T_D   & ~OR_{E>D}   T_E          voxel X     creates its corner D   = (0,dy,dz)
//   R(X) = T_D+1 & ~OR_{E>D+1} T_E          voxel X - 1 creates its corner D+1 = (1,dy,dz)
// Only voxels X - 1 and X are looked at: the rows `c` and the rows shifted up by one bit `m` (no x+1 rows).
// Away from the image border the chains collapse.  "No member of H = {E >= k} activates" means every inside member of H
// has its three neighbours inside; H is connected for k = 2 and k = 4 and every member of the block has a neighbour in
// it, so either no member of H is inside or all eight are (and then nobody activates):
//   D = 0:  P = b0 & ~(b1 | b2 | ... | b7)                   R = b1 & ~(b2 | ... | b7)
//   D = 2:  P = b2 & ~(b3 | b4 | b5 | b6 | b7)               R = b3 & ~(b4 | b5 | b6 | b7)
//   D = 4:  P = b4 & ~(b5|b6|b7)  |  b2..b7 & b1 & ~b0       R = b5 & ~(b6|b7)  |  b2..b7 & ~b1
//   D = 6:  P = b6 & (~b7 | b5 & b3 & ~(b4 & b2))            R = b7 & ~(b6 & b5 & b3)
// (all 256 blocks x 4 rows checked against the chains: tests/test_host.py) -- 23 three-input operations per 32 corners
// instead of 72 for the 36 terms.  On the border of the image members that do not exist must not activate, and a
// clamped copy of a row would: waves that touch a y or z border take the chains with the existence masks (lat_row_chain);
// the two x borders are one bit per row each -- P at X = 0 (no voxel -1), R at X = nx (no voxel nx) -- and a pass of
// their own over the block's rows (border_nibs) supplies them: no word evaluates them in passing.
// What a word lacks besides is R(64k+64) for its voxel 63 -- bit 0 of the NEXT word's R, fetched from that word's lane
// through LDS.
// ---------------------------------------------------------------------------------------------
template <int DY, int DZ, int E>
__device__ __forceinline__ u64 lat_b(const u64 (&c)[3][3], const u64 (&m)[3][3]) {
  return (E & 1) ? m[DZ - (E >> 2) + 1][DY - ((E >> 1) & 1) + 1] : c[DZ - (E >> 2) + 1][DY - ((E >> 1) & 1) + 1];
}
constexpr int TT_A_ANDN_BC = 0x10, TT_A_OR_B_ANDN_C = 0xf4, TT_AB_ANDN_C = 0x40, TT_A_AND_NB_OR_C = 0xb0;
// interior corner rows: the closed forms
template <int DY, int DZ>
__device__ __forceinline__ void lat_row_closed(const u64 (&c)[3][3], const u64 (&m)[3][3], u64 &P, u64 &R) {
  const u64 b0 = lat_b<DY, DZ, 0>(c, m), b1 = lat_b<DY, DZ, 1>(c, m), b2 = lat_b<DY, DZ, 2>(c, m), b3 = lat_b<DY, DZ, 3>(c, m),
            b4 = lat_b<DY, DZ, 4>(c, m), b5 = lat_b<DY, DZ, 5>(c, m), b6 = lat_b<DY, DZ, 6>(c, m), b7 = lat_b<DY, DZ, 7>(c, m);
  constexpr int D = (DY << 1) | (DZ << 2);
  if constexpr (D == 0) {
    const u64 z = bop3<TT_OR3>(b2, b3, b4) | bop3<TT_OR3>(b5, b6, b7);
    R = b1 & ~z;
    P = bop3<TT_A_ANDN_BC>(b0, b1, z);
  } else if constexpr (D == 2) {
    const u64 x = bop3<TT_OR3>(b4, b5, b6);
    R = bop3<TT_A_ANDN_BC>(b3, x, b7);
    P = bop3<TT_A_ANDN_BC>(b2, b3, x | b7);
  } else if constexpr (D == 4) {
    const u64 w = bop3<TT_AND3>(b2, b3, b4) & bop3<TT_AND3>(b5, b6, b7);
    R = bop3<TT_A_OR_B_ANDN_C>(bop3<TT_A_ANDN_BC>(b5, b6, b7), w, b1);
    P = (b4 & ~bop3<TT_OR3>(b5, b6, b7)) | bop3<TT_AB_ANDN_C>(w, b1, b0);
  } else {
    R = b7 & ~bop3<TT_AND3>(b6, b5, b3);
    P = bop3<TT_A_AND_NB_OR_C>(b6, b7, bop3<TT_AB_ANDN_C>(b5, b3, b4 & b2));
  }
}
// the chains, with the existence of every member: T_E ...
template <int DY, int DZ, int E>
__device__ __forceinline__ u64 lat_T(const u64 (&c)[3][3], const u64 (&m)[3][3], const u32 (&exy)[3], const u32 (&exz)[3]) {
  constexpr int sy = DY - ((E >> 1) & 1), sz = DZ - (E >> 2);                 // the member's row
  const u64 t = bop3<TT_AND3>(lat_b<DY, DZ, E ^ 1>(c, m), lat_b<DY, DZ, E ^ 2>(c, m), lat_b<DY, DZ, E ^ 4>(c, m));
  const u64 mem = lat_b<DY, DZ, E>(c, m);
  u64 a = bop3<TT_A_ANDN_B>(mem, t, t);
  if (sy != 0 || sz != 0) {
    const u32 e = exy[sy + 1] & exz[sz + 1];
    a &= (u64)e | ((u64)e << 32);
  }
  return a;
}
// ... their OR over E = E0, E0+2, ... <= 7 (one parity of x: the members at X for even E, at X-1 for odd E) ...
template <int DY, int DZ, int E0>
__device__ __forceinline__ u64 lat_or(const u64 (&c)[3][3], const u64 (&m)[3][3], const u32 (&exy)[3], const u32 (&exz)[3]) {
  if constexpr (E0 > 7) return 0ull;
  else if constexpr (E0 + 2 > 7) return lat_T<DY, DZ, E0>(c, m, exy, exz);
  else if constexpr (E0 + 4 > 7) return lat_T<DY, DZ, E0>(c, m, exy, exz) | lat_T<DY, DZ, E0 + 2>(c, m, exy, exz);
  else return bop3<TT_OR3>(lat_T<DY, DZ, E0>(c, m, exy, exz), lat_T<DY, DZ, E0 + 2>(c, m, exy, exz), lat_or<DY, DZ, E0 + 4>(c, m, exy, exz));
}
// ... and P and R of one corner row (every voxel X and X - 1 taken to exist: the x borders are not evaluated here)
template <int DY, int DZ>
__device__ __forceinline__ void lat_row_chain(const u64 (&c)[3][3], const u64 (&m)[3][3], const u32 (&exy)[3], const u32 (&exz)[3], u64 &P, u64 &R) {
  constexpr int D = (DY << 1) | (DZ << 2);
  const u64 s = lat_or<DY, DZ, D + 3>(c, m, exy, exz) | lat_or<DY, DZ, D + 2>(c, m, exy, exz);
  const u64 t1 = lat_T<DY, DZ, D + 1>(c, m, exy, exz);
  const u64 t0 = lat_T<DY, DZ, D>(c, m, exy, exz);
  P = bop3<TT_A_ANDN_BC>(t0, t1, s);
  R = t1 & ~s;
}

// Where a word of the dense count reads its rows: the staged copy of the bit rows (k_count_dense), word index `idx`,
// rows W words apart, the copies of the slices below / above `below` / `above` words away
struct DenseAt {
  const u64 *tile;
  int idx, W, below, above;
  int k, y, z;
};

// One word of the dense count: its quads and the vertices its voxels create as V | Q<<16 -- all but the (at most four)
// that voxel 63 creates at the lattice corners X = 64k+64 and, in a row's first word, those that voxel 0 creates at
// X = 0 -- and `nibs`: how many of the four corner rows have R set at X = 64k, i.e. the vertices the word BEFORE this one
// lacks (already taken off this word's count: bit 0 of R is that word's voxel 63).
// YZ: the chains with the existence masks (waves on a y or z border); zp: quirk Q1's source slice (or -1).
template <bool YZ, bool NOCORNER = false>
__device__ __forceinline__ void lattice_word(const DenseAt &w, const u64 *__restrict__ bits, const Grid &g, int zp, u32 &packed, u32 &nibs) {
  u64 c[3][3], m[3][3];
  u32 exy[3] = {~0u, ~0u, ~0u}, exz[3] = {~0u, ~0u, ~0u};
  int yo[3] = {-w.W, 0, w.W}, zo[3] = {w.below, 0, w.above};
  if (YZ) {
    if (w.y == 0) { exy[0] = 0u; yo[0] = 0; }
    if (w.y == g.ny - 1) { exy[2] = 0u; yo[2] = 0; }
    if (w.z == 0) { exz[0] = 0u; zo[0] = 0; }
    if (w.z == g.nzb - 1) { exz[2] = 0u; zo[2] = 0; }
  }
  const bool first = w.k == 0;
#pragma unroll
  for (int dz = 0; dz < 3; dz++)
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
      const int off = w.idx + yo[dy] + zo[dz];
      // the word and the word before it (one two-word LDS read).  Before a row's first word that is the last word of the
      // row before: bit 0 of `m` is then meaningless, and so is bit 0 of P and R -- which the x-border pass supplies
      const u64 cc = w.tile[off], pw = w.tile[off - 1];
      const u32 lo = (u32)cc, hi = (u32)(cc >> 32);
      c[dz][dy] = cc;
      m[dz][dy] = (u64)__builtin_amdgcn_alignbit(lo, (u32)(pw >> 32), 31) | ((u64)__builtin_amdgcn_alignbit(hi, lo, 31) << 32);
    }
  const u64 I = c[1][1];
  u64 P[4] = {0, 0, 0, 0}, R[4] = {0, 0, 0, 0};
  if (!NOCORNER) {
    // every P is a subset of voxel X's own bits (it is member D of each corner row): without bit 0 of a first word there
    // (R's bit 0 stays meaningless there: nobody reads a first word's `nibs`)
    c[1][1] = first ? (I & ~1ull) : I;
    if (YZ) {
      lat_row_chain<0, 0>(c, m, exy, exz, P[0], R[0]);
      lat_row_chain<1, 0>(c, m, exy, exz, P[1], R[1]);
      lat_row_chain<0, 1>(c, m, exy, exz, P[2], R[2]);
      lat_row_chain<1, 1>(c, m, exy, exz, P[3], R[3]);
    } else {
      lat_row_closed<0, 0>(c, m, P[0], R[0]);
      lat_row_closed<1, 0>(c, m, P[1], R[1]);
      lat_row_closed<0, 1>(c, m, P[2], R[2]);
      lat_row_closed<1, 1>(c, m, P[3], R[3]);
    }
  }
  if (zp >= 0) {
    // quirk Q1: a bottom corner (dz = 0) that exists as a top-plane corner of the aliased source slice zp -- an existing
    // inside voxel of that slice touches the lattice corner (X, y+dy) -- is looked up, not created
    const u64 *src = bits + ((size_t)zp * g.ny + w.y) * g.W + w.k;
    u64 any[3];
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
      const u64 *r = src + yo[dy];
      const u64 cc = r[0];
      any[dy] = (cc | (cc << 1) | (first ? 0ull : (r[-1] >> 63))) & ((u64)exy[dy] | ((u64)exy[dy] << 32));
    }
    const u64 L0 = any[0] | any[1], L1 = any[1] | any[2];
    P[0] &= ~L0; R[0] &= ~L0; P[1] &= ~L1; R[1] &= ~L1;
  }
  nibs = ((u32)R[0] & 1u) + ((u32)R[1] & 1u) + ((u32)R[2] & 1u) + ((u32)R[3] & 1u);
  const bool last = w.k == w.W - 1;
  if (last && g.lastpos != 63) {
    // a row that ends inside its last word: the corners X = nx are bit lastpos + 1 of this word's R -- evaluated with the
    // zero bits of the voxels that do not exist, and the x-border pass's to supply, like those behind a whole last word
    const u64 keep = ~(2ull << g.lastpos);
#pragma unroll
    for (int r = 0; r < 4; r++) R[r] &= keep;
  }
  const u64 nw = w.tile[w.idx + (last ? 0 : 1)];
  const u64 p11 = (I >> 1) | (last ? (I & (1ull << g.lastpos)) : (nw << 63));      // (the clamped neighbour of the row's last voxel: itself)
  // (at X = 0 the face test reads the clamped neighbour: voxel 0 itself)
  const u64 m11 = first ? (m[1][1] | 1ull) : m[1][1];
  const int nQ = popc64(I & ~m11) + popc64(I & ~c[1][0]) + popc64(I & ~p11) + popc64(I & ~c[1][2]) + popc64(I & ~c[0][1]) + popc64(I & ~c[2][1]);
  // (a member that activates has a face: no mask is "and"ed with the word's faces)
  int nV = -(int)nibs;
#pragma unroll
  for (int r = 0; r < 4; r++) nV += popc64(P[r]) + popc64(R[r]);
  packed = (u32)nV | ((u32)nQ << 16);
}

// The two x borders of a row, one bit per corner row each: the corners X = 0 of voxel 0 (AT_START: no voxel -1 exists,
// the answer is P) and the corners X = nx of voxel nx-1 (the answer is R; no voxel nx).  Only one parity of the block is
// left, a 2x2 block in (y, z) with members e = ey | ez<<1: T_e = b_e & ~(b_e^1 & b_e^2) -- the missing x neighbour is the
// clamped voxel itself -- and corner row d = dy | dz<<1 is created by the row's own voxel, member d, when T_d & ~OR_{e>d} T_e.
// lo: the first word of the row in the staged rows (AT_START) or its last one; returns how many of the four are set.
template <bool AT_START>
__device__ __forceinline__ u32 border_nibs(const DenseAt &w, const u64 *__restrict__ bits, const Grid &g, int zp) {
  // bit 0 of the first word's low dword / the bit of the row's last voxel in its last word
  const u32 *t32 = reinterpret_cast<const u32 *>(w.tile) + (AT_START ? 0 : g.lastpos >> 5);
  u32 exy[3] = {~0u, ~0u, ~0u}, exz[3] = {~0u, ~0u, ~0u};
  int yo[3] = {-w.W, 0, w.W}, zo[3] = {w.below, 0, w.above};
  if (w.y == 0) { exy[0] = 0u; yo[0] = 0; }
  if (w.y == g.ny - 1) { exy[2] = 0u; yo[2] = 0; }
  if (w.z == 0) { exz[0] = 0u; zo[0] = 0; }
  if (w.z == g.nzb - 1) { exz[2] = 0u; zo[2] = 0; }
  u32 b[3][3];
#pragma unroll
  for (int dz = 0; dz < 3; dz++)
#pragma unroll
    for (int dy = 0; dy < 3; dy++) b[dz][dy] = t32[2 * (w.idx + yo[dy] + zo[dz])];
  u32 created[4];
  if (__ballot(w.y == 0 || w.y == g.ny - 1 || w.z == 0 || w.z == g.nzb - 1) == 0ull) {
    // away from the y / z borders the 2D chains collapse like the blocks' (member d of corner row d is the row's own voxel):
    //   d = 0: b0 & ~(b1|b2|b3)    d = 1: b1 & ~(b2|b3)    d = 2: b2 & (~b3 | b1 & ~b0)    d = 3: b3 & ~(b2 & b1)
    // with b_e = b[dz - ez + 1][dy - ey + 1]
    created[0] = b[1][1] & ~(b[1][0] | b[0][1] | b[0][0]);
    created[1] = b[1][1] & ~(b[0][2] | b[0][1]);
    created[2] = b[1][1] & (~b[1][0] | (b[2][0] & ~b[2][1]));
    created[3] = b[1][1] & ~(b[1][2] & b[2][1]);
  } else
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const int dy = d & 1, dz = d >> 1;
    u32 earlier = 0u, own = 0u;
#pragma unroll
    for (int e = 3; e >= d; e--) {
      const int sy = dy - (e & 1), sz = dz - (e >> 1);              // the member's row; its neighbours: the other y, the other z
      const u32 t = b[sz + 1][sy + 1] & ~(b[sz + 1][dy - ((e ^ 1) & 1) + 1] & b[dz - ((e ^ 2) >> 1) + 1][sy + 1]) & exy[sy + 1] & exz[sz + 1];
      if (e > d) earlier |= t;
      else own = t;
    }
    created[d] = own & ~earlier;
  }
  if (zp >= 0) {
    // quirk Q1 (see lattice_word): the aliased source slice's voxels at x = 0 / x = nx-1 of rows y-1, y, y+1
    const u32 *s32 = reinterpret_cast<const u32 *>(bits + ((size_t)zp * g.ny + w.y) * g.W + w.k) + (AT_START ? 0 : g.lastpos >> 5);
    const u32 a0 = s32[2 * yo[0]] & exy[0], a1 = s32[0], a2 = s32[2 * yo[2]] & exy[2];
    created[0] &= ~(a0 | a1);
    created[1] &= ~(a1 | a2);
  }
  const u32 any4 = AT_START ? 0u : (u32)(g.lastpos & 31);
  return ((created[0] >> any4) & 1u) + ((created[1] >> any4) & 1u) + ((created[2] >> any4) & 1u) + ((created[3] >> any4) & 1u);
}

// ---------------------------------------------------------------------------------------------
// K2: count + scan.  A block owns COUNT_WB consecutive words of the flat raster order (32 scan segments).
//   phase 1  every word: the six face masks (7 bit-rows) -> quad count; words with a face are
//            queued in LDS (a voxel only creates corners on faces it emits, so words without a
//            face create nothing)
//   phase 2  one lane per QUEUED word: the 8 created-corner masks (27 bit-rows) -> vertex count.  The surface
//            touches a fraction of the words, so the expensive part runs on densely packed lanes.
//   phase 3  one wave per segment: exclusive scan of the packed counts -> prefix; then the block's 32 segment
//            totals are scanned -> segPre, and the block total goes to blockTot.
// k_block_scan (one workgroup) then turns the block totals into blockBase and the grand totals.  (Doing that in
// the last count block to arrive -- totals published with agent-scope atomics, a ticket counter -- works and was
// measured: every block then sits on its CU for the ~5 us its returning atomics take, 0.05 ms in all at 1024^3
// against 0.008 ms for this second launch: profiles/microbench/r2_count_sweep2.log.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 wave_inclusive_sum2(u64 v) {     // two independent 32-bit sums packed lo | hi<<32
  return (u64)wave_inclusive_sum((u32)v) | ((u64)wave_inclusive_sum((u32)(v >> 32)) << 32);
}

// The block scan as the epilogue of the LAST count block to publish its totals (k_block_scan's work for launches of at most
// FOLD_MAX_BLOCKS blocks -- every volume the reference ships, where an extraction's time is its number of launches: blob0 0.117
// -> 0.109 ms, nucleon 0.174 -> 0.168, silicium 0.192 -> 0.188 wall.  Measured beyond that too, round 5: at 1024 blocks, all
// resident at once (512^3), the ticket's return and the tail in ONE workgroup cost 4 us MORE than the second launch -- pass
// 0.1214 against 0.1171 ms, profiles/microbench/r5_sphere512_{fold,nofold}.json; the LDS-tiled form of that size with the tail's
// loads all issued at once: 0.1249 against 0.1150 -- and with several rounds of blocks a lingering block per round costs more
// still (round 2): k_block_scan stays there).
// Hand-off (MI355X guide, inter-workgroup visibility): every block's wave 0 stores its total (and g0pre) write-through
// (agent-scope atomic stores = sc1), waits for them (vmcnt(0)), then adds to the ticket; the block whose add returns
// nblk - 1 acquires at agent scope and reads every total with agent-scope loads.  The err and nVertexWords words are only ever
// touched by device-scope atomics.
constexpr u32 FOLD_MAX_BLOCKS = 64;
template <int NT>
__device__ void block_scan_tail(const u64 *blockTot, u64 *__restrict__ blockBase, u32 nblk, size_t g0, Totals *tot, const Gate &gate,
                                const u32 *__restrict__ sliceOcc, const Grid &g, u64 *waveSum /* NT / 64 LDS words */,
                                int *occ3 /* 3 LDS ints */) {
  constexpr int NWAVES = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cz0 = g.cz0, oz0 = g.oz0, oz1 = g.oz1;
  if (part_of_a_volume(g)) {
    // a SLAB's row: first occupied slice of the counted range, highest and second-highest occupied owned slice (row_flags)
    auto wave_min = [](int v) { for (int sft = 32; sft > 0; sft >>= 1) { const int o = __shfl_xor(v, sft, 64); v = o < v ? o : v; } return v; };
    auto wave_max = [](int v) { for (int sft = 32; sft > 0; sft >>= 1) { const int o = __shfl_xor(v, sft, 64); v = o > v ? o : v; } return v; };
    if (tid == 0) { occ3[0] = 0x7fffffff; occ3[1] = -1; occ3[2] = -1; }
    __syncthreads();
    int lo = 0x7fffffff, hi = -1;
    for (int z = cz0 + tid; z < oz1; z += NT)
      if (sliceOcc[z]) {
        lo = z < lo ? z : lo;
        if (z >= oz0) hi = z;
      }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane == 0 && lo != 0x7fffffff) atomicMin(&occ3[0], lo);
    if (lane == 0 && hi >= 0) atomicMax(&occ3[1], hi);
    __syncthreads();
    const int top = occ3[1];
    int second = -1;
    for (int z = oz0 + tid; z < top; z += NT)
      if (sliceOcc[z]) second = z;
    second = wave_max(second);
    if (lane == 0 && second >= 0) atomicMax(&occ3[2], second);
    __syncthreads();
    if (tid == 0) {
      tot->aliasZ = occ3[0] == 0x7fffffff ? -1 : (int)(g.zglob0 + occ3[0]);
      tot->topZ = top < 0 ? -1 : (int)(g.zglob0 + top);
      tot->top2Z = occ3[2] < 0 ? -1 : (int)(g.zglob0 + occ3[2]);
    }
  } else if (tid == 0) {
    tot->aliasZ = tot->topZ = tot->top2Z = -1;
  }
  u64 runV = 0, runQ = 0;
  for (u32 base = 0; base < nblk; base += NT) {
    const u32 b = base + tid;
    const u64 v = b < nblk ? __hip_atomic_load(&blockTot[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    const u64 incl = wave_inclusive_sum2(v);
    if (lane == 63) waveSum[wv] = incl;
    __syncthreads();
    u64 before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NWAVES; w++) { const u64 t = waveSum[w]; all += t; if (w < wv) before += t; }
    const u64 excl = before + incl - v;                            // halves stay below 2^32: no carry crosses
    const u64 bV = runV + (excl & 0xffffffffull), bQ = runQ + (excl >> 32);
    if (b < nblk) {
      blockBase[2 * (size_t)b] = bV;
      blockBase[2 * (size_t)b + 1] = bQ;
      if (g0 > 0 && (g0 >> COUNT_LG) == b) {
        const u64 in = __hip_atomic_load(&tot->g0pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot->V0 = bV + (in & 0xffffffffull);
        tot->Q0 = bQ + (in >> 32);
      }
    }
    runV += all & 0xffffffffull;
    runQ += all >> 32;
    __syncthreads();
  }
  if (tid == 0) {
    tot->totV = runV;
    tot->totQ = runQ;
    if (gate.on) {
      u32 err = __hip_atomic_load(&tot->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u32 nvw = __hip_atomic_load(&tot->nVertexWords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (runV > gate.coverV || runQ > gate.coverQ || nvw > gate.coverVW) err |= (u32)ERRF_CAPACITY;
      if (err & (u32)ERRF_CAPACITY) atomicOr(&tot->err, (u32)ERRF_CAPACITY);
      tot->go = (err & (u32)(ERRF_ALIAS_UNKNOWN | ERRF_CAPACITY)) == 0 ? 1u : 0u;      // (see k_block_scan)
    }
  }
}

// TILED: every bit row the block reads -- the rows of its own 2048 words, one row before and after, the same of the
// slices below and above -- is first copied into LDS as three contiguous, coalesced ranges (rows are consecutive in the
// flat order; ~54-58 KB), and the face tests and the corner logic read the copy: the 27 rows of a surface word are
// then 27 LDS reads (2 cycles per wave-instruction) instead of 27 trips through the vector cache (8 cycles each, and
// a queue of surface words is unordered, so nothing coalesces), which is what bounds this kernel once most words
// carry faces (2048^3 uint8 noise: 2.2 ms untiled, the figure in DESIGN.md tiled).  Two workgroups of 512 fit a CU.
// Rows wider than TILE_WMAX words (nx > 6144) take the untiled form.
constexpr int TILE_WMAX = 96;
constexpr int TILE_PLANE = COUNT_WB + 4 * TILE_WMAX;      // words per staged slice: the block's rows (two of them partly) + 2
constexpr int COUNT_ZRUN = 8;                             // blocks a workgroup of the column form takes, one above the other

// (untiled, 5 waves per SIMD: 96 VGPRs and a 48-byte spill beat 103 VGPRs at 4 waves, 0.135 vs 0.144 ms; 6 waves spill
//  too much)
template <int MODE, bool TILED, int NT, bool FOLD = false>   // MODE 0 in the library; 2: no block scan, 4: no corner logic (microbench)
__global__ __launch_bounds__(NT, (TILED || NT > 512 ? 4 : 5)) void k_count(const u64 *__restrict__ bits, const u32 *__restrict__ occ, Grid g,
                                               size_t nwords, int q1, u32 *__restrict__ prefix, u64 *__restrict__ segPre,
                                               u64 *blockTot, u32 *__restrict__ vqueue,
                                               Totals *tot, int zrun, u64 *__restrict__ blockBase, Gate gate, int foldArg) {
  const int fold = FOLD ? foldArg : 0;          // FOLD: the block scan as the last block's epilogue (fold = blocks of the launch)
  __shared__ u32 cnt[COUNT_WB];                 // V | Q<<16 per word (<= 512 and <= 384: the packed scan cannot carry)
  __shared__ u64 tailSum[FOLD ? NT / 64 : 1];   // fold: the last block's scan of the block totals
  __shared__ int tailOcc[3];
  __shared__ int lastBlock;
  __shared__ unsigned short queue[COUNT_WB];
  __shared__ u64 segTot[COUNT_WB / 64];
  __shared__ u64 segVW[COUNT_WB / 64];          // per segment: which of its 64 words create vertices
  __shared__ u32 segVWPre[COUNT_WB / 64];       // ... and how many such words the block's earlier segments hold
  __shared__ int nQueued;
  __shared__ u32 vbase, g0InSeg;
  __shared__ u64 tile[TILED ? 3 * TILE_PLANE : 1];
  constexpr int NWAVES = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t g0 = (size_t)(g.oz0 - g.cz0) * g.ny * g.W;   // first owned word (0: no ghost slice)
  // zrun > 0 (TILED, slices that are whole count blocks): the workgroup takes the SAME rows of zrun consecutive slices, one
  // count block after the other, and keeps the staged planes: per block one new plane is copied instead of three
  if (!TILED) zrun = 0;                          // (the untiled forms keep their one block: the loop folds away)
  const u32 bps = zrun ? (u32)(((size_t)g.ny * g.W) >> COUNT_LG) : 1u;        // count blocks per slice
  const u32 colBlock = zrun ? blockIdx.x % bps : 0u, colRun = zrun ? blockIdx.x / bps : 0u;
  for (int it = 0; it < (zrun ? zrun : 1); it++) {
    const u32 blk = zrun ? (colRun * (u32)zrun + (u32)it) * bps + colBlock : blockIdx.x;
    const size_t w0 = (size_t)blk * COUNT_WB;
    if (w0 >= nwords) break;                       // (the same for every thread)
    // (MODE & 16, microbench only: thread 0's clock at the phase boundaries of every block, 16 words per block in blockBase)
    auto stamp = [&](int i) {
      if constexpr ((MODE & 16) != 0) { if (tid == 0) blockBase[(size_t)blk * 16 + i] = (u64)clock64(); }
    };
    stamp(0);
    if (tid == 0) { nQueued = 0; g0InSeg = 0; lastBlock = 0; }
    long long rowFirst = 0;                        // TILED: buffer row (z * ny + y) of the tile's second row
    int slot = 1;                                  // ... and which third of the tile holds the block's own slice
    if (TILED) {
      int k0, y0, z0, k1, y1, z1;
      word_coords(g, w0, k0, y0, z0);
      const size_t wl = w0 + COUNT_WB - 1 < nwords ? w0 + COUNT_WB - 1 : nwords - 1;
      word_coords(g, wl, k1, y1, z1);
      rowFirst = (long long)z0 * g.ny + y0;
      const long long rowLast = (long long)z1 * g.ny + y1;
      const int len = (int)(rowLast - rowFirst + 3) * g.W;    // the block's rows, one before, one after
      const long long nbuf = (long long)g.nzb * g.ny * g.W;
      // (every load of a thread in flight at once -- addresses clamped into the buffer, nothing conditional -- then the LDS
      //  writes: as a loop of load, wait, write the copy was 13 dependent round trips to memory per block, a third of the
      //  kernel's time on a dense field)
      constexpr int TL = (TILE_PLANE + NT - 1) / NT;
      if (it == 0) {
        u64 v[3][TL];
#pragma unroll
        for (int p = 0; p < 3; p++) {
          const long long gstart = (rowFirst - 1 + (long long)(p - 1) * g.ny) * g.W;
#pragma unroll
          for (int u = 0; u < TL; u++) {
            long long gidx = gstart + tid + u * NT;
            gidx = gidx < 0 ? 0 : (gidx >= nbuf ? nbuf - 1 : gidx);      // (rows off the buffer are never read)
            v[p][u] = bits[gidx];
          }
        }
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
          for (int u = 0; u < TL; u++)
            if (tid + u * NT < len) tile[p * TILE_PLANE + tid + u * NT] = v[p][u];
      } else {
        // the planes roll: the slice above the previous block is this block's own, the new slice above takes the place of
        // the one that was below (every thread is past its last read of it: the barrier behind phase 2).  (Loading it a
        // phase early into registers, behind that barrier, was measured: 13 spilled registers, 2.09 vs 1.86 ms.)
        slot = (it + 1) % 3;
        const int fresh = (it + 2) % 3;
        const long long gstart = (rowFirst - 1 + (long long)g.ny) * g.W;
        u64 v[TL];
#pragma unroll
        for (int u = 0; u < TL; u++) {
          long long gidx = gstart + tid + u * NT;
          gidx = gidx < 0 ? 0 : (gidx >= nbuf ? nbuf - 1 : gidx);
          v[u] = bits[gidx];
        }
#pragma unroll
        for (int u = 0; u < TL; u++)
          if (tid + u * NT < len) tile[fresh * TILE_PLANE + tid + u * NT] = v[u];
      }
    }
    __syncthreads();
    stamp(1);
    const long long planeBelow = (long long)((slot + 2) % 3 - slot) * TILE_PLANE, planeAbove = (long long)((slot + 1) % 3 - slot) * TILE_PLANE;
    // where word (k, y, z) of this block is read from
    auto at = [&](int k, int y, int z) -> WordPos {
      if (TILED) {
        const long long t = (long long)z * g.ny + y - rowFirst + 1;
        return word_pos_tile(&tile[slot * TILE_PLANE + t * g.W + k], planeBelow, planeAbove, g, y, z, k);
      }
      return word_pos(bits, g, y, z, k);
    };
    // (untiled: the thread's own words in ONE batch of loads instead of one round trip per word ahead of each word's neighbour
    //  loads; what is kept is one bit per word -- 1024^3 Marschner-Lobb 0.1455 -> 0.143 ms, same box)
    u32 nonEmpty = 0;
    if (!TILED) {
      const u64 *own = bits + (size_t)g.cz0 * g.ny * g.W + w0;
      u64 v[COUNT_WB / NT];
#pragma unroll
      for (int u = 0; u < COUNT_WB / NT; u++) v[u] = w0 + tid + u * NT < nwords ? own[tid + u * NT] : 0ull;
#pragma unroll
      for (int u = 0; u < COUNT_WB / NT; u++) nonEmpty |= (v[u] != 0ull ? 1u : 0u) << u;
    }
    for (int i = tid, u = 0; i < COUNT_WB; i += NT, u++) {
      const size_t gi = w0 + i;
      u32 packed = 0;
      if (TILED ? gi < nwords : ((nonEmpty >> u) & 1u) != 0u) {
        int k, y, z;
        word_coords(g, gi, k, y, z);
        const WordPos wp = at(k, y, z);
        // a word without inside voxels emits nothing: skip its six neighbour loads (outside regions are
        // whole runs of such words, so whole waves take the short way)
        if (!TILED || wp.q[0] != 0) {
          u64 F[6];
          faces_at(wp, g, F);
          int nQ = 0;
#pragma unroll
          for (int f = 0; f < 6; f++) nQ += popc64(F[f]);
          packed = (u32)nQ << 16;
          // (one LDS atomic per word; one per wave -- ballot, popcount, broadcast -- was measured: no faster, 7 more
          //  registers spilled)
          if (nQ) queue[atomicAdd(&nQueued, 1)] = (unsigned short)i;
        }
      }
      cnt[i] = packed;
    }
    stamp(2);
    __syncthreads();
    stamp(3);
    const int nq = (MODE & 4) ? 0 : nQueued;
    if constexpr ((MODE & 16) != 0) { if (tid == 0) blockBase[(size_t)blk * 16 + 9] = (u64)nq; }
    u32 errBits = 0;
    for (int j = tid; j < nq; j += NT) {
      const int i = queue[j];
      const size_t gi = w0 + i;
      int k, y, z;
      word_coords(g, gi, k, y, z);
      WordInfo w;
      u32 unknown;
      classify_word_at<false>(at(k, y, z), bits, occ, g, q1, w, unknown);
      errBits |= unknown;
      int nV = 0;
#pragma unroll
      for (int c = 0; c < 8; c++) nV += popc64(w.C[c]);
      cnt[i] |= (u32)nV;
    }
    if (errBits) {
      atomicOr(&tot->err, errBits);
      if (fold) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (on its way before this block's ticket, behind the barrier)
    }
    stamp(4);
    __syncthreads();
    stamp(5);
    for (int sg = wv; sg < COUNT_WB / 64; sg += NWAVES) {
      const size_t gi = w0 + sg * 64 + lane;
      const u32 packed = cnt[sg * 64 + lane];     // 0 past the end
      const u32 incl = wave_inclusive_sum(packed);
      if (gi < nwords) prefix[gi] = incl - packed;
      if (gi == g0) g0InSeg = incl - packed;
      if (lane == 63) segTot[sg] = (u64)(incl & 0xffffu) | ((u64)(incl >> 16) << 32);
      // words that create vertices go to the global vertex-word queue IN ORDER (the point pass runs one lane per such
      // word; a wave of it then owns one contiguous run of vertex ids): their places follow from these ballots
      const u64 vm = __ballot((packed & 0xffffu) != 0u);
      if (lane == 0) segVW[sg] = vm;
    }
    stamp(6);
    __syncthreads();
    stamp(7);
    if (vqueue) {
      if (wv == 1) {
        // (wave 1, beside wave 0's segment scan below: 32 segment counts -> exclusive prefix, one global atomic per block)
        const u32 n = lane < COUNT_WB / 64 ? (u32)__popcll(segVW[lane]) : 0u;
        const u32 incl = wave_inclusive_sum(n);
        if (lane < COUNT_WB / 64) segVWPre[lane] = incl - n;
        const u32 total = __shfl(incl, 63, 64);
        if (lane == 0 && total) vbase = atomicAdd(&tot->nVertexWords, total);
      }
    }
    if (MODE & 2) continue;
    if (wv == 0) {
      // the block's 32 segments: exclusive scan of their totals -> segPre; block total -> blockTot
      const u64 t = lane < COUNT_WB / 64 ? segTot[lane] : 0ull;
      const u64 incl = wave_inclusive_sum2(t);
      const u64 excl = incl - t;                                   // both halves stay below 2^21: no borrow crosses
      const size_t seg = (w0 >> 6) + lane;
      if (lane < COUNT_WB / 64 && (seg << 6) < nwords) segPre[seg] = excl;
      if (g0 > 0 && (g0 >> COUNT_LG) == blk && lane == (int)((g0 >> 6) & (COUNT_WB / 64 - 1))) {
        const u32 in = g0InSeg;
        const u64 pre = excl + ((u64)(in & 0xffffu) | ((u64)(in >> 16) << 32));
        if (fold) __hip_atomic_store(&tot->g0pre, pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else tot->g0pre = pre;
      }
      if (lane == COUNT_WB / 64 - 1) {
        if (fold) __hip_atomic_store(&blockTot[blk], incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // write-through
        else blockTot[blk] = incl;
      }
    }
    if (vqueue) {
      __syncthreads();
      for (int sg = wv; sg < COUNT_WB / 64; sg += NWAVES) {
        const u64 vm = segVW[sg];
        if ((vm >> lane) & 1ull)
          vqueue[vbase + segVWPre[sg] + (u32)__popcll(vm & lowmask(lane))] = (u32)(w0 + sg * 64 + lane);
      }
    }
    stamp(8);
    if (fold) {
      // the ticket, behind everything else of the block (the other waves are gone by the time it returns): wave 0's stores
      // have landed (vmcnt(0)), then one agent-scope add; the block that draws the last ticket scans the totals
      if (wv == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32 t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&tot->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __shfl(t, 0, 64);
        if (lane == 0 && t == (u32)fold - 1u) lastBlock = 1;
      }
      __syncthreads();
      // (lastBlock is reset at the top of a workgroup's NEXT block, unguarded: the workgroup that reads 1 here has no next
      //  block -- the last ticket says every block of the launch is done -- and elsewhere 0 is written over 0)
      if (lastBlock) {                               // (the same for every thread)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        block_scan_tail<NT>(blockTot, blockBase, (u32)fold, g0, tot, gate, occ, g, tailSum, tailOcc);
      }
    }
  }   // the workgroup's next block
}

// ---------------------------------------------------------------------------------------------
// K2, the DENSE form (a surface that touches most words; rows of a power of two of words -- whole or not -- so that a row never
// straddles two blocks).  Same outputs as k_count, another shape:
//   * ONE phase instead of two: every word of the block, in raster order, takes its faces and its corner logic from the
//     same nine staged rows, the corner logic per lattice corner (lattice_word: the closed forms, no x+1 rows, no queue of
//     surface words, no data-dependent loads); the vertices a word's voxel 63 creates at the corners one past the word are
//     the next word's four `nib` bits, fetched through LDS; the x borders of the rows are a short pass of their own;
//   * the counts stay in registers between that phase and the scan (thread t owns words t, t+512, ... = lane `lane` of
//     segments wv, wv+8, ...);
//   * a workgroup walks up its column of blocks with the memory round trips of a block taken off its path: the plane
//     the NEXT block adds is loaded into registers before this block's words are looked at and written to LDS behind them
//     (a block's time was a third copy, and the copy pure latency); the returning atomic that reserves the block's
//     stretch of the vertex-word queue is waited for one block later, and the queue is written then;
//   * two barriers per block instead of five.
// zrun <= 1: one block per workgroup (slices that are not whole blocks, thin slabs): the same code without the roll.
// ---------------------------------------------------------------------------------------------
template <int MODE>    // MODE 0 in the library; microbench: 4 no corner logic, 8 no x-border pass
__global__ __launch_bounds__(512, 4) void k_count_dense(const u64 *__restrict__ bits, const u32 *__restrict__ occ, Grid g, size_t nwords,
                                                       int q1, u32 *__restrict__ prefix, u64 *__restrict__ segPre,
                                                       u64 *__restrict__ blockTot, u32 *__restrict__ vqueue, Totals *__restrict__ tot,
                                                       int zrun) {
  constexpr int NT = 512, NWAVES = NT / 64, NSEG = COUNT_WB / 64, PER = COUNT_WB / NT, TL = (TILE_PLANE + NT - 1) / NT;
  __shared__ u64 tileStore[3 * TILE_PLANE + 1];
  u64 *const tile = tileStore + 1;       // (the word before the first word of the staged rows is read -- and not looked at)
  __shared__ unsigned char seam[COUNT_WB], seamV[COUNT_WB], seamF[COUNT_WB];   // lattice_word's `nibs` of every word; border_nibs behind a row / at its start
  __shared__ u64 segTot[NSEG];
  __shared__ u64 segVW[2][NSEG];         // per segment: which of its 64 words create vertices (this block's, the one before's)
  __shared__ u32 segVWPre[2][NSEG];      // ... and how many such words the block's earlier segments hold
  __shared__ u32 vbaseS[2];
  __shared__ u32 g0InSeg;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t g0 = (size_t)(g.oz0 - g.cz0) * g.ny * g.W;   // first owned word (0: no ghost slice)
  if (zrun < 1) zrun = 1;
  const u32 bps = zrun > 1 ? (u32)(((size_t)g.ny * g.W) >> COUNT_LG) : 1u;        // count blocks per slice
  const u32 colBlock = zrun > 1 ? blockIdx.x % bps : 0u, colRun = zrun > 1 ? blockIdx.x / bps : 0u;
  auto block_of = [&](int it) -> u32 { return zrun > 1 ? (colRun * (u32)zrun + (u32)it) * bps + colBlock : blockIdx.x; };
  int nsteps = 0;
  while (nsteps < zrun && (size_t)block_of(nsteps) * COUNT_WB < nwords) nsteps++;
  if (tid == 0) g0InSeg = 0;
  const long long nbuf = (long long)g.nzb * g.ny * g.W;
  // geometry of a block's staged rows: its own rows, one before, one after (the same of the slices below and above)
  auto rows_of = [&](size_t w0, long long &rowFirst, int &len, int &zFirst, int &zLast) {
    int k0, y0, z0, k1, y1, z1;
    word_coords(g, w0, k0, y0, z0);
    const size_t wl = w0 + COUNT_WB - 1 < nwords ? w0 + COUNT_WB - 1 : nwords - 1;
    word_coords(g, wl, k1, y1, z1);
    rowFirst = (long long)z0 * g.ny + y0;
    len = (int)((long long)z1 * g.ny + y1 - rowFirst + 3) * g.W;
    zFirst = z0; zLast = z1;
  };
  auto load_plane = [&](long long gstart, u64 (&v)[TL]) {
#pragma unroll
    for (int u = 0; u < TL; u++) {
      long long gidx = gstart + tid + u * NT;
      gidx = gidx < 0 ? 0 : (gidx >= nbuf ? nbuf - 1 : gidx);      // (rows off the buffer are never read)
      v[u] = bits[gidx];
    }
  };
  auto store_plane = [&](int p, int len, const u64 (&v)[TL]) {
#pragma unroll
    for (int u = 0; u < TL; u++)
      if (tid + u * NT < len) tile[p * TILE_PLANE + tid + u * NT] = v[u];
  };
  if (nsteps == 0) return;
  {
    long long rowFirst; int len, za, zb;
    rows_of((size_t)block_of(0) * COUNT_WB, rowFirst, len, za, zb);
    u64 v[3][TL];
#pragma unroll
    for (int p = 0; p < 3; p++) load_plane((rowFirst - 1 + (long long)(p - 1) * g.ny) * g.W, v[p]);
#pragma unroll
    for (int p = 0; p < 3; p++) store_plane(p, len, v[p]);
  }
  __syncthreads();
  u32 errBits = 0, pendingBase = 0;
  // the queue stretch of block `it`: reserved by wave 1 behind that block's scan, written a block later
  auto flush_queue = [&](int it) {
    const int b = it & 1;
    const size_t w0 = (size_t)block_of(it) * COUNT_WB;
    const u32 base = vbaseS[b];
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const int sg = wv + NWAVES * j;
      const u64 vm = segVW[b][sg];
      if ((vm >> lane) & 1ull) vqueue[base + segVWPre[b][sg] + (u32)__popcll(vm & lowmask(lane))] = (u32)(w0 + sg * 64 + lane);
    }
  };
  for (int it = 0; it < nsteps; it++) {
    const u32 blk = block_of(it);
    const size_t w0 = (size_t)blk * COUNT_WB;
    const int cur = it & 1;
    long long rowFirst; int len, zFirst, zLast;
    rows_of(w0, rowFirst, len, zFirst, zLast);
    const int slot = (it + 1) % 3;                       // which third of the tile holds the block's own slice
    const bool more = it + 1 < nsteps;
    u64 pv[TL];
    if (more) load_plane((rowFirst - 1 + 2ll * g.ny) * g.W, pv);     // the slice above the NEXT block (same rows, one slice up)
    // word i of the block sits at tile[own + W + i]: the staged rows are the block's rows behind one row more
    DenseAt da;
    da.tile = tile; da.W = g.W;
    da.below = ((slot + 2) % 3 - slot) * TILE_PLANE; da.above = ((slot + 1) % 3 - slot) * TILE_PLANE;
    const int own = slot * TILE_PLANE + g.W;
    // quirk Q1's source slice: one look-up per block where the block lies in one slice (the usual case)
    u32 unknownBlk = 0;
    const int zpBlk = zFirst == zLast ? alias_of(occ, g, q1, zFirst, unknownBlk) : -1;
    // the two x borders of every row of the block: the corners X = nx of its last word's voxel 63 (first half of the
    // items), the corners X = 0 of its first word's voxel 0 (second half)
    // (ahead of the words, a share for every wave: behind them two waves would walk these few dependent LDS reads alone
    //  while six wait at the barrier)
    const int lgRows = COUNT_LG - g.wShift, nrows = (MODE & 8) ? 0 : 1 << lgRows, share = (2 * nrows + NWAVES - 1) / NWAVES;
    for (int u = lane; u < share; u += 64) {
      const int t = wv * share + u;
      if (t >= 2 * nrows) break;
      const int v = t & (nrows - 1);
      const bool atStart = (t >> lgRows) != 0;
      const size_t gi = w0 + ((size_t)v << g.wShift);
      u32 nibs = 0;
      if (gi < nwords) {
        word_coords(g, gi, da.k, da.y, da.z);
        u32 unknown;
        const int zp = zFirst == zLast ? zpBlk : alias_of(occ, g, q1, da.z, unknown);
        da.idx = own + (v << g.wShift);
        if (atStart) {
          if (tile[da.idx] & 1ull) nibs = border_nibs<true>(da, bits, g, zp);
        } else {
          da.idx += g.W - 1; da.k = g.W - 1;
          if ((tile[da.idx] >> g.lastpos) & 1ull) nibs = border_nibs<false>(da, bits, g, zp);
        }
      }
      if (atStart) seamF[v] = (unsigned char)nibs;
      else seamV[v] = (unsigned char)nibs;
    }
    u32 pk[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const int i = tid + NT * j;
      const size_t gi = w0 + i;
      u32 packed = 0, nibs = 0;
      if (gi < nwords) {
        word_coords(g, gi, da.k, da.y, da.z);
        da.idx = own + i;
        // nothing inside the word and nothing in the voxel before it: no face, no vertex, nothing the word before lacks
        const bool live = tile[da.idx] != 0ull || (da.k > 0 && (tile[da.idx - 1] >> 63));
        u32 unknown = unknownBlk;
        const int zp = zFirst == zLast ? zpBlk : (live ? alias_of(occ, g, q1, da.z, unknown) : -1);
        if (live) errBits |= unknown;
        const bool yzBorder = da.y == 0 || da.y == g.ny - 1 || da.z == 0 || da.z == g.nzb - 1;
        if (__ballot(yzBorder && live) != 0ull) { if (live) lattice_word<true, (MODE & 4) != 0>(da, bits, g, zp, packed, nibs); }
        else if (live) lattice_word<false, (MODE & 4) != 0>(da, bits, g, zp, packed, nibs);
      }
      pk[j] = packed;
      seam[i] = (unsigned char)nibs;
    }
    if (vqueue && it > 0 && tid == 64) vbaseS[cur ^ 1] = pendingBase;       // (the atomic of the block before has long returned)
    __syncthreads();                                                        // ---- A: every read of the staged rows is done
    if (more) store_plane(it % 3, len, pv);                                 // the plane that was below this block: no longer needed
    if (vqueue && it > 0) flush_queue(it - 1);
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const int sg = wv + NWAVES * j, i = sg * 64 + lane, k = i & (g.W - 1);
      const size_t gi = w0 + i;
      // the vertices voxel 63 creates at the corners one past the word -- the next word's four bits, or those of the place
      // behind the row -- and in a row's first word those of voxel 0 at X = 0
      u32 packed = pk[j] + (k == g.W - 1 ? (u32)seamV[i >> g.wShift] : (u32)seam[(i + 1) & (COUNT_WB - 1)]);
      if (k == 0) packed += (u32)seamF[i >> g.wShift];
      const u32 incl = wave_inclusive_sum(packed);
      if (gi < nwords) prefix[gi] = incl - packed;
      if (gi == g0) g0InSeg = incl - packed;
      if (lane == 63) segTot[sg] = (u64)(incl & 0xffffu) | ((u64)(incl >> 16) << 32);
      // words that create vertices go to the global vertex-word queue IN ORDER (see k_count)
      const u64 vm = __ballot((packed & 0xffffu) != 0u);
      if (lane == 0) segVW[cur][sg] = vm;
    }
    __syncthreads();                                                        // ---- B: segment totals, the next block's plane
    if (wv == 0) {
      // the block's 32 segments: exclusive scan of their totals -> segPre; block total -> blockTot
      const u64 t = lane < NSEG ? segTot[lane] : 0ull;
      const u64 incl = wave_inclusive_sum2(t);
      const u64 excl = incl - t;                                   // both halves stay below 2^21: no borrow crosses
      const size_t seg = (w0 >> 6) + lane;
      if (lane < NSEG && (seg << 6) < nwords) segPre[seg] = excl;
      if (g0 > 0 && (g0 >> COUNT_LG) == blk && lane == (int)((g0 >> 6) & (NSEG - 1))) {
        const u32 in = g0InSeg;
        tot->g0pre = excl + ((u64)(in & 0xffffu) | ((u64)(in >> 16) << 32));
      }
      if (lane == NSEG - 1) blockTot[blk] = incl;
    } else if (wv == 1 && vqueue) {
      // 32 segment counts -> exclusive prefix, one global atomic per block (its answer is read a block later)
      const u32 n = lane < NSEG ? (u32)__popcll(segVW[cur][lane]) : 0u;
      const u32 incl = wave_inclusive_sum(n);
      if (lane < NSEG) segVWPre[cur][lane] = incl - n;
      const u32 total = __shfl(incl, 63, 64);
      pendingBase = 0;
      if (lane == 0 && total) pendingBase = atomicAdd(&tot->nVertexWords, total);
    }
  }
  if (errBits) atomicOr(&tot->err, errBits);
  if (vqueue) {
    if (tid == 64) vbaseS[(nsteps - 1) & 1] = pendingBase;
    __syncthreads();
    flush_queue(nsteps - 1);
  }
}

// More than 8192 count blocks (volumes beyond 1024^3): the scan below runs as one workgroup per chunk of 8192 blocks, each
// starting from the sums of the chunks before it -- which this launch leaves behind the block totals, blockTot[nblk + 2j]
// (vertices) and [nblk + 2j + 1] (quads) for chunk j.  (One workgroup walking 65 536 totals in eight dependent batches was
// 0.056 ms of a 2048^3 volume's pass.)
constexpr u32 SCAN_CHUNK = 8192;
__global__ __launch_bounds__(1024) void k_block_partial(u64 *__restrict__ blockTot, u32 nblk) {
  __shared__ u64 sumV[16], sumQ[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const u32 first = blockIdx.x * SCAN_CHUNK;
  u64 v = 0, q = 0;
#pragma unroll
  for (int r = 0; r < (int)(SCAN_CHUNK / 1024); r++) {
    const u32 b = first + r * 1024 + tid;
    const u64 t = b < nblk ? blockTot[b] : 0ull;
    v += t & 0xffffffffull;
    q += t >> 32;
  }
  for (int sft = 32; sft > 0; sft >>= 1) { v += __shfl_xor(v, sft, 64); q += __shfl_xor(q, sft, 64); }
  if (lane == 0) { sumV[wv] = v; sumQ[wv] = q; }
  __syncthreads();
  if (tid == 0) {
    u64 a = 0, c = 0;
    for (int w = 0; w < 16; w++) { a += sumV[w]; c += sumQ[w]; }
    blockTot[(size_t)nblk + 2 * blockIdx.x] = a;
    blockTot[(size_t)nblk + 2 * blockIdx.x + 1] = c;
  }
}

// Exclusive scan of the count blocks' totals (V | Q<<32 each) -> blockBase[2b], [2b+1] and the grand totals.  One
// workgroup: rows of 1024 consecutive blocks, coalesced loads, ROWS rows in flight at once (the loads are what
// takes time), then per row a workgroup-wide exclusive scan; V and Q of a row fit 32 bits each (1024 x 2^21),
// the running bases are 64-bit.
// gate (cuberille_step_begin): the launches behind this one were sized from the previous extraction; they only run
// (Totals::go) when the counts fit what they were sized for and no flag of the count stands.
__global__ __launch_bounds__(1024) void k_block_scan(const u64 *__restrict__ blockTot, u64 *__restrict__ blockBase, u32 nblk,
                                                     size_t g0, Totals *__restrict__ tot, Gate gate,
                                                     const u32 *__restrict__ sliceOcc, int cz0, int oz0, int oz1, long long zglob0,
                                                     int slab) {
  constexpr int ROWS = 8;
  __shared__ u64 waveSum[8][16];
  __shared__ int firstOcc, topOcc, top2Occ;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // the three slices of a SLAB's row (row_flags; a whole volume has no neighbours to tell): first occupied slice of the
  // counted range, highest and second-highest occupied owned slice.  Wave reductions, then one LDS atomic per wave.
  // (gridDim.x > 1: this workgroup scans chunk blockIdx.x of SCAN_CHUNK blocks, from the sums k_block_partial left)
  const u32 chunk = blockIdx.x, nchunks = gridDim.x;
  if (chunk > 0) {
    // (the slab's three slices, the totals and the gate belong to the first / last workgroup)
  } else
  if (slab) {
    auto wave_min = [](int v) { for (int sft = 32; sft > 0; sft >>= 1) { const int o = __shfl_xor(v, sft, 64); v = o < v ? o : v; } return v; };
    auto wave_max = [](int v) { for (int sft = 32; sft > 0; sft >>= 1) { const int o = __shfl_xor(v, sft, 64); v = o > v ? o : v; } return v; };
    if (tid == 0) { firstOcc = 0x7fffffff; topOcc = -1; top2Occ = -1; }
    __syncthreads();
    int lo = 0x7fffffff, hi = -1;
    for (int z = cz0 + tid; z < oz1; z += 1024)
      if (sliceOcc[z]) {
        lo = z < lo ? z : lo;
        if (z >= oz0) hi = z;                      // (ascending: the last one is this thread's highest)
      }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane == 0 && lo != 0x7fffffff) atomicMin(&firstOcc, lo);
    if (lane == 0 && hi >= 0) atomicMax(&topOcc, hi);
    __syncthreads();
    const int top = topOcc;
    int second = -1;
    for (int z = oz0 + tid; z < top; z += 1024)
      if (sliceOcc[z]) second = z;
    second = wave_max(second);
    if (lane == 0 && second >= 0) atomicMax(&top2Occ, second);
    __syncthreads();
    if (tid == 0) {
      tot->aliasZ = firstOcc == 0x7fffffff ? -1 : (int)(zglob0 + firstOcc);
      tot->topZ = top < 0 ? -1 : (int)(zglob0 + top);
      tot->top2Z = top2Occ < 0 ? -1 : (int)(zglob0 + top2Occ);
    }
  } else if (tid == 0) {
    tot->aliasZ = tot->topZ = tot->top2Z = -1;
  }
  u64 runV = 0, runQ = 0;
  const u32 nblkAll = nblk;
  u32 base0 = 0;
  if (nchunks > 1) {
    for (u32 j = 0; j < chunk; j++) { runV += blockTot[(size_t)nblkAll + 2 * j]; runQ += blockTot[(size_t)nblkAll + 2 * j + 1]; }
    base0 = chunk * SCAN_CHUNK;
    nblk = base0 + SCAN_CHUNK < nblkAll ? base0 + SCAN_CHUNK : nblkAll;      // (this workgroup's blocks: [base0, nblk))
  }
  u64 v[ROWS], vn[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const u32 b = base0 + r * 1024 + tid;
    vn[r] = b < nblk ? blockTot[b] : 0ull;
  }
  for (u32 base = base0; base < nblk; base += 1024 * ROWS) {
    // (the next batch's totals on their way while this one is scanned)
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      v[r] = vn[r];
      const u64 b = (u64)base + 1024 * ROWS + r * 1024 + tid;
      vn[r] = b < nblk ? blockTot[b] : 0ull;
    }
    // (the rows of a batch between ONE pair of barriers: a pair per row was 64 of them, 0.068 ms, for the 65 536 blocks of a
    //  2048^3 volume)
    u64 incl[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      incl[r] = wave_inclusive_sum2(v[r]);
      if (lane == 63) waveSum[r][wv] = incl[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      if (base + r * 1024 >= nblk) break;                          // workgroup-uniform
      const u32 b = base + r * 1024 + tid;
      u64 before = 0, all = 0;
#pragma unroll
      for (int w = 0; w < 16; w++) { const u64 t = waveSum[r][w]; all += t; if (w < wv) before += t; }
      const u64 excl = before + incl[r] - v[r];                    // halves stay below 2^32: no carry crosses
      const u64 bV = runV + (excl & 0xffffffffull), bQ = runQ + (excl >> 32);
      if (b < nblk) {
        blockBase[2 * (size_t)b] = bV;
        blockBase[2 * (size_t)b + 1] = bQ;
        if (g0 > 0 && (g0 >> COUNT_LG) == b) {
          const u64 in = tot->g0pre;
          tot->V0 = bV + (in & 0xffffffffull);
          tot->Q0 = bQ + (in >> 32);
        }
      }
      runV += all & 0xffffffffull;
      runQ += all >> 32;
    }
    __syncthreads();
  }
  if (tid == 0 && chunk == nchunks - 1) {
    tot->totV = runV;
    tot->totQ = runQ;
    if (gate.on) {
      u32 err = tot->err;
      if (runV > gate.coverV || runQ > gate.coverQ || tot->nVertexWords > gate.coverVW) err |= (u32)ERRF_CAPACITY;
      tot->err = err;
      // (ERRF_ALIAS_BELOW_BUFFER alone does not close the gate: the count assumed that nothing is occupied below this
      //  buffer, which is what the rows of the ranks below usually confirm -- the vertex phase runs on that assumption and
      //  the cell pass, which sees all rows, decides; a recount voids it when the assumption was wrong)
      tot->go = (err & (u32)(ERRF_ALIAS_UNKNOWN | ERRF_CAPACITY)) == 0 ? 1u : 0u;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K3: emit.
// ---------------------------------------------------------------------------------------------
struct EmitArgs {
  const u64 *bits;
  const u32 *occ;      // per-slice occupancy (quirk Q1)
  int q1;
  const u32 *prefix;   // the three levels of the prefix sums: word in segment, segment in block, block
  const u64 *segPre;
  const u64 *blockBase;
  size_t nblk;
  const Totals *tot;
  float *points;
  u64 *cells;          // 4 ids per quad, or 2 x 3 ids per quad when triangulating
  u64 pointOffset;     // global id of this rank's first point
  u32 *cmap;           // dense lattice-corner -> vertex index map, or null (see corner_map_index)
  const u32 *headV, *headQ;   // word producing output 64*i (k_heads_search), or null
  const u64 *extIds;   // Grid::extAlias: global ids of the top-plane corners of the source slice, dense (nx+1) x (ny+1);
                       // their positions stand behind this rank's own points, from index totV on
  const Totals *rows;  // cuberille_step_end: the gathered totals of all ranks (device memory), or null
  int nRanks, rank;    //   -> this rank's point id offset = owned points of the ranks below; any flag on any rank: no cells
  int dyn;             // the launch was sized blindly (cuberille_step_begin): sizes from `tot`, and only when tot->go
};

// absolute exclusive prefix (SHIFT 0: vertices, 16: quads) at the start of the segment that holds word gi
template <int SHIFT>
__device__ __forceinline__ u64 seg_base(const EmitArgs &a, size_t gi) {
  const u64 sp = a.segPre[gi >> 6];
  return a.blockBase[2 * (gi >> COUNT_LG) + (SHIFT ? 1 : 0)] + (SHIFT ? (sp >> 32) : (sp & 0xffffffffull));
}

// The reference finds a corner's id in a std::map keyed by (x,y) per z-plane (h:272-313).  With
// 288 GB of HBM the MI355X equivalent is a dense array over all (nx+1)(ny+1)(nz+1) lattice corners,
// never initialised: the creator of a vertex stores its index there (k_emit_points) and only
// corners of emitted quads -- which are always vertices -- are ever read (k_emit_cells).
// The map is bricked: 4 (x) x 2 (y) x 2 (z) corners share one 64-byte half of a cache line, two such bricks that are
// neighbours in y the whole 128-byte line, bricks in raster order.  The four corners of a quad differ in two of the
// three coordinates; in the row-major layout of round 2 a quad across x touched four lines (its corners differ in y
// and z), here one to four (two 64-byte pieces on average over the three orientations instead of 3.3), and the
// creators' 4-byte stores of neighbouring corners meet in the same lines too: cell pass 0.39 against 0.43 ms, point
// pass 0.219 against 0.229 at 1024^3 Marschner-Lobb, 8.2 / 4.8 against 8.9 / 5.6 ms at 2048^3 noise.
// (Tuning::cmap_linear 1: the row-major form, for the A/B.)
__device__ __forceinline__ size_t corner_map_index(const Grid &g, int cx, int cy, int cz) {
  if (g.cmapLinear) return ((size_t)cz * (g.ny + 1) + cy) * (size_t)(g.nx + 1) + cx;
  const size_t bx = (size_t)(g.nx + 4) >> 2, by = (size_t)(g.ny + 4) >> 2;         // bricks per row / per slice column
  const size_t brick = ((size_t)(cz >> 1) * by + (size_t)(cy >> 2)) * bx + (size_t)(cx >> 2);
  return brick * 32 + (size_t)((((cy >> 1) & 1) << 4) | ((cz & 1) << 3) | ((cy & 1) << 2) | (cx & 3));
}

__device__ __forceinline__ size_t word_index(const Grid &g, int y, int z, int k) {
  return ((size_t)(z - g.cz0) * g.ny + y) * g.W + k;
}
__device__ __forceinline__ int getbit(const u64 *__restrict__ bits, const Grid &g, int x, int y, int z) {
  return (int)((bits[((size_t)z * g.ny + y) * g.W + (x >> 6)] >> (x & 63)) & 1ull);
}

// I3 + txx:268-270: lattice corner -> physical point - spacing/2, in the mesh's float coordinates
__device__ __forceinline__ void corner_point(const Geo &geo, long long cx, long long cy, long long cz, float p[3]) {
  // (cx, cy, cz: positions in the volume; TransformIndexToPhysicalPoint takes the INDEX, position + region start -- both within
  //  2^31, cuberille_api.hip: validate -- added as integers, one conversion each)
  const double idx[3] = {(double)((int)cx + geo.istart[0]), (double)((int)cy + geo.istart[1]), (double)((int)cz + geo.istart[2])};
#pragma unroll
  for (int r = 0; r < 3; r++) {
    double sum = 0.0;
#pragma unroll
    for (int c = 0; c < 3; c++) sum += geo.i2p[r * 3 + c] * idx[c];
    const float v = (float)(sum + geo.origin[r]);
    p[r] = (float)((double)v - (geo.spacing[r] / 2.0));
  }
}

// ... and a vertex on the BOTTOM plane of a slab's ghost slice gets a NaN x: it belongs to the rank below, no cell of
// this rank touches that plane, and the projection skips it (it would need one more halo slice than anything else)
__device__ __forceinline__ void ghost_bottom_mark(const Grid &g, int cz, float p[3]) {
  if (cz == g.cz0 && g.cz0 < g.oz0) p[0] = __builtin_nanf("");
}

// Global id of lattice corner (cx,cy,cz) (cz local) by the closed form: creator = first block
// member in raster order that activates it; id = creator's word base + created corners before the
// creator inside its word + rank of the corner among the creator's created corners.
__device__ u64 corner_id_generic(const EmitArgs &a, const Grid &g, int cx, int cy, int cz) {
  int b[8];
  bool ex[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int x = cx - (e & 1), y = cy - ((e >> 1) & 1), z = cz - (e >> 2);
    ex[e] = (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nzb);
    b[e] = getbit(a.bits, g, clampi(x, 0, g.nx - 1), clampi(y, 0, g.ny - 1), clampi(z, 0, g.nzb - 1));
  }
  int creator = -1;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const bool ac = ex[e] && b[e] && !(b[e ^ 1] && b[e ^ 2] && b[e ^ 4]);
    if (ac) creator = e;     // keep the largest code = earliest in raster order
  }
  if (creator < 0) return ~0ull;    // not a mesh vertex (cannot happen for a corner of an emitted quad)
  const int wx = cx - (creator & 1), wy = cy - ((creator >> 1) & 1), wz = cz - (creator >> 2);
  const int k = wx >> 6, bx = wx & 63;
  WordInfo w;
  u32 unk;
  classify_word<false>(a.bits, a.occ, g, a.q1, wy, wz, k, w, unk);
  const size_t gi = word_index(g, wy, wz, k);
  u64 id = seg_base<0>(a, gi) + (a.prefix[gi] & 0xffffu);
  unsigned cm = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    id += popc64(w.C[i] & lowmask(bx));
    cm |= (unsigned)((w.C[i] >> bx) & 1ull) << i;
  }
  const int j = kEncCorner[creator];
  id += __popc(cm & ((1u << j) - 1u));
  return id;
}

// position of the set bit of rank r (0-based) in an 8-bit mask
__device__ __forceinline__ int select_bit8(unsigned m, int r) {
  for (int i = 0; i < r; i++) m &= m - 1;
  return __ffs((int)m) - 1;
}

// Inverse mapping: output index -> source.  Every output (vertex, quad) has exactly one producing voxel and its
// id is its position in the raster-ordered enumeration.  The surface is ~1 % of the voxels: a lane per voxel (or
// per word) would idle almost every lane, a lane per OUTPUT keeps them all busy and writes each output buffer
// front to back.  The word that produces output idx is found through the three levels of the prefix sums: the
// largest block, then segment, then word whose exclusive prefix is <= idx (entries with equal prefixes are empty,
// so "the largest" is the one that holds the output).
template <int SHIFT>
__device__ __forceinline__ size_t locate_word(const EmitArgs &a, size_t nwords, u64 idx, u32 &within) {
  constexpr int C = SHIFT ? 1 : 0;
  size_t lo = 0, hi = a.nblk;
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) >> 1;
    if (a.blockBase[2 * mid + C] <= idx) lo = mid; else hi = mid;
  }
  u32 r = (u32)(idx - a.blockBase[2 * lo + C]);             // < 2^21
  const size_t nseg = (nwords + 63) >> 6;
  size_t slo = lo << (COUNT_LG - 6), shi = slo + (COUNT_WB / 64);
  if (shi > nseg) shi = nseg;
  while (shi - slo > 1) {
    const size_t mid = (slo + shi) >> 1;
    const u64 sp = a.segPre[mid];
    if ((u32)(SHIFT ? (sp >> 32) : (sp & 0xffffffffull)) <= r) slo = mid; else shi = mid;
  }
  {
    const u64 sp = a.segPre[slo];
    r -= (u32)(SHIFT ? (sp >> 32) : (sp & 0xffffffffull));
  }
  size_t wlo = slo << 6, whi = wlo + 64;
  if (whi > nwords) whi = nwords;
  while (whi - wlo > 1) {                        // largest word with prefix <= r (skips empty words)
    const size_t mid = (wlo + whi) >> 1;
    if (((a.prefix[mid] >> SHIFT) & 0xffffu) <= r) wlo = mid; else whi = mid;
  }
  within = r - ((a.prefix[wlo] >> SHIFT) & 0xffffu);
  return wlo;
}

// head table: entry t = the word that produces output 64 t.  One lane per ENTRY (outputs/64 of them): ~25 dependent
// L2 reads each, but only a few hundred thousand lanes.
template <int SHIFT>
__global__ __launch_bounds__(256) void k_heads_search(EmitArgs a, size_t nwords, u64 nHeads, u32 *__restrict__ head) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (a.dyn) {
    if (!a.tot->go) return;
    nHeads = ((SHIFT ? a.tot->totQ : a.tot->totV) + 63) / 64;
  }
  if (t >= nHeads) return;
  u32 within;
  head[t] = (u32)locate_word<SHIFT>(a, nwords, t * 64, within);
}

// Per-wave form of the search: a wave holds 64 consecutive outputs, which come from a short run of words:
// `head[o/64]` names the word that produces output o & ~63; the 64 lanes load the absolute prefixes of the 256 words
// from there on -- four consecutive words per lane, ONE 16-byte load of the prefix array plus the segment's base (a group of
// four aligned words shares its segment) -- and each lane finds its own word with a 6-step shuffle search over the
// groups and a look at the four words of its group: two dependent memory round trips instead of the ~25 of a per-lane
// search.  (A window of 64 words, one per lane, slid two or three times on a surface like the Marschner-Lobb sheet,
// where 64 quads come from ~19 words spread over ~130: every slide one more dependent round trip.)  Outputs that lie
// beyond the window slide it; after a few slides the lane falls back to locate_word.  The prefix array is readable up
// to three words past `nwords` (cuberille_api.hip reserves them).
// All 64 lanes of the wave must call this together (idx = consecutive outputs, `valid` lanes only).
template <int SHIFT>
__device__ __forceinline__ size_t locate_word_wave(const EmitArgs &a, const u32 *__restrict__ head, size_t nwords, u64 idx,
                                                   bool valid, u32 &within) {
  const int lane = threadIdx.x & 63;
  // lane 0 is always valid; first is a multiple of 64 (taken through readfirstlane: the head's address is then
  // wave-uniform and the load a scalar one -- this kernel is bound by its vector-memory instructions)
  const u64 first = (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)idx) |
                    ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(idx >> 32)) << 32);
  size_t w0 = (size_t)head[first >> 6] & ~(size_t)3;   // (the words of the head's group before it produce earlier outputs)
  size_t found = 0;
  bool done = !valid;
  for (int slide = 0; slide < 4; slide++) {
    const size_t w = w0 + 4 * (size_t)lane;      // this lane's four words
    int t[4];
    if (w < nwords) {
      const uint4 pw = *reinterpret_cast<const uint4 *>(a.prefix + w);
      // (the window lies in one count block or two: their bases through wave-uniform addresses, scalar loads)
      const size_t b0 = w0 >> COUNT_LG, b1 = b0 + 1 < a.nblk ? b0 + 1 : b0;
      const u64 bb0 = a.blockBase[2 * b0 + (SHIFT ? 1 : 0)], bb1 = a.blockBase[2 * b1 + (SHIFT ? 1 : 0)];
      // (the window's 256 words lie in five 64-word segments at most: their prefixes through wave-uniform addresses --
      //  scalar loads -- and a select per lane, instead of one more vector load)
      const size_t s0 = w0 >> 6, sLast = (nwords - 1) >> 6;
      u64 sp = a.segPre[s0];
#pragma unroll
      for (int i = 1; i < 5; i++) {
        const u64 spi = a.segPre[s0 + i <= sLast ? s0 + i : sLast];
        if ((w >> 6) == s0 + i) sp = spi;
      }
      const u64 base = ((w >> COUNT_LG) == b0 ? bb0 : bb1) + (SHIFT ? (sp >> 32) : (sp & 0xffffffffull));
      const u32 p4[4] = {pw.x, pw.y, pw.z, pw.w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        // first output of the word relative to the wave's first output: the head word starts at most one word's worth
        // (< 2^16) below it, everything that matters is <= 63, so 32 bits hold it exactly (large values clamp)
        const long long rel = (long long)(base + ((p4[i] >> SHIFT) & 0xffffu) - first);
        t[i] = (w + i >= nwords || rel > (1 << 20)) ? (1 << 20) : (int)rel;
      }
    } else {
      t[0] = t[1] = t[2] = t[3] = 1 << 20;
    }
    int lo = 0, hi = 64;                         // largest group whose first word starts at or before this lane's output
#pragma unroll
    for (int st = 0; st < 6; st++) {
      const int mid = (lo + hi) >> 1;
      const int v = __shfl(t[0], mid, 64);
      if (v <= lane) lo = mid; else hi = mid;
    }
    const int g0 = __shfl(t[0], lo, 64), g1 = __shfl(t[1], lo, 64), g2 = __shfl(t[2], lo, 64), g3 = __shfl(t[3], lo, 64);
    const int i = g3 <= lane ? 3 : g2 <= lane ? 2 : g1 <= lane ? 1 : 0;
    const int tw = i == 3 ? g3 : i == 2 ? g2 : i == 1 ? g1 : g0;
    // inside the window unless the last loaded word is still <= idx (its successor is unknown)
    if (!done && (lo < 63 || i < 3)) { found = w0 + 4 * (size_t)lo + i; within = (u32)(lane - tw); done = true; }
    if (!__ballot(!done)) return found;
    w0 += 252;
  }
  if (!done) found = locate_word<SHIFT>(a, nwords, idx, within);
  return found;
}

// K3a, search form (the fallback when the vertex-word queue could not be had): one lane per vertex, per-wave
// window search through the head table, or a per-lane search without it.  Covers the counted range (a slab's
// ghost slice included: the rank above needs those coordinates for the triangle split of its first slice).
__global__ __launch_bounds__(256) void k_emit_points_wave(EmitArgs a, Grid g, Geo geo, size_t nwords, u64 nV) {
  const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 r = 0;
  size_t gi;
  if (a.headV) gi = locate_word_wave<0>(a, a.headV, nwords, v, v < nV, r);
  else gi = v < nV ? locate_word<0>(a, nwords, v, r) : 0;
  if (v >= nV) return;
  int k, y, z;
  word_coords(g, gi, k, y, z);
  WordInfo w;
  u32 unk;
  classify_word<false>(a.bits, a.occ, g, a.q1, y, z, k, w, unk);
  int lo = 0, hi = 64;                           // largest bit position with (#created before it) <= r
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) c += popc64(w.C[i] & lowmask(mid));
    if ((u32)c <= r) lo = mid; else hi = mid;
  }
  int before = 0;
  unsigned cm = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    before += popc64(w.C[i] & lowmask(lo));
    cm |= (unsigned)((w.C[i] >> lo) & 1ull) << i;
  }
  const int e = kCornerEnc[select_bit8(cm, (int)r - before)];
  const int cx = k * 64 + lo + (e & 1), cy = y + ((e >> 1) & 1), cz = z + (e >> 2);
  float p[3];
  corner_point(geo, cx, cy, g.zglob0 + cz, p);
  ghost_bottom_mark(g, cz, p);
  float *dst = a.points + 3 * v;                 // ghost points first, owned points from 3*V0 on
  dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
  if (a.cmap) a.cmap[corner_map_index(g, cx, cy, cz)] = (u32)v;
}

// K3a, queue form in two phases per wave (the queue k_count left is dense in lanes and unordered: a word's vertex
// ids follow from its own prefix).  Phase 1, one lane per vertex word: classify, count, wave scan, and a
// walk that only writes 2-byte descriptors (source lane, voxel, corner) into LDS in id order.  Phase 2, one lane
// per VERTEX: descriptor -> lattice point, point store (lanes of one word hold consecutive ids: runs of
// contiguous 12-byte stores), corner-map store.  The point arithmetic and the stores, the expensive part, run on
// full waves instead of on the few lanes that still have vertices left.  A wave with more vertices than its LDS
// slice holds (cannot happen on smooth surfaces) walks and stores directly.
constexpr int POINTS_CAP = 1024;                 // descriptors per wave

// SPLIT lanes share a vertex word, each walking 64 / SPLIT of its voxels (short queues: the walk of phase 1 is serial per
// lane -- up to 64 voxels with corners on the flat faces of the reference's small volumes -- and there are wave slots to spare).
template <int SPLIT>
__global__ __launch_bounds__(256) void k_emit_points_dense(EmitArgs a, Grid g, Geo geo, const u32 *__restrict__ vqueue,
                                                           u32 nVertexWords) {
  __shared__ unsigned short desc[4][POINTS_CAP];
  __shared__ int wk[4][64], wy[4][64], wz[4][64];
  __shared__ u32 woff[4][64];
  __shared__ u64 wv0[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (a.dyn) {
    if (!a.tot->go) return;
    nVertexWords = a.tot->nVertexWords;
  }
  if ((t - lane) / SPLIT >= nVertexWords) return;   // wave-uniform
  const bool valid = t / SPLIT < nVertexWords;
  const u64 mine = SPLIT == 1 ? ~0ull : ((1ull << (64 / SPLIT)) - 1ull) << ((t % SPLIT) * (64 / SPLIT));   // this lane's voxels
  WordInfo w;
  int k = 0, y = 0, z = 0;
  u64 v0 = 0;
  u32 cnt = 0;
  if (valid) {
    const u32 gi = vqueue[t / SPLIT];
    const u32 row = gi / (u32)g.W;
    k = (int)(gi - row * (u32)g.W);
    const u32 zz = row / (u32)g.ny;
    y = (int)(row - zz * (u32)g.ny);
    z = g.cz0 + (int)zz;
    u32 unk;
    classify_word<false>(a.bits, a.occ, g, a.q1, y, z, k, w, unk);
    v0 = seg_base<0>(a, gi) + (a.prefix[gi] & 0xffffu);        // id of this word's first vertex
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (SPLIT > 1) v0 += (u32)popc64(w.C[i] & lowmask((int)(t % SPLIT) * (64 / SPLIT)));   // ... of this lane's first
      cnt += (u32)popc64(w.C[i] & mine);
    }
  }
  const u32 incl = wave_inclusive_sum(cnt);
  const u32 off = incl - cnt;
  const u32 total = __shfl(incl, 63, 64);
  const bool dense = total <= (u32)POINTS_CAP;   // wave-uniform
  if (valid) {
    u64 any = (w.C[0] | w.C[1] | w.C[2] | w.C[3] | w.C[4] | w.C[5] | w.C[6] | w.C[7]) & mine;
    u32 j = off;
    u64 v = v0;
    while (any) {
      const int bx = __ffsll((long long)any) - 1;
      any &= any - 1;
      unsigned cm = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) cm |= (unsigned)((w.C[i] >> bx) & 1ull) << i;
      while (cm) {
        const int e = kCornerEnc[__ffs((int)cm) - 1];
        cm &= cm - 1;
        if (dense) {
          desc[wv][j++] = (unsigned short)((lane << 9) | (bx << 3) | e);
        } else {
          const int cx = k * 64 + bx + (e & 1), cy = y + ((e >> 1) & 1), cz = z + (e >> 2);
          float p[3];
          corner_point(geo, cx, cy, g.zglob0 + cz, p);
          ghost_bottom_mark(g, cz, p);
          float *dst = a.points + 3 * v;
          dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
          if (a.cmap) a.cmap[corner_map_index(g, cx, cy, cz)] = (u32)v;
          v++;
        }
      }
    }
  }
  if (!dense) return;
  wk[wv][lane] = k; wy[wv][lane] = y; wz[wv][lane] = z; woff[wv][lane] = off; wv0[wv][lane] = v0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (u32 i = lane; i < total; i += 64) {
    const unsigned d = desc[wv][i];
    const int src = d >> 9, bx = (d >> 3) & 63, e = d & 7;
    const u64 v = wv0[wv][src] + (i - woff[wv][src]);
    const int cx = wk[wv][src] * 64 + bx + (e & 1), cy = wy[wv][src] + ((e >> 1) & 1), cz = wz[wv][src] + (e >> 2);
    float p[3];
    corner_point(geo, cx, cy, g.zglob0 + cz, p);
    ghost_bottom_mark(g, cz, p);
    float *dst = a.points + 3 * v;
    dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
    if (a.cmap) a.cmap[corner_map_index(g, cx, cy, cz)] = (u32)v;
  }
}

// The four corners of face f of voxel (x, y, z) as vertex indices in this rank's point buffer, in the order of
// txx:197-202: from the dense corner map (MAP) or recomputed; quirk Q1 redirects bottom corners of an aliased slice to
// the top corners of its source slice; an index >= totV names a vertex of the rank below (Grid::extAlias): entry
// index - totV of the plane that rank sent, whose positions stand behind this rank's own points.
template <bool MAP>
__device__ __forceinline__ void quad_corners(const EmitArgs &a, const Grid &g, int x, int y, int z, int f, int zp, u64 (&lid)[4]) {
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int i = kFaceCorner[f][c];
    const int e = kCornerEnc[i];
    const int cx = x + (e & 1), cy = y + ((e >> 1) & 1);
    int cz = z + (e >> 2);
    if (i < 4 && zp >= 0) {
      // bottom corner on an aliased slice (quirk Q1): the reference finds the (x,y) key among the
      // top corners of slice zp if any voxel of zp touching that corner is inside
      bool hit = false;
      for (int ee = 0; ee < 4; ee++) {
        const int vx = cx - (ee & 1), vy = cy - (ee >> 1);
        if (vx >= 0 && vx < g.nx && vy >= 0 && vy < g.ny && getbit(a.bits, g, vx, vy, zp)) hit = true;
      }
      if (hit) cz = zp + 1;
    }
    if (cz == g.nzb + 1) lid[c] = a.tot->totV + (u64)cy * (u64)(g.nx + 1) + (u64)cx;   // a vertex of the rank below
    // MAP: ids from the dense corner map; otherwise recomputed (kept out of the MAP instantiation: its
    // 27-row classification would cost the common kernel a quarter of its wave slots)
    else if (MAP) lid[c] = (u64)a.cmap[corner_map_index(g, cx, cy, cz)];
    else lid[c] = corner_id_generic(a, g, cx, cy, cz);
  }
}

// ... and from those indices the cell(s): global ids, and for triangles the split along the shorter diagonal of the
// PROJECTED quad (txx:286-321; squared distances accumulated in double from the float coordinates, ties -> first form)
template <bool TRI>
__device__ __forceinline__ void finish_cell(const EmitArgs &a, u64 V0, u64 totV, const u64 (&lid)[4], u64 (&o)[TRI ? 6 : 4]) {
  u64 id[4];
#pragma unroll
  for (int c = 0; c < 4; c++) id[c] = lid[c] >= totV ? a.extIds[lid[c] - totV] : lid[c] - V0 + a.pointOffset;
  if constexpr (!TRI) {
    o[0] = id[0]; o[1] = id[1]; o[2] = id[2]; o[3] = id[3];
  } else {
    float v[4][3];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const float *p = a.points + 3 * lid[c];
      v[c][0] = p[0]; v[c][1] = p[1]; v[c][2] = p[2];
    }
    double d02 = 0.0, d13 = 0.0;                                              // I10
#pragma unroll
    for (int t = 0; t < 3; t++) { const double d = (double)v[2][t] - (double)v[0][t]; d02 += d * d; }
#pragma unroll
    for (int t = 0; t < 3; t++) { const double d = (double)v[3][t] - (double)v[1][t]; d13 += d * d; }
    if (d02 >= d13) {                                                         // txx:298-302
      o[0] = id[0]; o[1] = id[1]; o[2] = id[3];
      o[3] = id[1]; o[4] = id[2]; o[5] = id[3];
    } else {                                                                  // txx:303-307
      o[0] = id[0]; o[1] = id[1]; o[2] = id[2];
      o[3] = id[0]; o[4] = id[2]; o[5] = id[3];
    }
  }
}

// the quad that produces output q (owned range): its voxel and face, through the prefix sums
// (all 64 lanes of the wave call this together)
__device__ __forceinline__ bool locate_quad(const EmitArgs &a, const Grid &g, size_t nwords, u64 q, bool valid, u64 Q0, int &x,
                                            int &y, int &z, int &f) {
  u32 r = 0;
  size_t gi;
  // (the head table is over ABSOLUTE outputs, the ghost slice's Q0 quads included: a wave whose first output is not a
  //  multiple of 64 starts its window at the word of the multiple below it -- at or before its own first word)
  if (a.headQ) gi = locate_word_wave<16>(a, a.headQ, nwords, q + Q0, valid, r);
  else gi = valid ? locate_word<16>(a, nwords, q + Q0, r) : 0;
  if (!valid) return false;
  int k;
  word_coords(g, gi, k, y, z);
  u64 F[6];
  faces_word(a.bits, g, y, z, k, F);
  // quads are numbered voxel by voxel: the per-voxel count (0..6) as three bit planes b0 + 2 b1 + 4 b2 -- two full adders, a
  // half adder and one more full adder, two v_bitop3 each -- so that "quads of the voxels below mid" is three masked
  // popcounts instead of six (the search was 180 of the kernel's 630 vector instructions per wave)
  const u64 s1 = bop3<0x96>(F[0], F[1], F[2]), c1 = bop3<0xe8>(F[0], F[1], F[2]);
  const u64 s2 = bop3<0x96>(F[3], F[4], F[5]), c2 = bop3<0xe8>(F[3], F[4], F[5]);
  const u64 b0 = s1 ^ s2, c3 = s1 & s2;
  const u64 b1 = bop3<0x96>(c1, c2, c3), b2 = bop3<0xe8>(c1, c2, c3);
  auto below = [&](int pos) -> int {
    const u64 m = lowmask(pos);
    return popc64(b0 & m) + 2 * popc64(b1 & m) + 4 * popc64(b2 & m);
  };
  int lo = 0, hi = 64;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((u32)below(mid) <= r) lo = mid; else hi = mid;
  }
  const int before = below(lo);
  unsigned fm = 0;
#pragma unroll
  for (int ff = 0; ff < 6; ff++) fm |= (unsigned)((F[ff] >> lo) & 1ull) << ff;
  x = k * 64 + lo;
  f = select_bit8(fm, (int)r - before);
  return true;
}

// a wave's NV ids per quad are 32 or 48 contiguous bytes per lane, 2 or 3 KiB per wave: staged through LDS so that the
// wave writes them as whole 16-byte lanes side by side instead of 64 strided 8-byte pieces per store
template <int NV>
__device__ __forceinline__ void store_wave_cells(u64 *stageWave, u64 *cells, u64 waveFirst, u64 nQ) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const u64 left = nQ - waveFirst;
  const unsigned nPairs = (unsigned)(left < 64 ? left : 64) * (NV / 2);      // 16-byte pieces this wave holds
  const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(stageWave);
  ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(cells + waveFirst * NV);
#pragma unroll
  for (int i = 0; i < NV / 2; i++) {
    const unsigned piece = i * 64 + lane;
    if (piece < nPairs) dst[piece] = src[piece];
  }
}

// K3b: one lane per quad of the owned range; runs AFTER the projection so that the triangle split
// (txx:286-321: along the shorter diagonal of the PROJECTED quad, ties -> first form) is fused in.
// (Which four vertices a quad joins does not depend on the projection, so the lookup half of this kernel was also
//  run on a second stream BESIDE the walk, with a short split pass after both: the walk slowed from 1.18 to 1.52 ms
//  while the cell pass shrank from 0.45 to 0.15 ms -- both kernels want the vector issue slots; not kept.)
template <bool TRI, bool MAP>
__global__ __launch_bounds__(256) void k_emit_cells(EmitArgs a, Grid g, size_t nwords, u64 nQ) {
  constexpr int NV = TRI ? 6 : 4;                // ids per quad
  __shared__ u64 stage[4][64 * NV];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 waveFirst = q - lane;
  if (a.dyn) {
    if (!a.tot->go) return;
    nQ = a.tot->totQ - a.tot->Q0;
  }
  if (waveFirst >= nQ) return;                   // wave-uniform
  if (a.rows) {
    // the id offset from the gathered totals of the ranks below; a flag anywhere and the step is taken again by the host
    u32 flags = 0;
    u64 off = 0;
    for (int r = 0; r < a.nRanks; r++) {
      flags |= row_flags(a.rows, r);
      if (r < a.rank) off += a.rows[r].totV - a.rows[r].V0;
    }
    if (flags) return;
    a.pointOffset += off;
  }
  const u64 V0 = a.tot->V0, Q0 = a.tot->Q0, totV = a.tot->totV;
  int x, y, z = 0, f;
  const bool have = locate_quad(a, g, nwords, q, q < nQ, Q0, x, y, z, f);
  // quirk Q1's source slice of the quad's slice: the 64 quads of a wave nearly always share a slice -- one look-up through
  // scalar loads then (lane 0 always holds a quad) instead of one vector load per lane: this kernel is bound by those
  int zp;
  {
    u32 unk;
    const int zfirst = __builtin_amdgcn_readfirstlane(z);
    if (__ballot(have && z != zfirst) == 0ull) zp = alias_of(a.occ, g, a.q1, zfirst, unk);
    else zp = have ? alias_of(a.occ, g, a.q1, z, unk) : -1;
  }
  if (have) {
    u64 lid[4], o[NV];
    quad_corners<MAP>(a, g, x, y, z, f, zp, lid);
    finish_cell<TRI>(a, V0, totV, lid, o);
#pragma unroll
    for (int i = 0; i < NV; i++) stage[wv][lane * NV + i] = o[i];
  }
  store_wave_cells<NV>(stage[wv], a.cells, waveFirst, nQ);
}

// ---------------------------------------------------------------------------------------------
// K4: projection (txx:439-474) with the gradient image (txx:478-498) evaluated on the fly.
// Every arithmetic step mirrors the ITK 3.x contract I3..I9 in the same
// operation order as the oracle, so the float coordinates come out bit-identical.
// ---------------------------------------------------------------------------------------------
template <class T>
struct Sampler {
  const T *vox;
  int nx, ny, nzb;
  int zglob0, gnz;
  __device__ __forceinline__ int zlocal(int zg) const {          // global z -> buffer slice
    const int z = zg - zglob0;
    return z < 0 ? 0 : (z > nzb - 1 ? nzb - 1 : z);
  }
  __device__ __forceinline__ T at(int x, int y, int zg) const {   // global z in
    return vox[((size_t)zlocal(zg) * ny + y) * nx + x];
  }
  __device__ __forceinline__ T at_clamped(int x, int y, int zg) const {
    x = x < 0 ? 0 : (x > nx - 1 ? nx - 1 : x);
    y = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y);
    zg = zg < 0 ? 0 : (zg > gnz - 1 ? gnz - 1 : zg);
    return at(x, y, zg);
  }
};

__device__ __forceinline__ int to_index_clamped(double b, int end) {
  // branch-free: fmax(NaN, 0) is 0, so a NaN coordinate (quirk Q4) lands on index 0 and never reads
  // out of bounds; b is integral (a floor), so the clamp in double and the conversion are exact
  return (int)fmin(fmax(b, 0.0), (double)end);
}

struct Cell8 {
  int bc[3];       // floor of the continuous index clamped to [-1, end]: lo and hi are functions of it
  int lo[3], hi[3];
  double d[3];
};

__device__ __forceinline__ void make_cell(const Geo &geo, bool unitP2I, const int n[3], const double p[3], Cell8 &c,
                                          bool diagP2I = false) {
  double cv[3], ci[3];
#pragma unroll
  for (int k = 0; k < 3; k++) cv[k] = p[k] - geo.origin[k];                // I4
  if (diagP2I) {
    // a diagonal PhysicalPointToIndex matrix (identity direction, any spacing): 0 + m*a + 0*b + 0*c is m*a + 0 while b and c
    // are finite, and the zero's sign washes out as below; with a NaN or an infinity among b, c the full form makes ALL three
    // indices NaN where this one keeps a's -- the same pass either way: one NaN weight factor makes all eight weights NaN,
    // the value and the normal with them, and the vertex is NaN in every coordinate from then on
#pragma unroll
    for (int r = 0; r < 3; r++) ci[r] = geo.p2i[4 * r] * cv[r];
  } else
  if (unitP2I) {
    // identity PhysicalPointToIndex matrix: 0 + 1*a + 0*b + 0*c is a + 0 for every a, b, c a vertex can hold
    // (finite or NaN), i.e. a itself except that -0 becomes +0 -- and ci is only used through floor(ci) and
    // ci - floor(ci), which give the same index and the same +0 fraction for either zero
#pragma unroll
    for (int r = 0; r < 3; r++) ci[r] = cv[r];
  } else {
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) sum += geo.p2i[r * 3 + k] * cv[k];
      ci[r] = sum;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double b = floor(ci[k]);
    c.d[k] = ci[k] - b;
    // lo = clamp(b, 0, end), hi = clamp(b + 1, 0, end) on integers: the conversion saturates (and sends the NaN of
    // quirk Q4 to 0, so a NaN vertex still reads inside the image; its coordinates stay NaN whatever it reads)
    int bi;
    asm("v_cvt_i32_f64_e32 %0, %1" : "=v"(bi) : "v"(b));
    const int end = n[k] - 1;
    // (the interpolators clamp into [StartIndex, EndIndex] of the buffered region.  The cell is NAMED by its clamped INDEX --
    //  the bounds are scalars, so a pass costs what it cost without a start index -- and only a gather turns that into
    //  positions in the volume)
    const int bcI = min(max(bi, geo.istart[k] - 1), end + geo.istart[k]);
    c.bc[k] = bcI;
    const int bc = bcI - geo.istart[k];
    c.lo[k] = max(bc, 0);
    c.hi[k] = min(bc + 1, end);
  }
}

__device__ __forceinline__ bool finite_f(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

// I6: GradientImageFilter at one pixel
__device__ __forceinline__ void gradient_from_taps(const Geo &geo, bool dirIdentity, const float fm[3], float f0,
                                                   const float fp[3], float out[3]) {
  float local[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float c = geo.gcoef[a];
    float sum = 0.0f;
    sum += (-c) * fm[a];
    sum += 0.0f * f0;
    sum += c * fp[a];
    local[a] = sum;
  }
  if (dirIdentity && finite_f(local[0]) && finite_f(local[1]) && finite_f(local[2])) {
    // TransformLocalVectorToPhysicalVector with the identity matrix: 0 + 1*v + 0*w + 0*u, which is
    // v + 0 for finite inputs (turns -0 into +0 exactly like the general form below)
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = local[r] + 0.0f;
    return;
  }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) sum = (float)((double)sum + geo.dir[r * 3 + c] * (double)local[c]);
    out[r] = sum;
  }
}

template <class T>
__device__ __forceinline__ void gradient_at(const Sampler<T> &s, const Geo &geo, bool dirIdentity, int x, int y, int z,
                                            float f0, float out[3]) {
  float fm[3], fp[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const int dx = (a == 0), dy = (a == 1), dz = (a == 2);
    fm[a] = (float)s.at_clamped(x - dx, y - dy, z - dz);
    fp[a] = (float)s.at_clamped(x + dx, y + dy, z + dz);
  }
  gradient_from_taps(geo, dirIdentity, fm, f0, fp, out);
}

// Register type of the eight cached site values: the walk needs them as doubles; 1- and 2-byte integer pixels are
// exact in a float, so their cache is half the registers (and is widened at use -- those kernels are short of
// registers, and their walks are short).
template <class T>
struct SiteValue {
  typedef typename std::conditional<std::is_integral<T>::value && sizeof(T) <= 2, float, double>::type type;
};

// the eight lattice-site values and gradients of a cell from its gathered 4x4x4 neighbourhood (32 entries used)
template <class T, bool LITERAL>
__device__ __forceinline__ void cell_gradients(const Geo &geo, bool dirIdentity, const T (&V)[4][4][4], float G[8][3],
                                               typename SiteValue<T>::type Vd[8]) {
#pragma unroll
  for (int counter = 0; counter < 8; counter++) {
    const int a = (counter & 1) + 1, b = ((counter >> 1) & 1) + 1, cz = (counter >> 2) + 1;
    const T pix = V[cz][b][a];
    const float fm[3] = {(float)V[cz][b][a - 1], (float)V[cz][b - 1][a], (float)V[cz - 1][b][a]};
    const float fp[3] = {(float)V[cz][b][a + 1], (float)V[cz][b + 1][a], (float)V[cz + 1][b][a]};
    Vd[counter] = (typename SiteValue<T>::type)pix;
    if (LITERAL) {
      gradient_from_taps(geo, dirIdentity, fm, (float)pix, fp, G[counter]);
    } else {
      float local[3];
#pragma unroll
      for (int a_ = 0; a_ < 3; a_++) local[a_] = (-geo.gcoef[a_]) * fm[a_] + geo.gcoef[a_] * fp[a_];
      if (dirIdentity) {                         // wave-uniform
#pragma unroll
        for (int r = 0; r < 3; r++) G[counter][r] = local[r];
      } else {
#pragma unroll
        for (int r = 0; r < 3; r++) {
          float sum = 0.0f;
#pragma unroll
          for (int cc = 0; cc < 3; cc++) sum = (float)((double)sum + geo.dir[r * 3 + cc] * (double)local[cc]);
          G[counter][r] = sum;
        }
      }
    }
  }
}

// Gather of the cell around a vertex: the eight lattice-site pixel values and gradients.  When the
// cell lies inside the image (lo+1 == hi on every axis) the 56 taps are 32 distinct pixels: all 32
// loads are issued back to back (one memory latency per gather) from 12 row segments.
// LITERAL = false shortens the gradient of an interior cell to (-c)*f(-1) + c*f(+1) per axis: with finite taps
// that is the reference's 0 + (-c)*f(-1) + 0*f(0) + c*f(+1) (and, for the identity direction, its 0 + 1*v + 0*w + 0*u)
// up to the SIGN OF A ZERO result, and a zero gradient component only ever enters the walk as a term added to a
// sum that starts at +0 -- so the walk cannot tell.  Non-finite taps always give a non-finite component here too;
// the caller tests for that and gathers again with LITERAL = true.
template <class T, bool LITERAL>
__device__ __forceinline__ void gather_cell(const Sampler<T> &s, const Geo &geo, bool dirIdentity, const Cell8 &c,
                                            float G[8][3], typename SiteValue<T>::type Vd[8]) {
  const bool unit = c.lo[0] + 1 == c.hi[0] && c.lo[1] + 1 == c.hi[1] && c.lo[2] + 1 == c.hi[2];
  const int zl = c.lo[2] - s.zglob0;              // buffer slice of the cell's lower z
  const bool interior = unit && c.lo[0] >= 1 && c.lo[0] + 2 < s.nx && c.lo[1] >= 1 && c.lo[1] + 2 < s.ny && zl >= 1 && zl + 2 < s.nzb;
  // one address form per gather pass: the immediate-offset form only when EVERY lane gathering now sits in the
  // interior; a wave with border cells among them takes the clamped form for all its unit cells (it is the same
  // 32 pixels for an interior cell), instead of running both forms one after the other
  const bool allInterior = __ballot(unit && !interior) == 0ull;
  if (interior && allInterior) {
    // the cell and its ring of neighbours lie inside the buffer: nothing is clamped, so the 12 row segments are
    // the cell's own address plus wave-uniform strides, and the x neighbours are immediate offsets
    const T *base = s.vox + ((size_t)zl * s.ny + c.lo[1]) * s.nx + c.lo[0];
    const ptrdiff_t rowS = (ptrdiff_t)s.nx, sliceS = (ptrdiff_t)s.ny * s.nx;
    T V[4][4][4];
#pragma unroll
    for (int zi = 0; zi < 4; zi++)
#pragma unroll
      for (int yi = 0; yi < 4; yi++) {
        const bool zin = (zi == 1 || zi == 2), yin = (yi == 1 || yi == 2);
        if (!zin && !yin) continue;
        const T *row = base + (zi - 1) * sliceS + (yi - 1) * rowS;
        if (zin && yin) {
#pragma unroll
          for (int xi = 0; xi < 4; xi++) V[zi][yi][xi] = row[xi - 1];
        } else {
          V[zi][yi][1] = row[0];
          V[zi][yi][2] = row[1];
        }
      }
    cell_gradients<T, LITERAL>(geo, dirIdentity, V, G, Vd);
  } else if (unit) {
    int xs[4], ys[4], zs[4];
    xs[0] = c.lo[0] > 0 ? c.lo[0] - 1 : 0;  xs[1] = c.lo[0];  xs[2] = c.hi[0];  xs[3] = c.hi[0] < s.nx - 1 ? c.hi[0] + 1 : s.nx - 1;
    ys[0] = c.lo[1] > 0 ? c.lo[1] - 1 : 0;  ys[1] = c.lo[1];  ys[2] = c.hi[1];  ys[3] = c.hi[1] < s.ny - 1 ? c.hi[1] + 1 : s.ny - 1;
    zs[0] = s.zlocal(c.lo[2] > 0 ? c.lo[2] - 1 : 0);  zs[1] = s.zlocal(c.lo[2]);  zs[2] = s.zlocal(c.hi[2]);
    zs[3] = s.zlocal(c.hi[2] < s.gnz - 1 ? c.hi[2] + 1 : s.gnz - 1);
    T V[4][4][4];      // [z][y][x]; only the 32 entries below are ever touched
#pragma unroll
    for (int zi = 0; zi < 4; zi++)
#pragma unroll
      for (int yi = 0; yi < 4; yi++) {
        const bool zin = (zi == 1 || zi == 2), yin = (yi == 1 || yi == 2);
        if (!zin && !yin) continue;
        const T *row = s.vox + ((size_t)zs[zi] * s.ny + ys[yi]) * s.nx;
        if (zin && yin) {
#pragma unroll
          for (int xi = 0; xi < 4; xi++) V[zi][yi][xi] = row[xs[xi]];
        } else {
          V[zi][yi][1] = row[xs[1]];
          V[zi][yi][2] = row[xs[2]];
        }
      }
    cell_gradients<T, LITERAL>(geo, dirIdentity, V, G, Vd);
  } else {
    // vertex on or beyond the image border: clamped neighbours, generic taps
#pragma unroll
    for (int counter = 0; counter < 8; counter++) {
      const int nx_ = (counter & 1) ? c.hi[0] : c.lo[0];
      const int ny_ = (counter & 2) ? c.hi[1] : c.lo[1];
      const int nz_ = (counter & 4) ? c.hi[2] : c.lo[2];
      const T pix = s.at(nx_, ny_, nz_);
      Vd[counter] = (typename SiteValue<T>::type)pix;
      gradient_at(s, geo, dirIdentity, nx_, ny_, nz_, (float)pix, G[counter]);
    }
  }
}

// The walk takes 1..max_steps+2 iterations per vertex, so a plain lane-per-vertex launch idles
// most lanes behind the slowest vertex of each wave.  Here every wave owns a contiguous chunk of
// vertices and REFILLS lanes whose vertex has converged (when at least REFILL lanes are idle),
// so the f64-bound loop body runs with nearly full waves until the chunk is drained.
// The eight lattice-site gradients and pixel values of the cell a vertex sits in are kept in
// registers and only re-gathered when the walk enters another cell (steps are <= a quarter voxel
// and shrink): the per-iteration work is then the trilinear weights, 32 multiply-adds, one sqrt
// and three divides, all in f64 in the reference's operation order.
// MODE 0: every slice a walk can reach is in the buffer.  MODE 1 (THIN_HALO slab): a walk that enters a cell whose
// slices (gradient ring included) the buffer lacks although the volume has them is not clamped: the vertex goes on
// the escape list untouched and is walked again from its start once the deeper halo is there (MODE 2: the vertices
// are taken from that list; `g` then describes the deeper buffer).  dyn (cuberille_step_begin): the launch was sized
// blindly, the real counts are read from `tot`.
// GEOM 2 (IDENT): the PhysicalPointToIndex matrix and the direction matrix are both the identity and the buffered region starts at index 0
// (unit spacing, no rotation, an image as read from a file -- every volume the reference ships and the bench's): their 39
// scalars then never enter the kernel, which keeps its whole argument
// block in SGPRs and spills what does not fit (62 of them before) into the lanes of a vector register its loop reads back.
// GEOM 2 = that; 1: identity direction, region at 0, ANY spacing (a diagonal matrix: three scalars instead of eighteen -- the
// anisotropic volumes of CT and MR); 0: anything (a rotation, a region that starts elsewhere).
template <class T, int MODE, int GEOM>
__global__ __launch_bounds__(256, (sizeof(T) == 8 ? 3 : 4)) void k_project(const T *__restrict__ vox, Grid g, Geo geo, Params prm, int dirIdentityArg,
                                                 float *__restrict__ points, u64 nPoints, u64 nGhost, u64 chunk,
                                                 int REFILL, int xcdRemap, int forceLiteral, Totals *__restrict__ tot,
                                                 u32 *__restrict__ escList, u32 escCap, int dyn) {
  const int lane = threadIdx.x & 63;
  if (dyn) {
    if (!tot->go) return;
    nPoints = tot->totV;
    nGhost = tot->V0;
  }
  // blocks are dealt round-robin over the 8 XCDs (b and b+8 share one, each XCD has its own L2): give
  // every XCD one contiguous eighth of the vertex list so that chunks whose cells overlap (adjacent
  // rows and slices) meet in the same L2.  Placement only affects speed, never results.
  u64 lb = blockIdx.x;
  if (xcdRemap == 1) {
    const u64 nb = gridDim.x, per = nb / 8, rem = nb % 8, x = lb % 8, j = lb / 8;
    lb = x * per + (x < rem ? x : rem) + j;
  } else if (xcdRemap > 1) {
    // runs of G consecutive workgroups' worth of batches on one XCD -- workgroup b sits on XCD b % 8 -- so that batches whose
    // cells overlap (neighbouring rows) meet in one L2; super-groups of 8 G workgroups, the grid's ragged end left as it is
    // (launch_project: short walks)
    const u64 G = (u64)xcdRemap, nb = gridDim.x, sg = lb / (8 * G);
    if ((sg + 1) * 8 * G <= nb) {
      const u64 in = lb - sg * 8 * G, x = in % 8, j = in / 8;
      lb = sg * 8 * G + x * G + j;
    }
  }
  // Work assignment: the vertex list is cut into batches of `chunk` (a power of two) vertices; wave w
  // takes batches w, w+NW, w+2NW, ... (NW = waves in the grid).  With the grid resident, all waves
  // advance through the list together, so the vertices in flight on the chip stay inside a narrow
  // band of slices; `next` counts positions in the wave's own sequence.
  const u64 wave = (lb * blockDim.x + threadIdx.x) >> 6;
  const u64 NW = ((u64)gridDim.x * blockDim.x) >> 6;
  const int lgChunk = 63 - __clzll((long long)chunk);
  const u64 nBatches = (nPoints + chunk - 1) >> lgChunk;
  if (wave >= nBatches) return;
  u64 next = 0;                                   // wave-uniform cursor
  const u64 end = ((nBatches - wave + NW - 1) / NW) << lgChunk;
  Sampler<T> s{vox, g.nx, g.ny, g.nzb, (int)g.zglob0, (int)g.gnz};
  const int n[3] = {g.nx, g.ny, (int)g.gnz};
  const double iso = (double)iso_as<T>(prm.iso, prm.isoInt);
  unsigned myIters = 0;
  u32 stopStepsW = 0;                              // wave-uniform: owned walks of this wave that ran out of steps
  bool escapedW = false;                           // wave-uniform: some walk of this wave left the buffer (MODE 1)
  // (both are only ever updated where the whole wave passes, from lane flags set inside the divergent walk)
  bool active = false;
  u64 idx = 0;
  float vertex[3] = {0.f, 0.f, 0.f};
  double step = 0.0;
  unsigned numberOfSteps = 0;
  constexpr int NO_CELL = -0x7fffffff - 1;         // (no clamped index is that low: start indices lie within +-2^30)
  int kc[3] = {NO_CELL, NO_CELL, NO_CELL};         // cell held in registers, named by its clamped floor indices
  float G[8][3];
  double Gd[8][3];
  typename SiteValue<T>::type Vd[8];
  bool cellFinite = false;
  constexpr bool IDENT = GEOM == 2, DIAG = GEOM == 1;
  bool unitP2I = true;
  if constexpr (GEOM == 0) {
#pragma unroll
    for (int i = 0; i < 9; i++) unitP2I = unitP2I && (geo.p2i[i] == ((i % 4 == 0) ? 1.0 : 0.0));
  } else {
    // (constants from here on: the rare paths that multiply by the matrices -- a non-finite tap -- read these, not the arguments)
#pragma unroll
    for (int i = 0; i < 9; i++) {
      geo.dir[i] = (i % 4 == 0) ? 1.0 : 0.0;
      if (IDENT || i % 4 != 0) geo.p2i[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    geo.istart[0] = geo.istart[1] = geo.istart[2] = 0;
    unitP2I = IDENT;
  }
  const int dirIdentity = GEOM ? 1 : dirIdentityArg;
  for (;;) {
    const u64 idle = __ballot(!active);
    if (idle && next < end && (__popcll(idle) >= REFILL || idle == ~0ull)) {
      const u64 remaining = end - next;
      if (!active) {
        const u64 rank = (u64)__popcll(idle & lowmask(lane));
        const u64 pos = next + rank;
        const u64 cand = ((wave + (pos >> lgChunk) * NW) << lgChunk) + (pos & (chunk - 1));
        if (rank < remaining && cand < nPoints) {
          idx = MODE == 2 ? (u64)escList[cand] : cand;
          vertex[0] = points[3 * idx]; vertex[1] = points[3 * idx + 1]; vertex[2] = points[3 * idx + 2];
          step = prm.step;
          numberOfSteps = 0;
          kc[0] = NO_CELL;
          // (a ghost vertex on the ghost slice's BOTTOM plane was written as NaN by the point pass: no cell of this
          //  rank touches that plane, its position is the rank below's business)
          active = !(idx < nGhost && vertex[0] != vertex[0]);
        }
      }
      const u64 take = (u64)__popcll(idle);
      next += take < remaining ? take : remaining;
    }
    if (!__ballot(active)) break;
    bool bySteps = false, escaped = false;
    if (active) {
      bool done = false;
      const double p[3] = {(double)vertex[0], (double)vertex[1], (double)vertex[2]};
      Cell8 c;
      make_cell(geo, unitP2I, n, p, c, DIAG);
      if (MODE == 1 && (c.bc[0] != kc[0] || c.bc[1] != kc[1] || c.bc[2] != kc[2])) {
        // global slices the cell and its gradient ring read, against the buffer
        const int zlo = max(c.lo[2] - 1, 0), zhi = min(c.hi[2] + 1, n[2] - 1);
        escaped = zlo < s.zglob0 || zhi > s.zglob0 + s.nzb - 1;
        if (escaped) {
          const u32 slot = atomicAdd(&tot->nEscaped, 1u);
          if (slot < escCap && idx <= 0xffffffffull) escList[slot] = (u32)idx;
          else atomicOr(&tot->err, (u32)ERRF_ESCAPE_OVERFLOW);
          active = false;                            // the point keeps its start position
        }
      }
      if (!escaped) {
      if (c.bc[0] != kc[0] || c.bc[1] != kc[1] || c.bc[2] != kc[2]) {
        gather_cell<T, false>(s, geo, dirIdentity != 0, c, G, Vd);
#pragma unroll
        for (int k = 0; k < 3; k++) kc[k] = c.bc[k];
        // all 32 cached numbers finite?  x*0 accumulates to 0 for finite x, to NaN for an infinity or a NaN
        float tf = 0.0f;
        double td = 0.0;
#pragma unroll
        for (int counter = 0; counter < 8; counter++) {
          td = __builtin_fma((double)Vd[counter], 0.0, td);
#pragma unroll
          for (int k = 0; k < 3; k++) tf = __builtin_fmaf(G[counter][k], 0.0f, tf);
        }
        cellFinite = (tf == 0.0f) && (td == 0.0);
        if (!cellFinite) gather_cell<T, true>(s, geo, dirIdentity != 0, c, G, Vd);   // rare: the reference's formula to the letter
#pragma unroll
        for (int counter = 0; counter < 8; counter++)
#pragma unroll
          for (int k = 0; k < 3; k++) Gd[counter][k] = (double)G[counter][k];
      }
      // I7 (gradient, txx:451) and I5 (value, txx:455) share the cell and the weights.  The reference
      // loop skips zero weights and stops once the accumulated weight is exactly 1.  With finite
      // pixels a zero weight adds nothing, so only the early stop can change bits: when no partial
      // sum of the weights hits 1.0 before the last term (the usual case) the eight terms are summed
      // straight; otherwise, or with non-finite pixels in the cell, the loop is replayed literally.
      double o[8];
#pragma unroll
      for (unsigned counter = 0; counter < 8; counter++) {
        double overlap = 1.0;
#pragma unroll
        for (int k = 0; k < 3; k++) overlap *= (counter & (1u << k)) ? c.d[k] : (1.0 - c.d[k]);
        o[counter] = overlap;
      }
      bool literal = !cellFinite || forceLiteral;
      // Can a partial sum of the weights be exactly 1.0 before the last term?  The weights are >= 0, so the rounded partial
      // sums never decrease and none exceeds the rounded sum of the first seven, S7.  The eight weights are the rounded
      // products of (d, fl(1 - d)) pairs, each pair summing to 1 within 2^-53 and every product within 2^-52 of exact: all
      // eight add up to 1 within 1e-15, so S7 <= 1 + 1e-15 - o[7], and seven rounded additions add at most 8e-16 more.
      // With o[7] > 1e-14 every partial sum is below 1: ONE comparison instead of six dependent f64 additions per pass
      // (round 5; ~3 % of a pass).  A last weight that small -- a vertex within a hair of a cell face -- or a NaN takes the
      // literal loop, which is always right.
      literal |= !(o[7] > 1e-14);
      double acc[3] = {0.0, 0.0, 0.0}, value = 0.0;
      if (!literal) {
        // (the first term of each sum is 0.0 + o*g: the product rounded once, and a -0 product made +0 -- which is fma(o, g, +0.0)
        //  bit for bit, one instruction instead of two; the other seven terms stay a multiplication and an addition, rounded twice
        //  like the reference's)
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] = __builtin_fma(o[0], Gd[0][k], 0.0);
        value = __builtin_fma(o[0], (double)Vd[0], 0.0);
#pragma unroll
        for (int counter = 1; counter < 8; counter++) {
#pragma unroll
          for (int k = 0; k < 3; k++) acc[k] += o[counter] * Gd[counter][k];
          value += o[counter] * (double)Vd[counter];
        }
      } else {
        double total = 0.0;
#pragma unroll
        for (int counter = 0; counter < 8; counter++) {
          if (o[counter] != 0.0 && total != 1.0) {   // "if (overlap)" + "break once total == 1"
#pragma unroll
            for (int k = 0; k < 3; k++) acc[k] += o[counter] * Gd[counter][k];
            value += o[counter] * (double)Vd[counter];
            total += o[counter];
          }
        }
      }
      done = fabs(value - iso) < prm.thr;                                     // txx:456
      const unsigned passes = numberOfSteps + 1;       // loop passes of this vertex if it ends in this one
      if (!done) {
        // (the reference normalises before the test, txx:452; the normal is only used when stepping)
        float normal[3] = {(float)acc[0], (float)acc[1], (float)acc[2]};
        // I8: 0.0 + e0*e0 + e1*e1 + e2*e2 -- a square is never -0, so the leading 0.0 adds nothing
        const double e0 = (double)normal[0], e1 = (double)normal[1], e2 = (double)normal[2];
        double sq = e0 * e0;
        sq += e1 * e1;
        sq += e2 * e2;
        if (sq > 0.0 && sq < __builtin_inf()) {
          // sq is a sum of squares of floats, so it lies in [2^-298, 2^258): the compiler's f64 sqrt (rsq + the
          // refinement below, wrapped in a rescaling for arguments under 2^-767 and a pass-through for 0 and
          // infinity) reduces to exactly these instructions.
          const double y0 = __builtin_amdgcn_rsq(sq);
          double gs = sq * y0, hs = y0 * 0.5;
          const double rs = __builtin_fma(-hs, gs, 0.5);
          gs = __builtin_fma(gs, rs, gs);
          hs = __builtin_fma(hs, rs, hs);
          double ds = __builtin_fma(-gs, gs, sq);
          gs = __builtin_fma(ds, hs, gs);
          ds = __builtin_fma(-gs, gs, sq);
          const double norm = __builtin_fma(ds, hs, gs);
          // Three quotients by one denominator.  The compiler expands an f64 `/` into div_scale, rcp, two Newton
          // steps on the reciprocal, q0 = n*y, r = fma(-d, q0, n), q = fma(r, y, q0), div_fixup.  Here every
          // numerator is a float (zero, or 2^-149 <= |n| < 2^128) and d = sqrt of their squares is finite and
          // positive, so div_scale never scales and div_fixup only restores the sign of a zero quotient: the
          // same instructions with the reciprocal refined ONCE give the same bits (IEEE-rounded quotients).
          double y = __builtin_amdgcn_rcp(norm);
          double e = __builtin_fma(-norm, y, 1.0);
          y = __builtin_fma(y, e, y);
          e = __builtin_fma(-norm, y, 1.0);
          y = __builtin_fma(y, e, y);
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const double x = (double)normal[k];
            const double q0 = x * y;
            const double r = __builtin_fma(-norm, q0, x);
            const double q = __builtin_fma(r, y, q0);
            normal[k] = (float)__builtin_copysign(q, x);     // -0 / d is -0 (the fma chain alone would give +0)
          }
        } else {
          const double norm = sqrt(sq);
#pragma unroll
          for (int k = 0; k < 3; k++) normal[k] = (float)((double)normal[k] / norm);
        }
        // txx:463-467 (I9): normal * sign * step with sign = +-1 -- (n * +-1) * s and n * (+-s) are the same rounded product with the
        // same sign, zeros included: one multiplication per axis instead of two
        const double signedStep = (value < iso) ? step : -step;
#pragma unroll
        for (int k = 0; k < 3; k++)
          vertex[k] = (float)((double)vertex[k] + ((double)normal[k] * signedStep));
        step *= prm.relax;                                                    // txx:468
        done = numberOfSteps++ > prm.max_steps;                               // txx:469
        bySteps = done && idx >= nGhost;                                      // txx:470-472's counter
      }
      if (done) {
        points[3 * idx] = vertex[0]; points[3 * idx + 1] = vertex[1]; points[3 * idx + 2] = vertex[2];
        if (idx >= nGhost) myIters += passes;          // the iteration statistic counts owned vertices only
        active = false;
      }
      }
    }
    // (a scalar per wave; the walks that end within the threshold are the rest)
    stopStepsW += (u32)__popcll(__ballot(bySteps));
    if (MODE == 1) escapedW = escapedW || __ballot(escaped) != 0ull;
  }
  // one atomic per wave for the iteration statistic
  unsigned sum = myIters;
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) sum += __shfl_down(sum, sft, 64);
  if (lane == 0 && sum) atomicAdd(&tot->iters, (u64)sum);
  if (lane == 0 && stopStepsW) atomicAdd(&tot->stopSteps, (u64)stopStepsW);
  if (MODE == 1 && lane == 0 && escapedW) atomicOr(&tot->err, (u32)ERRF_ESCAPE);
}

// ---------------------------------------------------------------------------------------------
// K4b: the reference's two OTHER projection branches, which it compiles out (h:22-23 set both macros to 0):
// USE_ADVANCED_PROJECTION (txx:340-397) and USE_LINESEARCH_PROJECTION (txx:398-437).  Offered for builds of the
// reference that switch one on; never on the default path, so they are written plainly -- one lane per vertex,
// every interpolation replayed to the letter (zero weights skipped, stop once the weights sum to 1), IEEE sqrt and
// divide -- and share only the gather with k_project.
// ---------------------------------------------------------------------------------------------
template <class T>
struct VariantCtx {
  Sampler<T> s;
  Geo geo;
  bool dirIdentity, unitP2I;
  int n[3];
  double iso;
  const double *gimg;   // the recursive-Gaussian variant's gradient image (CovariantVector<double,3> per pixel), or null
  HeldGradient held;    // quirk Q3 on request: the gradient image (and geometry) of an earlier volume, or img == null
  bool heldUnitP2I;
};

// I7 + I8: the interpolated gradient at `vertex`, normalised (txx:356-357, 408-409, 451-452), as the double vector the
// step multiplies.  The shipped gradient type is CovariantVector<float,3>: the interpolated sum is narrowed to float
// and Normalize() rounds each component to float again.  With USE_GRADIENT_RECURSIVE_GAUSSIAN the gradient image holds
// doubles (x.gimg): interpolation and Normalize() stay in double.
template <class T>
__device__ void variant_normal(const VariantCtx<T> &x, const float vertex[3], double nd[3]) {
  const double p[3] = {(double)vertex[0], (double)vertex[1], (double)vertex[2]};
  Cell8 c;
  make_cell(x.geo, x.unitP2I, x.n, p, c);
  if (x.gimg) {
    double acc[3] = {0.0, 0.0, 0.0}, total = 0.0;
#pragma unroll
    for (unsigned counter = 0; counter < 8; counter++) {
      double overlap = 1.0;
#pragma unroll
      for (int k = 0; k < 3; k++) overlap *= (counter & (1u << k)) ? c.d[k] : (1.0 - c.d[k]);
      if (overlap != 0.0 && total != 1.0) {
        const int sx = (counter & 1) ? c.hi[0] : c.lo[0], sy = (counter & 2) ? c.hi[1] : c.lo[1], sz = (counter & 4) ? c.hi[2] : c.lo[2];
        const double *gp = x.gimg + 3 * (((size_t)sz * x.n[1] + sy) * x.n[0] + sx);
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] += overlap * gp[k];
        total += overlap;
      }
    }
    double sq = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) sq += acc[k] * acc[k];
    const double norm = sqrt(sq);
#pragma unroll
    for (int k = 0; k < 3; k++) nd[k] = acc[k] / norm;
    return;
  }
  float normal[3];
  double acc[3] = {0.0, 0.0, 0.0}, total = 0.0;
  if (x.held.img) {
    // the cached interpolator of txx:484: the point goes through the HELD image's geometry, the eight sites come from its
    // gradient image (I7), clamped to ITS extent (I5)
    Cell8 h;
    make_cell(x.held.geo, x.heldUnitP2I, x.held.n, p, h);
#pragma unroll
    for (unsigned counter = 0; counter < 8; counter++) {
      double overlap = 1.0;
#pragma unroll
      for (int k = 0; k < 3; k++) overlap *= (counter & (1u << k)) ? h.d[k] : (1.0 - h.d[k]);
      if (overlap != 0.0 && total != 1.0) {
        const int sx = (counter & 1) ? h.hi[0] : h.lo[0], sy = (counter & 2) ? h.hi[1] : h.lo[1], sz = (counter & 4) ? h.hi[2] : h.lo[2];
        const float *gp = x.held.img + 3 * (((size_t)sz * x.held.n[1] + sy) * x.held.n[0] + sx);
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] += overlap * (double)gp[k];
        total += overlap;
      }
    }
  } else {
    float G[8][3];
    typename SiteValue<T>::type Vd[8];
    gather_cell<T, true>(x.s, x.geo, x.dirIdentity, c, G, Vd);
#pragma unroll
    for (unsigned counter = 0; counter < 8; counter++) {
      double overlap = 1.0;
#pragma unroll
      for (int k = 0; k < 3; k++) overlap *= (counter & (1u << k)) ? c.d[k] : (1.0 - c.d[k]);
      if (overlap != 0.0 && total != 1.0) {
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] += overlap * (double)G[counter][k];
        total += overlap;
      }
    }
  }
  double sq = 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++) { normal[k] = (float)acc[k]; const double e = (double)normal[k]; sq += e * e; }
  const double norm = sqrt(sq);
#pragma unroll
  for (int k = 0; k < 3; k++) nd[k] = (double)(float)((double)normal[k] / norm);
}

// I5: the interpolated pixel value at `q` (txx:368-369, 424)
template <class T>
__device__ double variant_value(const VariantCtx<T> &x, const float q[3]) {
  const double p[3] = {(double)q[0], (double)q[1], (double)q[2]};
  Cell8 c;
  make_cell(x.geo, x.unitP2I, x.n, p, c);
  double value = 0.0, total = 0.0;
#pragma unroll
  for (unsigned counter = 0; counter < 8; counter++) {
    double overlap = 1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) overlap *= (counter & (1u << k)) ? c.d[k] : (1.0 - c.d[k]);
    if (overlap != 0.0 && total != 1.0) {
      const T pix = x.s.at((counter & 1) ? c.hi[0] : c.lo[0], (counter & 2) ? c.hi[1] : c.lo[1], (counter & 4) ? c.hi[2] : c.lo[2]);
      value += overlap * (double)pix;
      total += overlap;
    }
  }
  return value;
}

template <class T>
__global__ __launch_bounds__(256) void k_project_variant(const T *__restrict__ vox, Grid g, Geo geo, Params prm, int dirIdentity,
                                                         float *__restrict__ points, u64 nPoints, u64 nGhost,
                                                         Totals *__restrict__ tot, const double *__restrict__ gimg,
                                                         HeldGradient held) {
  const int lane = threadIdx.x & 63;
  const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned passes = 0;
  bool byThr = false, bySteps = false;
  // (a ghost vertex on the ghost slice's bottom plane is NaN and nobody's business here: see k_project)
  if (idx < nPoints && !(idx < nGhost && points[3 * idx] != points[3 * idx])) {
    VariantCtx<T> x;
    x.s = Sampler<T>{vox, g.nx, g.ny, g.nzb, (int)g.zglob0, (int)g.gnz};
    x.geo = geo;
    x.dirIdentity = dirIdentity != 0;
    x.unitP2I = true;
    for (int i = 0; i < 9; i++) x.unitP2I = x.unitP2I && (geo.p2i[i] == ((i % 4 == 0) ? 1.0 : 0.0));
    x.n[0] = g.nx; x.n[1] = g.ny; x.n[2] = (int)g.gnz;
    x.iso = (double)iso_as<T>(prm.iso, prm.isoInt);
    x.gimg = gimg;
    x.held = held;
    x.heldUnitP2I = true;
    for (int i = 0; i < 9; i++) x.heldUnitP2I = x.heldUnitP2I && (held.geo.p2i[i] == ((i % 4 == 0) ? 1.0 : 0.0));
    float vertex[3] = {points[3 * idx], points[3 * idx + 1], points[3 * idx + 2]};
    double normal[3];
    if (prm.variant == CUBERILLE_PROJECT_DEFAULT) {
      // txx:439-474 to the letter (the refilling kernel k_project is this branch with the shipped gradient; here for the
      // recursive-Gaussian one)
      double step = prm.step;
      unsigned numberOfSteps = 0;
      for (;;) {
        passes++;
        variant_normal(x, vertex, normal);                                        // txx:451-452
        const double value = variant_value(x, vertex);                            // txx:455
        if (fabs(value - x.iso) < prm.thr) { byThr = true; break; }               // txx:456-460
        const double sign = (value < x.iso) ? +1.0 : -1.0;                        // txx:463
#pragma unroll
        for (int k = 0; k < 3; k++) vertex[k] = (float)((double)vertex[k] + (normal[k] * sign * step));   // txx:464-467
        step *= prm.relax;                                                        // txx:468
        if (numberOfSteps++ > prm.max_steps) { bySteps = true; break; }           // txx:469-473
      }
    } else if (prm.variant == CUBERILLE_PROJECT_ADVANCED) {
      double step = prm.step;
      unsigned numberOfSteps = 0, swaps = 0;
      int previousi = -1;
      for (;;) {
        passes++;
        variant_normal(x, vertex, normal);                                        // txx:356-357
        float temp[2][3];
#pragma unroll
        for (int k = 0; k < 3; k++) {                                             // txx:360-364
          temp[0][k] = (float)((double)vertex[k] + (normal[k] * +1.0 * step));
          temp[1][k] = (float)((double)vertex[k] + (normal[k] * -1.0 * step));
        }
        step *= prm.relax;                                                        // txx:365
        const double d0 = fabs(variant_value(x, temp[0]) - x.iso);                // txx:368-371
        const double d1 = fabs(variant_value(x, temp[1]) - x.iso);
        const int i = (d0 <= d1) ? 0 : 1;                                         // txx:372
        if (previousi < 0) previousi = i;                                         // txx:373
        swaps += (unsigned)(previousi != i);                                      // txx:374
#pragma unroll
        for (int k = 0; k < 3; k++) vertex[k] = i ? temp[1][k] : temp[0][k];      // txx:375
        if ((i ? d1 : d0) < prm.thr) { byThr = true; break; }                     // txx:378-382
        if (numberOfSteps++ > prm.max_steps) { bySteps = true; break; }           // txx:385-389
        if (swaps >= 5) break;                                                    // txx:392-396
      }
    } else {
      // the reference leaves bestVertex unset when no sample beats the initial 10000 (txx:404-405,437): the vertex
      // then stays where it is, like the CPU checker
      float best[3] = {vertex[0], vertex[1], vertex[2]};
      double bestMetric = 10000;
      variant_normal(x, vertex, normal);                                          // txx:408-409
      for (double sign = -1.0; sign <= 1.0; sign += 2.0)                          // txx:412
        for (unsigned j = 1; j < prm.max_steps / 2; j++) {                        // txx:415
          passes++;
          const double d = (double)j / ((double)prm.max_steps / 2.0);             // txx:418
          float temp[3];
#pragma unroll
          for (int k = 0; k < 3; k++)                                             // txx:419-422
            temp[k] = (float)((double)vertex[k] + (normal[k] * sign * prm.step * d));
          const double metric = fabs(variant_value(x, temp) - x.iso);             // txx:424-425
          if (metric < bestMetric) {                                              // txx:430-434
            bestMetric = metric;
#pragma unroll
            for (int k = 0; k < 3; k++) best[k] = temp[k];
          }
        }
#pragma unroll
      for (int k = 0; k < 3; k++) vertex[k] = best[k];                            // txx:437
    }
    points[3 * idx] = vertex[0]; points[3 * idx + 1] = vertex[1]; points[3 * idx + 2] = vertex[2];
    if (idx < nGhost) { passes = 0; byThr = bySteps = false; }   // the statistics count owned vertices only
  }
  unsigned sum = passes;
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) sum += __shfl_down(sum, sft, 64);
  if (lane == 0 && sum) atomicAdd(&tot->iters, (u64)sum);
  const u64 nThr = __ballot(byThr), nSteps = __ballot(bySteps);
  if (lane == 0 && nThr) atomicAdd(&tot->stopThr, (u64)__popcll(nThr));
  if (lane == 0 && nSteps) atomicAdd(&tot->stopSteps, (u64)__popcll(nSteps));
}

// ---------------------------------------------------------------------------------------------
// K4c: USE_GRADIENT_RECURSIVE_GAUSSIAN (h:21,163-164; txx:488-491; compiled out upstream): the gradient image of
// itk::GradientRecursiveGaussianImageFilter, sigma = max spacing, NormalizeAcrossScale on.  The reference holds three
// lines of it, the rest is ITK's recursive (Deriche, fourth order) separable filter: per component the first-derivative
// filter along its axis, then the smoothing filter along the two other axes, float images between the passes, the
// result divided by the spacing.  A line is a recurrence: one lane per line, the causal half forwards (kept in a double
// scratch volume), the anti-causal half backwards, every sum in the order ITK writes it.  Off the default path and
// written for exactness, not speed (lines along x are walked with a stride of a row per lane).  Parity unpinned.
// ---------------------------------------------------------------------------------------------
struct DericheCoef {
  double N0, N1, N2, N3, D1, D2, D3, D4, M1, M2, M3, M4, BN1, BN2, BN3, BN4, BM1, BM2, BM3, BM4;
};

template <class TIn>
__global__ __launch_bounds__(256) void k_rg_pass(const TIn *__restrict__ in, float *__restrict__ out, double *__restrict__ causal,
                                                 DericheCoef c, long long n0, long long n1, long long n2, int axis) {
  // lines along `axis`; the lanes of a wave take lines that are neighbours along the lowest other axis
  const long long n[3] = {n0, n1, n2};
  const long long stride[3] = {1, n0, n0 * n1};
  const int a1 = axis == 0 ? 1 : 0, a2 = axis == 2 ? 1 : 2;
  const long long L = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (L >= n[a1] * n[a2]) return;
  const long long base = (L % n[a1]) * stride[a1] + (L / n[a1]) * stride[a2];
  const long long st = stride[axis], ln = n[axis];
  auto D = [&](long long i) -> double { return (double)in[base + i * st]; };
  const double outV1 = D(0);
  const double d1 = D(1), d2 = D(2), d3 = D(3);
  double s0 = outV1 * c.N0 + outV1 * c.N1 + outV1 * c.N2 + outV1 * c.N3;
  double s1 = d1 * c.N0 + outV1 * c.N1 + outV1 * c.N2 + outV1 * c.N3;
  double s2 = d2 * c.N0 + d1 * c.N1 + outV1 * c.N2 + outV1 * c.N3;
  double s3 = d3 * c.N0 + d2 * c.N1 + d1 * c.N2 + outV1 * c.N3;
  s0 -= outV1 * c.BN1 + outV1 * c.BN2 + outV1 * c.BN3 + outV1 * c.BN4;
  s1 -= s0 * c.D1 + outV1 * c.BN2 + outV1 * c.BN3 + outV1 * c.BN4;
  s2 -= s1 * c.D1 + s0 * c.D2 + outV1 * c.BN3 + outV1 * c.BN4;
  s3 -= s2 * c.D1 + s1 * c.D2 + s0 * c.D3 + outV1 * c.BN4;
  causal[base] = s0; causal[base + st] = s1; causal[base + 2 * st] = s2; causal[base + 3 * st] = s3;
  {
    double dm1 = d3, dm2 = d2, dm3 = d1;                  // data[i-1], [i-2], [i-3]
    double sm1 = s3, sm2 = s2, sm3 = s1, sm4 = s0;        // scratch[i-1] .. [i-4]
    for (long long i = 4; i < ln; i++) {
      const double di = D(i);
      double si = di * c.N0 + dm1 * c.N1 + dm2 * c.N2 + dm3 * c.N3;
      si -= sm1 * c.D1 + sm2 * c.D2 + sm3 * c.D3 + sm4 * c.D4;
      causal[base + i * st] = si;
      dm3 = dm2; dm2 = dm1; dm1 = di;
      sm4 = sm3; sm3 = sm2; sm2 = sm1; sm1 = si;
    }
  }
  const double outV2 = D(ln - 1);
  const double e1 = D(ln - 1), e2 = D(ln - 2), e3 = D(ln - 3);
  double t1 = outV2 * c.M1 + outV2 * c.M2 + outV2 * c.M3 + outV2 * c.M4;           // scratch[ln-1]
  double t2 = e1 * c.M1 + outV2 * c.M2 + outV2 * c.M3 + outV2 * c.M4;              // [ln-2]
  double t3 = e2 * c.M1 + e1 * c.M2 + outV2 * c.M3 + outV2 * c.M4;                 // [ln-3]
  double t4 = e3 * c.M1 + e2 * c.M2 + e1 * c.M3 + outV2 * c.M4;                    // [ln-4]
  t1 -= outV2 * c.BM1 + outV2 * c.BM2 + outV2 * c.BM3 + outV2 * c.BM4;
  t2 -= t1 * c.D1 + outV2 * c.BM2 + outV2 * c.BM3 + outV2 * c.BM4;
  t3 -= t2 * c.D1 + t1 * c.D2 + outV2 * c.BM3 + outV2 * c.BM4;
  t4 -= t3 * c.D1 + t2 * c.D2 + t1 * c.D3 + outV2 * c.BM4;
  auto put = [&](long long i, double anti) {
    const double o = causal[base + i * st] + anti;        // outs[i] = causal; outs[i] += anti-causal
    out[base + i * st] = (float)o;
  };
  put(ln - 1, t1); put(ln - 2, t2); put(ln - 3, t3); put(ln - 4, t4);
  {
    double dp0 = D(ln - 4), dp1 = e3, dp2 = e2, dp3 = e1; // data[i], [i+1], [i+2], [i+3] for i = ln-4
    double sp0 = t4, sp1 = t3, sp2 = t2, sp3 = t1;        // scratch[i] .. [i+3]
    for (long long i = ln - 4; i > 0; i--) {
      double si = dp0 * c.M1 + dp1 * c.M2 + dp2 * c.M3 + dp3 * c.M4;
      si -= sp0 * c.D1 + sp1 * c.D2 + sp2 * c.D3 + sp3 * c.D4;
      put(i - 1, si);
      dp3 = dp2; dp2 = dp1; dp1 = dp0; dp0 = D(i - 1);
      sp3 = sp2; sp2 = sp1; sp1 = sp0; sp0 = si;
    }
  }
}

// component `dim` of the gradient image: the filtered float image over the spacing
__global__ __launch_bounds__(256) void k_rg_store(const float *__restrict__ src, double *__restrict__ grad, int dim, double spacing,
                                                  u64 nvox) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nvox) grad[3 * i + dim] = (double)src[i] / spacing;
}
// ImageBase::TransformLocalVectorToPhysicalVector on every pixel (double coordinates)
__global__ __launch_bounds__(256) void k_rg_direction(double *__restrict__ grad, Geo geo, u64 nvox) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox) return;
  const double l[3] = {grad[3 * i], grad[3 * i + 1], grad[3 * i + 2]};
#pragma unroll
  for (int r = 0; r < 3; r++) {
    double sum = 0.0;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) sum += geo.dir[r * 3 + cc] * l[cc];
    grad[3 * i + r] = sum;
  }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <class F>
static hipError_t by_pixel_type(int pt, F &&fn) {
  switch (pt) {
    case CUBERILLE_PIX_U8:  return fn((const uint8_t *)nullptr);
    case CUBERILLE_PIX_I8:  return fn((const int8_t *)nullptr);
    case CUBERILLE_PIX_U16: return fn((const uint16_t *)nullptr);
    case CUBERILLE_PIX_I16: return fn((const int16_t *)nullptr);
    case CUBERILLE_PIX_U32: return fn((const uint32_t *)nullptr);
    case CUBERILLE_PIX_I32: return fn((const int32_t *)nullptr);
    case CUBERILLE_PIX_F32: return fn((const float *)nullptr);
    case CUBERILLE_PIX_F64: return fn((const double *)nullptr);
    case CUBERILLE_PIX_I64: return fn((const int64_t *)nullptr);
    case CUBERILLE_PIX_U64: return fn((const uint64_t *)nullptr);
  }
  return hipErrorInvalidValue;
}

static inline unsigned grid_for(u64 threads, unsigned block, unsigned cap) {
  u64 b = (threads + block - 1) / block;
  if (b < 1) b = 1;
  if (cap && b > cap) b = cap;
  return (unsigned)b;
}

// log2(words per slice) when the flat classify kernel can set the slice occupancy itself, else -1
static int occupancy_shift(const Grid &g) {
  const u64 wps = (u64)g.ny * g.W;
  if (g.nx % 64 != 0 || (wps & (wps - 1)) != 0) return -1;
  if (((u64)g.nzb * wps) % 16 != 0) return -1;    // a ragged tail goes through k_classify_rows
  int lg = 0;
  while ((1ull << lg) < wps) lg++;
  return lg;
}

// Volumes of a few million voxels (every volume the reference ships) are bound by the NUMBER of launches, not by any
// of them: their ragged rows go through the one-wave-per-word kernel (one launch that also marks the slice occupancy)
// instead of the flat stream's three launches and the occupancy pass.
static bool small_volume(const Grid &g) { return (u64)g.nx * (u64)g.ny * (u64)g.nzb <= (4ull << 20); }

// ragged rows: does the sweep go through the flat stream (k_classify_flat + tail + repack; occupancy derived afterwards)?
static bool ragged_stream_path(const Workspace &w, const Grid &g, size_t elem, const Tuning &tn) {
  return g.nx % 64 != 0 && w.flatBits && ((uintptr_t)w.vox % elem) == 0 && !tn.no_stream_classify && !small_volume(g);
}

// ... or, for buffers of 256 MiB and more, through the span sweep for rows that are not whole words (k_classify_span_rows:
// occupancy on the fly, no scratch stream)?  Also taken by whole-word rows behind a pointer that is not 16-byte aligned.
// A property of the BUFFER, so that every z-range of it takes the same route and launch_occupancy knows which.
static bool ragged_span_path(const Workspace &w, const Grid &g, size_t elem, const Tuning &tn) {
  const bool wholeAligned = g.nx % 64 == 0 && ((uintptr_t)w.vox % 16) == 0;
  return !wholeAligned && tn.classify_variant == 0 && !tn.no_stream_classify && ((uintptr_t)w.vox % elem) == 0 &&
         (u64)g.nx * (u64)g.ny * (u64)g.nzb * (u64)elem >= (256ull << 20) && (u64)g.ny * (u64)g.nzb * (u64)g.W < 0xffff0000ull;
}

// classify slices [z0, z1) of the buffer (a z-range is a contiguous range of voxels and of words)
hipError_t launch_classify(int pixel_type, const Workspace &wAll, const Grid &g, const Params &prm, int z0, int z1,
                           const Tuning &tn, hipStream_t s) {
  if (z1 <= z0) return hipSuccess;
  const double iso = prm.iso;
  const long long isoI = prm.isoInt;
  return by_pixel_type(pixel_type, [&](auto *tag) -> hipError_t {
    typedef typename std::remove_cv<typename std::remove_pointer<decltype(tag)>::type>::type T;
    Workspace w = wAll;
    const T *vox = (const T *)wAll.vox + (size_t)z0 * g.ny * g.nx;
    w.bits = wAll.bits + (size_t)z0 * g.ny * g.W;
    w.sliceOcc = wAll.sliceOcc + z0;
    const u64 nrows = (u64)g.ny * (z1 - z0);
    const bool aligned = ((uintptr_t)vox % 16) == 0;
    if (g.nx % 64 == 0 && aligned) {
      constexpr int VPL = 16 / sizeof(T);
      const u64 nwordsAll = nrows * g.W;
      const int lg = occupancy_shift(g);
      // large ranges: whole spans through the staged write-through kernel (below ~256 MiB the caches absorb the
      // word stores and more, smaller workgroups fill the chip better)
      u64 spanWords = 0;
      const u64 want0 = tn.classify_grid > 0 ? (u64)tn.classify_grid : 512;
      if (tn.classify_variant == 0 && nwordsAll * 64 * sizeof(T) >= (256ull << 20) && nwordsAll / SPAN_WORDS < 2 * want0 &&
          nwordsAll % (SPAN_WORDS / 4) == 0 && !tn.classify_keep_tail) {
        // fewer than two rounds of whole spans: quarter spans, four times the rounds
        // (eighth spans: 0.086 against 0.0825 ms at 512^3; 384 / 640 / 768 / 1024 workgroups of quarter spans: 0.097 / 0.095 / 0.087 / 0.088)
        const u64 nspans = nwordsAll / (SPAN_WORDS / 4);
        const unsigned blocks = (unsigned)(nspans < want0 ? nspans : want0);
        hipLaunchKernelGGL((k_classify_span<T, SPAN_WORDS / 4>), dim3(blocks), dim3(256), 0, s, vox, w.bits, nspans, iso, isoI, w.sliceOcc, lg);
        spanWords = nwordsAll;
      } else
      if (tn.classify_variant == 0 && nwordsAll * 64 * sizeof(T) >= (256ull << 20)) {
        u64 nspans = nwordsAll / SPAN_WORDS;
        const u64 want = tn.classify_grid > 0 ? (u64)tn.classify_grid : 512;   // two workgroups per CU
        // the workgroups take the spans in rounds; a last round that only a part of them has a span for costs a whole span's
        // time all the same (a 134-slice slab of 1024^2 float32: 536 spans, 0.135 ms where 512 take 0.065): below two
        // thirds of a round the rest goes to the plain sweep behind this launch, which runs at four fifths of the rate
        if (nspans > want && (nspans % want) * 3 < want * 2 && !tn.classify_keep_tail) nspans -= nspans % want;
        const unsigned blocks = (unsigned)(nspans < want ? nspans : want);
        hipLaunchKernelGGL((k_classify_span<T>), dim3(blocks), dim3(256), 0, s, vox, w.bits, nspans, iso, isoI, w.sliceOcc, lg);
        spanWords = nspans * SPAN_WORDS;
      }
      const u64 restWords = nwordsAll - spanWords;
      const u64 nchunks = restWords / VPL;        // whole 1 KiB chunks; the < VPL words left go below
      if (nchunks) {
        // 256 CUs x 8 blocks of 256 threads; grid-stride over the rest; 8 KiB per wave trip, nontemporal
        const unsigned blocks = grid_for((nchunks + 7) / 8 * 64, 256, tn.classify_grid > 0 ? tn.classify_grid : 2048);
        hipLaunchKernelGGL((k_classify_flat<T, 8, true>), dim3(blocks), dim3(256), 0, s, vox + spanWords * 64, w.bits + spanWords,
                           nchunks, iso, isoI, w.sliceOcc, lg, spanWords);
      }
      if (spanWords + nchunks * VPL < nwordsAll)
        hipLaunchKernelGGL((k_classify_rows<T>), dim3(1), dim3(256), 0, s, vox, w.bits, g.nx, g.W, spanWords + nchunks * VPL, nrows,
                           (u64)g.ny, iso, isoI, w.sliceOcc);
    } else if (ragged_span_path(wAll, g, sizeof(T), tn)) {
      const u64 nwordsAll = nrows * g.W;
      const u64 nspans = (nwordsAll + SPAN_WORDS - 1) / SPAN_WORDS;
      const u64 want = tn.classify_grid > 0 ? (u64)tn.classify_grid : 512;     // two workgroups per CU
      // (the workgroups take the spans in rounds: as many workgroups as fill every round -- 3907 spans of a 1000^3 volume
      //  are 8 rounds of 489 rather than 7 of 512 and one of 323, whose time is a whole round's)
      const u64 rounds = (nspans + want - 1) / want;
      const unsigned blocks = (unsigned)((nspans + rounds - 1) / rounds);
      hipLaunchKernelGGL((k_classify_span_rows<T>), dim3(blocks), dim3(256), 0, s, vox, w.bits, nspans, nwordsAll, g.nx, g.W, (u32)g.ny,
                         iso, isoI, w.sliceOcc);
    } else if (ragged_stream_path(wAll, g, sizeof(T), tn)) {
      // ragged rows: flat stream of aligned 16-byte vectors (the first and last vector may reach up to 15 bytes
      // outside the range -- same 16-byte granule as valid voxels, so the loads cannot fault, and those bits
      // are never used), then cut into rows.  Each z-range uses its own part of the scratch.
      constexpr int VPL = 16 / sizeof(T);
      const uintptr_t addr = (uintptr_t)vox, aaddr = addr & ~(uintptr_t)15;
      const u64 skew = (u64)(addr - aaddr) / sizeof(T);
      const T *abase = (const T *)aaddr;
      u64 *flat = wAll.flatBits + (size_t)z0 * g.ny * g.W;
      const u64 nvec = (skew + nrows * (u64)g.nx + VPL - 1) / VPL;
      const u64 nchunks = nvec / 64;
      if (nchunks) {
        // (the 32-KiB-in-flight geometry of the span kernel does not carry over: this loop waits for all its loads before
        //  it computes, 0.74 ms at U4 x 1024 workgroups vs 0.745 at U8 x 2048 on 1000^3 f32)
        const unsigned blocks = grid_for((nchunks + 7) / 8 * 64, 256, 2048);
        hipLaunchKernelGGL((k_classify_flat<T, 8, true>), dim3(blocks), dim3(256), 0, s, abase, flat, nchunks, iso, isoI,
                           (u32 *)nullptr, -1, (u64)0);
      }
      if (nvec % 64) hipLaunchKernelGGL((k_classify_tail<T>), dim3(1), dim3(64), 0, s, abase, flat, nchunks * 64, nvec, iso, isoI);
      hipLaunchKernelGGL(k_repack_rows, dim3(grid_for(nrows * g.W, 256, 0)), dim3(256), 0, s, flat, w.bits, g.nx, g.W, nrows, skew);
    } else {
      const u64 total = nrows * g.W;
      const unsigned blocks = grid_for(total * 64, 256, 8192);
      hipLaunchKernelGGL((k_classify_rows<T>), dim3(blocks), dim3(256), 0, s, vox, w.bits, g.nx, g.W, (u64)0, nrows,
                         (u64)g.ny, iso, isoI, w.sliceOcc);
    }
    return hipGetLastError();
  });
}

// per-slice occupancy from the packed bits where the sweep could not set it on the fly
// (the one-wave-per-word kernel marks the occupancy itself: rows that are not whole words off the stream path, and
//  whole-word rows behind a pointer that is not 16-byte aligned)
hipError_t launch_occupancy(int pixel_type, const Workspace &w, const Grid &g, const Tuning &tn, hipStream_t s) {
  size_t elem = 1;
  (void)by_pixel_type(pixel_type, [&](auto *tag) -> hipError_t { elem = sizeof(*tag); return hipSuccess; });
  const bool aligned = g.nx % 64 == 0 && ((uintptr_t)w.vox % 16) == 0;
  if (ragged_span_path(w, g, elem, tn)) return hipSuccess;           // (that sweep marks the occupancy itself)
  if ((aligned && occupancy_shift(g) < 0) || ragged_stream_path(w, g, elem, tn))
    hipLaunchKernelGGL(k_occupancy, dim3(g.nzb), dim3(256), 0, s, w.bits, (size_t)g.ny * g.W, w.sliceOcc);
  return hipGetLastError();
}

// ... and for slices [z0, z1) whose bits did not come out of this context's sweep (the planes a neighbour rank sent:
// cuberille_step_count)
hipError_t launch_occupancy_range(const Workspace &w, const Grid &g, int z0, int z1, hipStream_t s) {
  if (z1 <= z0) return hipSuccess;
  const size_t wps = (size_t)g.ny * g.W;
  hipLaunchKernelGGL(k_occupancy, dim3(z1 - z0), dim3(256), 0, s, w.bits + (size_t)z0 * wps, wps, w.sliceOcc + z0);
  return hipGetLastError();
}

// A context's FIRST extraction has no previous one to take the form of its count from: a sample of the bit volume -- every
// stride-th word of the counted range, a few thousand wave loads -- says how many words hold both inside and outside voxels
// (every such word has faces along x and creates vertices).  out: mixed words | sampled words << 32, added to.
__global__ __launch_bounds__(256) void k_density_probe(const u64 *__restrict__ bits, size_t nwords, size_t stride, int nx, int W,
                                                       u64 *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t * stride;
  bool mixed = false, live = i < nwords;
  if (live) {
    const u64 w = bits[i];
    // (the last word of a row that ends inside it is full at fewer than 64 bits)
    const int k = (int)(i % (size_t)W), n = nx - k * 64;
    const u64 full = n < 64 ? lowmask(n) : ~0ull;
    mixed = w != 0ull && w != full;
  }
  const u64 m = __ballot(mixed), l = __ballot(live);
  if ((threadIdx.x & 63) == 0 && l) atomicAdd((unsigned long long *)out, (unsigned long long)__popcll(m) | ((unsigned long long)__popcll(l) << 32));
}

hipError_t launch_density_probe(const Workspace &w, const Grid &g, size_t nwords, u64 *out, hipStream_t s) {
  const size_t samples = 1u << 18;
  const size_t stride = nwords / samples > 0 ? (nwords / samples) | 1 : 1;      // (odd: walks through the word columns)
  const size_t n = (nwords + stride - 1) / stride;
  const u64 *counted = w.bits + (size_t)g.cz0 * g.ny * g.W;
  hipLaunchKernelGGL(k_density_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, counted, nwords, stride, g.nx, g.W, out);
  return hipGetLastError();
}

hipError_t launch_count(const Workspace &w, const Grid &g, size_t nwords, int q1, const Gate &gate, int tiled, int noFold, hipStream_t s) {
  const unsigned blocks = (unsigned)((nwords + COUNT_WB - 1) / COUNT_WB);
  u32 *vq = nwords < 0xffffffffULL ? w.vqueue : nullptr;
  const size_t sliceWords = (size_t)g.ny * g.W;
  // (tiled 3, or 32 + one of the values below: the dense form -- faces and corner logic in one phase, per lattice corner --
  //  where a row is a power of two of words, else the two-phase tile)
  const bool fused = (tiled == 3 || tiled >= 32) && g.wShift >= 0;
  // the block scan inside the count launch where every count block is resident at once (block_scan_tail)
  const bool foldable = blocks <= FOLD_MAX_BLOCKS && !noFold;
  unsigned fold = 0;
  if (tiled == 3) tiled = 1;
  else if (tiled >= 32) tiled -= 32;
  if (tiled && g.W <= TILE_WMAX) {
    // (slices that are whole count blocks: a workgroup walks up a column of COUNT_ZRUN blocks and re-uses two of its three planes)
    // (tiled 2: one block per workgroup; >= 4: that run; the pipelined dense form takes columns twice as long where those
    //  still fill the chip: its first block is the one whose planes nobody prefetched -- 2048^3 noise 1.20 -> 1.18 ms)
    const bool longColumns = fused && tiled == 1 && (u64)(sliceWords / COUNT_WB) * (u64)((g.oz1 - g.cz0 + 2 * COUNT_ZRUN - 1) / (2 * COUNT_ZRUN)) >= 1024;
    const int want = tiled == 1 ? (longColumns ? 2 * COUNT_ZRUN : COUNT_ZRUN) : tiled >= 4 ? tiled : 0;
    // (... where the columns are still enough workgroups to fill the chip: a 129-slice slab of 1024^2 gives 136 columns of 8,
    //  0.107 ms against 0.05 one block per workgroup)
    const bool columnsFill = want && (u64)(sliceWords / COUNT_WB) * (u64)((g.oz1 - g.cz0 + want - 1) / want) >= 1024;
    const int zrun = want && sliceWords % COUNT_WB == 0 && g.oz1 - g.cz0 >= 2 * want && (columnsFill || tiled >= 4) ? want : 0;
    const unsigned grid = zrun ? (unsigned)(sliceWords / COUNT_WB) * (unsigned)((g.oz1 - g.cz0 + zrun - 1) / zrun) : blocks;
    if (fused)
      hipLaunchKernelGGL((k_count_dense<0>), dim3(grid), dim3(512), 0, s, w.bits, w.sliceOcc, g, nwords, q1, w.prefix, w.segPre, w.blockTot, vq,
                         w.totals, zrun);
    else
      hipLaunchKernelGGL((k_count<0, true, 512>), dim3(grid), dim3(512), 0, s, w.bits, w.sliceOcc, g, nwords, q1, w.prefix, w.segPre,
                         w.blockTot, vq, w.totals, zrun, w.blockBase, gate, 0);
  } else if (blocks <= 64)
    // (a handful of blocks -- every volume the reference ships: the kernel's time is a block's latency, two trips through its
    //  loops instead of eight)
    if (foldable)
      hipLaunchKernelGGL((k_count<0, false, 1024, true>), dim3(blocks), dim3(1024), 0, s, w.bits, w.sliceOcc, g, nwords, q1, w.prefix, w.segPre,
                         w.blockTot, vq, w.totals, 0, w.blockBase, gate, (int)(fold = blocks));
    else
      hipLaunchKernelGGL((k_count<0, false, 1024>), dim3(blocks), dim3(1024), 0, s, w.bits, w.sliceOcc, g, nwords, q1, w.prefix, w.segPre,
                         w.blockTot, vq, w.totals, 0, w.blockBase, gate, 0);
  else
    hipLaunchKernelGGL((k_count<0, false, 256>), dim3(blocks), dim3(256), 0, s, w.bits, w.sliceOcc, g, nwords, q1, w.prefix, w.segPre,
                       w.blockTot, vq, w.totals, 0, w.blockBase, gate, 0);
  if (fold) return hipGetLastError();              // (the last count block did the scan)
  const size_t g0 = (size_t)(g.oz0 - g.cz0) * g.ny * g.W;
  const unsigned chunks = blocks > SCAN_CHUNK ? (blocks + SCAN_CHUNK - 1) / SCAN_CHUNK : 1;
  if (chunks > 1) hipLaunchKernelGGL(k_block_partial, dim3(chunks), dim3(1024), 0, s, w.blockTot, blocks);
  hipLaunchKernelGGL(k_block_scan, dim3(chunks), dim3(1024), 0, s, w.blockTot, w.blockBase, blocks, g0, w.totals, gate, w.sliceOcc, g.cz0, g.oz0, g.oz1, g.zglob0,
                     part_of_a_volume(g) ? 1 : 0);
  return hipGetLastError();
}

static EmitArgs emit_args(const Workspace &w, const Grid &g, int q1, u64 pointOffset) {
  EmitArgs a;
  const size_t nwords = (size_t)(g.oz1 - g.cz0) * g.ny * g.W;
  a.bits = w.bits; a.occ = w.sliceOcc; a.q1 = q1; a.prefix = w.prefix;
  a.segPre = w.segPre; a.blockBase = w.blockBase; a.nblk = (nwords + COUNT_WB - 1) / COUNT_WB;
  a.tot = w.totals;
  a.points = w.points;
  a.cells = w.cells;
  a.pointOffset = pointOffset;
  a.cmap = w.cmap;
  a.headV = w.headV; a.headQ = w.headQ;
  a.extIds = nullptr;
  a.rows = nullptr; a.nRanks = 0; a.rank = 0; a.dyn = 0;
  return a;
}

hipError_t launch_heads(const Workspace &w, const Grid &g, u64 totV, u64 totQ, int dyn, hipStream_t s) {
  if (!w.headQ) return hipSuccess;
  const size_t nwords = (size_t)(g.oz1 - g.cz0) * g.ny * g.W;
  EmitArgs a = emit_args(w, g, 0, 0);
  a.dyn = dyn;
  const u64 nHQ = (totQ + 63) / 64, nHV = w.headV ? (totV + 63) / 64 : 0;
  if (nHQ) hipLaunchKernelGGL((k_heads_search<16>), dim3(grid_for(nHQ, 256, 0)), dim3(256), 0, s, a, nwords, nHQ, w.headQ);
  if (nHV) hipLaunchKernelGGL((k_heads_search<0>), dim3(grid_for(nHV, 256, 0)), dim3(256), 0, s, a, nwords, nHV, w.headV);
  return hipGetLastError();
}

// lanes per vertex word of the point pass (k_emit_points_dense<SPLIT>): its first phase walks a word's voxels serially, a
// word on a flat face holds up to 128 vertices, and a short queue leaves wave slots empty -- more lanes per word there
// (profiles/r4_size_sweep.log: Marschner-Lobb 256^3, 154 k vertex words, 0.089 ms with one lane per word, 0.036 with four;
//  512^3, 630 k words, 0.096 against 0.068 with two; from 768^3 on one lane per word wins, as on every noise field above 256^3)
static int points_split_for(u32 nVertexWords) {
  // (a launch sized blindly passes its cover, a quarter above the previous extraction's queue)
  return nVertexWords <= 32768u ? 8 : nVertexWords <= 200000u ? 4 : nVertexWords <= 900000u ? 2 : 1;
}

hipError_t launch_emit_points(const Workspace &w, const Grid &g, const Geo &geo, int q1, u64 nV, u32 nVertexWords,
                              const Tuning &tn, int dyn, hipStream_t s) {
  if (!nV) return hipSuccess;
  const size_t nwords = (size_t)(g.oz1 - g.cz0) * g.ny * g.W;
  EmitArgs a = emit_args(w, g, q1, 0);
  a.dyn = dyn;                                   // (only the queue form below is ever launched blindly)
  if (w.vqueue && nwords < 0xffffffffULL && tn.points_variant == 3) {
    const int split = tn.points_split > 0 ? tn.points_split : (tn.points_no_split ? 1 : points_split_for(nVertexWords));
#define CUBERILLE_LAUNCH_POINTS(SPLIT)                                                                                      \
    hipLaunchKernelGGL(k_emit_points_dense<SPLIT>, dim3(grid_for((u64)nVertexWords * SPLIT, 256, 0)), dim3(256), 0, s, a, g, geo, \
                       w.vqueue, nVertexWords)
    if (split >= 8) CUBERILLE_LAUNCH_POINTS(8);
    else if (split >= 4) CUBERILLE_LAUNCH_POINTS(4);
    else if (split >= 2) CUBERILLE_LAUNCH_POINTS(2);
    else CUBERILLE_LAUNCH_POINTS(1);
#undef CUBERILLE_LAUNCH_POINTS
  } else
    hipLaunchKernelGGL(k_emit_points_wave, dim3(grid_for(nV, 256, 0)), dim3(256), 0, s, a, g, geo, nwords, nV);
  return hipGetLastError();
}

// Quirk Q1 across a slab boundary, the serving side: for every lattice corner (cx, cy) on the plane above local slice
// z (this rank's highest occupied slice; everything above it in its range is empty) the global id and the final
// position of the vertex the reference would find under that (x, y) key -- it exists when an inside voxel of slice z
// touches the corner -- or ~0 where there is none.  The rank above re-uses these for its first occupied slice.
__global__ __launch_bounds__(256) void k_alias_plane(EmitArgs a, Grid g, int z, u64 *__restrict__ idsOut, float *__restrict__ ptsOut) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 n = (u64)(g.nx + 1) * (u64)(g.ny + 1);
  if (j >= n) return;
  const int cx = (int)(j % (u64)(g.nx + 1)), cy = (int)(j / (u64)(g.nx + 1));
  bool hit = false;
  for (int ee = 0; ee < 4; ee++) {
    const int vx = cx - (ee & 1), vy = cy - (ee >> 1);
    if (vx >= 0 && vx < g.nx && vy >= 0 && vy < g.ny && getbit(a.bits, g, vx, vy, z)) hit = true;
  }
  u64 id = ~0ull;
  float p[3] = {0.f, 0.f, 0.f};
  if (hit) {
    const u64 lid = a.cmap ? (u64)a.cmap[corner_map_index(g, cx, cy, z + 1)] : corner_id_generic(a, g, cx, cy, z + 1);
    id = lid - a.tot->V0 + a.pointOffset;
    p[0] = a.points[3 * lid]; p[1] = a.points[3 * lid + 1]; p[2] = a.points[3 * lid + 2];
  }
  idsOut[j] = id;
  ptsOut[3 * j] = p[0]; ptsOut[3 * j + 1] = p[1]; ptsOut[3 * j + 2] = p[2];
}

// Vertices created / quads emitted before the first word of every counted slice (entry i: slice cz0 + i; the last
// entry: the totals) -- what a driver that cuts the volume into slabs of equal WORK wants to know about the last count.
__global__ __launch_bounds__(256) void k_slice_prefix(EmitArgs a, Grid g, size_t nwords, int nSlices, u64 *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nSlices) return;
  const size_t gi = (size_t)i * g.ny * g.W;
  u64 V, Q;
  if (i < nSlices && gi < nwords) {
    const u32 pw = a.prefix[gi];
    V = seg_base<0>(a, gi) + (pw & 0xffffu);
    Q = seg_base<16>(a, gi) + (pw >> 16);
  } else {
    V = a.tot->totV;
    Q = a.tot->totQ;
  }
  out[2 * i] = V;
  out[2 * i + 1] = Q;
}

hipError_t launch_slice_prefix(const Workspace &w, const Grid &g, u64 *out, hipStream_t s) {
  const size_t nwords = (size_t)(g.oz1 - g.cz0) * g.ny * g.W;
  const EmitArgs a = emit_args(w, g, 0, 0);
  const int nSlices = g.oz1 - g.cz0;
  hipLaunchKernelGGL(k_slice_prefix, dim3(grid_for((u64)nSlices + 1, 256, 0)), dim3(256), 0, s, a, g, nwords, nSlices, out);
  return hipGetLastError();
}

hipError_t launch_alias_plane(const Workspace &w, const Grid &g, int zLocal, u64 pointOffset, u64 *idsOut, float *ptsOut,
                              hipStream_t s) {
  const EmitArgs a = emit_args(w, g, 1, pointOffset);
  const u64 n = (u64)(g.nx + 1) * (u64)(g.ny + 1);
  hipLaunchKernelGGL(k_alias_plane, dim3(grid_for(n, 256, 0)), dim3(256), 0, s, a, g, zLocal, idsOut, ptsOut);
  return hipGetLastError();
}

hipError_t launch_emit_cells(const Workspace &w, const Grid &g, int triangles, int q1, u64 pointOffset, u64 nQ,
                             const u64 *extIds, const Totals *rows, int nRanks, int rank, int dyn, hipStream_t s) {
  if (!nQ) return hipSuccess;
  const size_t nwords = (size_t)(g.oz1 - g.cz0) * g.ny * g.W;
  EmitArgs a = emit_args(w, g, q1, pointOffset);
  a.extIds = extIds;
  a.rows = rows; a.nRanks = nRanks; a.rank = rank; a.dyn = dyn;
  const dim3 grid(grid_for(nQ, 256, 0)), block(256);
  if (triangles && a.cmap) hipLaunchKernelGGL((k_emit_cells<true, true>), grid, block, 0, s, a, g, nwords, nQ);
  else if (triangles) hipLaunchKernelGGL((k_emit_cells<true, false>), grid, block, 0, s, a, g, nwords, nQ);
  else if (a.cmap) hipLaunchKernelGGL((k_emit_cells<false, true>), grid, block, 0, s, a, g, nwords, nQ);
  else hipLaunchKernelGGL((k_emit_cells<false, false>), grid, block, 0, s, a, g, nwords, nQ);
  return hipGetLastError();
}

// The gradient image of txx:478-498 materialised (I6 at every pixel) -- only for a context asked to hold it across
// extractions (quirk Q3, cuberille_hold_gradient); the walk of an ordinary extraction evaluates the same taps on the fly.
template <class T>
__global__ __launch_bounds__(256) void k_gradient_image(const T *__restrict__ vox, Grid g, Geo geo, int dirIdentity,
                                                        float *__restrict__ out, size_t nvox) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox) return;
  const int x = (int)(i % (size_t)g.nx), y = (int)((i / (size_t)g.nx) % (size_t)g.ny), z = (int)(i / ((size_t)g.nx * g.ny));
  const Sampler<T> s{vox, g.nx, g.ny, g.nzb, 0, g.nzb};
  float gr[3];
  gradient_at(s, geo, dirIdentity != 0, x, y, z, (float)vox[i], gr);
  out[3 * i] = gr[0]; out[3 * i + 1] = gr[1]; out[3 * i + 2] = gr[2];
}

hipError_t launch_gradient_image(int pixel_type, const Workspace &w, const Grid &g, const Geo &geo, float *out, hipStream_t s) {
  const size_t nvox = (size_t)g.nx * g.ny * g.nzb;
  if (nvox == 0) return hipSuccess;
  if ((nvox + 255) / 256 > 0x7fffffffull) return hipErrorInvalidValue;
  int dirIdentity = 1;
  for (int i = 0; i < 9; i++) if (geo.dir[i] != ((i % 4 == 0) ? 1.0 : 0.0)) dirIdentity = 0;
  return by_pixel_type(pixel_type, [&](auto *tag) -> hipError_t {
    typedef typename std::remove_cv<typename std::remove_pointer<decltype(tag)>::type>::type T;
    hipLaunchKernelGGL((k_gradient_image<T>), dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, s, (const T *)w.vox, g, geo,
                       dirIdentity, out, nvox);
    return hipGetLastError();
  });
}

// mode: 0 plain, 1 THIN_HALO escape detection (w.escList), 2 the escaped vertices again (nPoints = their number; `g`
// and w.vox describe the deeper buffer).  dyn: sizes from the device totals (cuberille_step_begin); nPoints is then
// only what the launch is sized for.
hipError_t launch_project(int pixel_type, const Workspace &w, const Grid &g, const Geo &geo, const Params &p, u64 nPoints,
                          u64 nGhost, const Tuning &tn, int mode, int dyn, hipStream_t s) {
  if (nPoints == 0) return hipSuccess;
  int dirIdentity = 1;
  for (int i = 0; i < 9; i++) if (geo.dir[i] != ((i % 4 == 0) ? 1.0 : 0.0)) dirIdentity = 0;
  if (p.variant != CUBERILLE_PROJECT_DEFAULT || p.gradVariant != 0 || (w.held && w.held->img))
    return by_pixel_type(pixel_type, [&](auto *tag) -> hipError_t {
      typedef typename std::remove_cv<typename std::remove_pointer<decltype(tag)>::type>::type T;
      HeldGradient held{};
      if (w.held && w.held->img) held = *w.held;
      hipLaunchKernelGGL((k_project_variant<T>), dim3(grid_for(nPoints, 256, 0)), dim3(256), 0, s, (const T *)w.vox, g, geo, p,
                         dirIdentity, w.points, nPoints, nGhost, w.totals, p.gradVariant ? w.gradImg : nullptr, held);
      return hipGetLastError();
    });
  // batches of 128 vertices dealt round-robin to 16384 waves (same-box A/B at 1024^3 M-L: 1.54 ms vs 1.68 ms
  // for one contiguous chunk of 256 per wave; u8 noise prefers contiguous, 2.23 vs 2.35 ms; earlier runs:
  // chunk per wave: 256 -> 1.64 ms, 906 -> 1.93 ms, 3648 -> 2.40 ms; 64 without refill 2.76 ms)
  // (a launch that leaves wave slots empty -- every volume the reference ships -- deals 64 per wave: one vertex per lane, no
  //  refill, the kernel ends with its slowest walk instead of with a wave's second helping; nucleon 0.212 -> 0.163 ms wall)
  // (... and up to a few batches of 128 per wave -- a slab of a multi-GPU run: 4.4 M vertices at 1024^3 -- batches of 64 spread
  //  the tails of the waves' sequences better: 0.83 -> 0.74 ms there, 0.38 -> 0.33 at 1.6 M; at 11.1 M 128 wins by 2 %)
  //  -- but only once batches of 64 fill the grid: between the two, 0.8 M vertices of a 512^3 sphere, 6170 waves that refill
  //  from their 128 beat 12 340 that cannot, 0.172 against 0.187 ms)
  // (round 5: where walks are SHORT -- waves that refill only when empty, a dense field -- the walk is bound by its gathers, and
  //  what those cost follows how far apart in the vertex list the waves resident at one time work: counters of the same
  //  field in rows of 512 and of 2048 voxels, profiles/r5_walk_row_width_counters.txt -- no translation misses; L1 hit rate 80 ->
  //  68 %, L2 37 -> 24 %, 1.7x the requests beyond L2.  A wave of a 16 384-wave launch works through batches that lie
  //  NW * chunk vertices apart, and the resident waves are at different places of their sequences: the fewer batches a wave
  //  takes the closer together the chip works -- down to the point where a wave's start costs more than it walks.  Measured
  //  (profiles/microbench/r5_walk_waves.log): the best launch deals batches of 64 to one wave per ~450 vertices -- 13 M
  //  vertices (768^3 noise) 24 576-32 768 waves, 0.650 -> 0.597 ms; 31 M (1024^3) 65 536, 1.52 -> 1.37; 249 M (2048^3) 524 288,
  //  17.0 -> 12.9 ms (18.4 in round 4) -- and loses again with half as many vertices per wave.  Long walks keep 16 384 waves.)
  //  LONG walks (the refilling waves, Marschner-Lobb): the same sweeps give one wave per ~680 vertices -- which IS the 16 384 waves
  //  of rounds 1-4 at the headline's 11.1 M vertices (12 288 / 16 384 / 20 480 / 24 576 waves: 1.157 / 1.144 / 1.193 / 1.170 ms) --
  //  512^3 (2.8 M vertices) 4096 waves 0.388 against 0.439 ms, 768^3 12 288 waves 0.717 against 0.739, 2048^3 (44.8 M)
  //  65 536 waves 4.14 against 4.30.  Below 4096 waves' worth of vertices the launch shapes of rounds 3-4 stand.)
  const bool shortWalks = tn.proj_short >= 0 ? tn.proj_short != 0 : tn.proj_refill >= 64;
  const u64 perWave = shortWalks ? 448 : 680;
  constexpr u64 WALK_RESIDENT = 3072;       // the unit the sweeps below settled on for the wave count (the kernel holds 128 VGPRs: 4 waves per SIMD)
  u64 autoWaves = 16384;
  // (... in whole rounds of the 4096 waves the chip holds at 4 per SIMD: 4435 waves for the 3.0 M vertices of a 1000^3 sphere are
  //  one round and a tail of 339 waves that run alone -- 0.571 against 0.496 ms for 16 384)
  if (shortWalks ? nPoints > 16384ull * perWave : nPoints > 16384ull * perWave)
    autoWaves = ((nPoints / perWave + WALK_RESIDENT - 1) / WALK_RESIDENT) * WALK_RESIDENT;
  else if (!shortWalks && nPoints >= WALK_RESIDENT * 64) {
    // (a launch sized blindly is sized for a quarter more than the previous extraction's vertices: the rule goes by the expected number)
    const u64 nExpected = dyn ? nPoints - nPoints / 5 : nPoints;
    // (the kernel with the gradients held as doubles, re-swept, profiles/microbench/r5_walk_three_waves.log: multiples
    //  of 3072 waves of ~300-520 vertices each, batches of 128 from 1.5 M vertices on: 0.8 / 1.8 / 2.8 / 3.2 / 6.3 M vertices
    //  want 3072 / 6144 / 6144 / 6144 / 12 288 waves; the headline's 11.1 M stay with 16 384 -- 21 504 cost it 2.5 %)
    autoWaves = ((nExpected / 520 + WALK_RESIDENT - 1) / WALK_RESIDENT) * WALK_RESIDENT;
    if (autoWaves > 16384) autoWaves = 16384;
  }
  const u64 gridWaves = tn.proj_waves > 0 ? (u64)tn.proj_waves : autoWaves;
  const u64 upTo = tn.proj_chunk64_below > 0 ? (u64)tn.proj_chunk64_below : (shortWalks ? 8000000ull : 1500000ull);
  u64 chunk = tn.proj_chunk > 0 ? (tn.proj_chunk < 64 ? 64 : (u64)tn.proj_chunk)
            : (nPoints <= 64ull * 4096 || (nPoints > 64 * gridWaves && (nPoints < upTo || shortWalks))) ? 64 : 128;
  while (chunk & (chunk - 1)) chunk &= chunk - 1;   // power of two (the kernel shifts instead of dividing)
  u64 nwaves = (nPoints + chunk - 1) / chunk;
  if (gridWaves && nwaves > gridWaves) nwaves = gridWaves;
  return by_pixel_type(pixel_type, [&](auto *tag) -> hipError_t {
    typedef typename std::remove_cv<typename std::remove_pointer<decltype(tag)>::type>::type T;
    const unsigned blocks = grid_for(nwaves * 64, 256, 0);
    // (giving each XCD a contiguous EIGHTH of the vertex list was measured 1.6x slower, round 1: proj_xcd = 1.  Round 5, after
    //  the counters had shown the dense-field walk to be bound by L1 / L2 misses of its gathers: RUNS of G consecutive
    //  workgroups on one XCD (proj_xcd = G > 1) -- 2048^3 noise 12.33 -> 10.9 ms at G = 64 (11.0 at 32 and 128), 1024^3 1.395 ->
    //  1.358 at 16 (1.363 / 1.381 at 32 / 64), and the row-width effect itself goes: 8192 x 512 x 256 77.2 -> 49.5 ps per vertex,
    //  the figure of 512-voxel rows.  Short walks take G = workgroups / 512 within 16 .. 64.  Long walks, interleaved A/B:
    //  G = 32 is worth 4.4 % on 2048^3 Marschner-Lobb (4.25 -> 4.06 ms), 9 % on a 1000^3 sphere (0.513 -> 0.466), 5 % on a 768^3
    //  one, 2 % on 512^3 Marschner-Lobb, and nothing either way on the headline's 1024^3 sheet (1.164-1.184 against 1.165-1.184);
    //  G = 8 .. 16 can lose (1000^3 sphere 0.60-0.62 ms) and G = 64 costs the headline 2 %.  With 32 as the rule the headline's
    //  walk read 1.187-1.191 ms in three stage-timed lines and 1.215 under rocprofv3 against 1.162-1.165 and 1.199 before:
    //  launches of 2304 .. 8191 workgroups (768^3 and 1024^3 Marschner-Lobb: nothing and -1.5 %) keep the plain order.)
    int xcd = tn.proj_xcd;
    if (xcd == 0 && shortWalks && blocks >= 16 * 512) {
      const unsigned gq = blocks / 512;
      xcd = (int)(gq < 16 ? 16 : gq > 64 ? 64 : gq);
    } else if (xcd == 0 && !shortWalks && blocks >= 1024 && (blocks < 2304 || blocks >= 8192)) {
      xcd = 32;
    } else if (xcd < 0) xcd = 0;
    // the kernel's form by the geometry: 2 identity matrices and a region at index 0; 1 identity direction, any spacing (the
    // inverse of a diagonal matrix by cofactors has exact zeros off its diagonal); 0 anything else
    bool diag = dirIdentity != 0 && geo.istart[0] == 0 && geo.istart[1] == 0 && geo.istart[2] == 0 && tn.proj_ident != 0;
    bool unit = diag;
    for (int i = 0; i < 9; i++) {
      if (i % 4 != 0) diag = diag && geo.p2i[i] == 0.0;
      unit = unit && geo.p2i[i] == ((i % 4 == 0) ? 1.0 : 0.0);
    }
    const int geom = unit ? 2 : diag ? 1 : 0;
#define CUBERILLE_LAUNCH_PROJECT(MODE, GEOM)                                                                                 \
    hipLaunchKernelGGL((k_project<T, MODE, GEOM>), dim3(blocks), dim3(256), 0, s, (const T *)w.vox, g, geo, p, dirIdentity,  \
                       w.points, nPoints, nGhost, chunk, tn.proj_refill, xcd, tn.proj_literal, w.totals, w.escList,  \
                       w.escCap, dyn)
#define CUBERILLE_LAUNCH_PROJECT_GEOM(MODE)                                                                                  \
    do { if (geom == 2) CUBERILLE_LAUNCH_PROJECT(MODE, 2); else if (geom == 1) CUBERILLE_LAUNCH_PROJECT(MODE, 1);            \
         else CUBERILLE_LAUNCH_PROJECT(MODE, 0); } while (0)
    if (mode == 1) CUBERILLE_LAUNCH_PROJECT_GEOM(1);
    else if (mode == 2) CUBERILLE_LAUNCH_PROJECT_GEOM(2);
    else CUBERILLE_LAUNCH_PROJECT_GEOM(0);
#undef CUBERILLE_LAUNCH_PROJECT_GEOM
#undef CUBERILLE_LAUNCH_PROJECT
    return hipGetLastError();
  });
}

// The recursive-Gaussian gradient image of the whole volume into w.gradImg (3 doubles per voxel), through the two float
// volumes w.rgA / w.rgB and the double scratch volume w.rgScratch.  coef[dim][0]: the derivative filter along dim;
// coef[ax][1]: the smoothing filter along ax (host: cuberille_api.hip, deriche_setup).
hipError_t launch_recursive_gaussian(int pixel_type, const Workspace &w, const Grid &g, const Geo &geo, const double coef[3][2][20],
                                     hipStream_t s) {
  const long long n[3] = {g.nx, g.ny, g.nzb};
  const u64 nvox = (u64)g.nx * g.ny * g.nzb;
  auto lines = [&](int axis) -> unsigned { return grid_for((u64)(nvox / (u64)n[axis]), 256, 0); };
  auto coefOf = [&](int ax, int which) { DericheCoef c; std::memcpy(&c, coef[ax][which], sizeof(c)); return c; };
  for (int dim = 0; dim < 3; dim++) {
    hipError_t e = by_pixel_type(pixel_type, [&](auto *tag) -> hipError_t {
      typedef typename std::remove_cv<typename std::remove_pointer<decltype(tag)>::type>::type T;
      hipLaunchKernelGGL((k_rg_pass<T>), dim3(lines(dim)), dim3(256), 0, s, (const T *)w.vox, w.rgA, w.rgScratch, coefOf(dim, 0),
                         n[0], n[1], n[2], dim);
      return hipGetLastError();
    });
    if (e != hipSuccess) return e;
    float *src = w.rgA, *dst = w.rgB;
    for (int ax = 0; ax < 3; ax++) {
      if (ax == dim) continue;
      hipLaunchKernelGGL((k_rg_pass<float>), dim3(lines(ax)), dim3(256), 0, s, (const float *)src, dst, w.rgScratch, coefOf(ax, 1),
                         n[0], n[1], n[2], ax);
      std::swap(src, dst);
    }
    hipLaunchKernelGGL(k_rg_store, dim3(grid_for(nvox, 256, 0)), dim3(256), 0, s, (const float *)src, w.gradImg, dim, geo.spacing[dim], nvox);
  }
  hipLaunchKernelGGL(k_rg_direction, dim3(grid_for(nvox, 256, 0)), dim3(256), 0, s, w.gradImg, geo, nvox);
  return hipGetLastError();
}

}  // namespace cuberille
