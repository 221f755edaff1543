// Development harness (not part of the product library): times variants of the threshold + bit-pack sweep
// (classify) on one MI355X against a plain read-only stream of the same bytes, same box, same process.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o classify_sweep classify_sweep.hip
//   ./classify_sweep [n=1024] [reps=10]
// The product kernels are pulled in from the library source, so "product" below is exactly what ships.
#include "../../midas-journal-740_amd/csrc/cuberille_kernels.hip"

#include <cstdio>
#include <vector>
#include <string>
#include <algorithm>

using namespace cuberille;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// Marschner-Lobb field inside a one-voxel shell of 0 (bench workload; libm on the device is fine here, the harness
// only compares variants with each other)
__global__ void k_fill_ml(float *vox, int n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n * n * n;
  if (i >= total) return;
  const int x = (int)(i % n), y = (int)((i / n) % n), z = (int)(i / ((size_t)n * n));
  float v = 0.f;
  if (x >= 1 && x <= n - 2 && y >= 1 && y <= n - 2 && z >= 1 && z <= n - 2) {
    const double m = n - 2;
    const double X = -1.0 + (2.0 * (x - 1.0) + 1.0) / m, Y = -1.0 + (2.0 * (y - 1.0) + 1.0) / m, Z = -1.0 + (2.0 * (z - 1.0) + 1.0) / m;
    const double r = sqrt(X * X + Y * Y);
    const double pr = cos(2.0 * M_PI * 6.0 * cos(M_PI * r / 2.0));
    v = (float)(((1.0 - sin(M_PI * Z / 2.0)) + 0.25 * (1.0 + pr)) / (2.0 * 1.25));
  }
  vox[i] = v;
}

// ---------------------------------------------------------------------------------------------------------
// read-only reference stream: what this box gives a kernel that only reads the volume
// ---------------------------------------------------------------------------------------------------------
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read_only(const uint4 *__restrict__ src, u64 nchunks, u32 *__restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u64 nwaves = (u64)gridDim.x * (blockDim.x >> 6);
  u32 acc = 0;
  for (u64 c = wave * U; c + U <= nchunks; c += nwaves * U) {
    uint4 r[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint4 *p = src + (c + u) * 64 + lane;
      if (NT) { r[u].x = __builtin_nontemporal_load(&p->x); r[u].y = __builtin_nontemporal_load(&p->y);
                r[u].z = __builtin_nontemporal_load(&p->z); r[u].w = __builtin_nontemporal_load(&p->w); }
      else r[u] = *p;
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc |= r[u].x | r[u].y | r[u].z | r[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;   // never true for this data; keeps the loads alive
}

// ---------------------------------------------------------------------------------------------------------
// v2: wave-uniform (scalar) chunk loop without per-load guards, DPP OR inside the 16-lane rows instead of
// ds_bpermute shuffles; the word ends up in the LAST lane of its lane group.
// STORE 0: each word's lane stores 8 B itself (U store instructions of 4 lanes).
// STORE 1: the U*VPL words of a trip are staged through LDS and written by consecutive lanes (one store).
// PIPE: prefetch the next trip's loads before processing the current one.
// ---------------------------------------------------------------------------------------------------------
template <class T, bool NT>
__device__ __forceinline__ void load16(Vec16<T> &r, const T *p) {
  const uint4 *src = reinterpret_cast<const uint4 *>(p);
  if (NT) { r.raw.x = __builtin_nontemporal_load(&src->x); r.raw.y = __builtin_nontemporal_load(&src->y);
            r.raw.z = __builtin_nontemporal_load(&src->z); r.raw.w = __builtin_nontemporal_load(&src->w); }
  else r.raw = *src;
}

template <class T, int U, bool NT, int STORE, bool PIPE, int BLOCK, int MODE = 0>
__global__ __launch_bounds__(BLOCK) void k_classify_v2(const T *__restrict__ vox, u64 *__restrict__ bits, u64 nchunks, double isoD,
                                                       u32 *__restrict__ sliceOcc, int lgWordsPerSlice) {
  constexpr int VPL = 16 / sizeof(T);
  constexpr int LPW = 64 / VPL;
  constexpr int WAVES = BLOCK / 64;
  __shared__ u64 stage[STORE ? WAVES : 1][STORE ? U * VPL : 1];
  const T iso = (T)isoD;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u64 wave = (u64)blockIdx.x * WAVES + wib;
  const u64 nwaves = (u64)gridDim.x * WAVES;
  const int sub = lane % LPW;
  const bool last = sub == LPW - 1;
  const u64 nfull = nchunks / U;                 // whole trips; the < U chunks left are done by the caller's tail launch
  u64 sinkAcc = 0;
  auto process = [&](const Vec16<T> (&r)[U], u64 c) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      u64 word;
      if (MODE == 2) word = (u64)r[u].raw.x | ((u64)r[u].raw.y << 32);
      else { const u32 m = inside_bits<T>(r[u], iso); word = group_or<LPW>((u64)m << (sub * VPL)); }
      if (MODE == 1) { sinkAcc |= word; continue; }
      if (MODE == 3) { sinkAcc |= word; if (u != U - 1) continue; word = sinkAcc; }
      if (STORE == 0) {
        if (last) {
          const u64 widx = (c + u) * VPL + lane / LPW;
          bits[widx] = word;
          if (lgWordsPerSlice >= 0 && word) sliceOcc[widx >> lgWordsPerSlice] = 1u;
        }
      } else {
        if (last) stage[wib][u * VPL + lane / LPW] = word;
      }
    }
    if (STORE == 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int i = lane; i < U * VPL; i += 64) {
        const u64 word = stage[wib][i];
        const u64 widx = c * VPL + i;
        bits[widx] = word;
        if (lgWordsPerSlice >= 0 && word) sliceOcc[widx >> lgWordsPerSlice] = 1u;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };
  if (!PIPE) {
    for (u64 t = wave; t < nfull; t += nwaves) {
      const u64 c = t * U;
      Vec16<T> r[U];
#pragma unroll
      for (int u = 0; u < U; u++) load16<T, NT>(r[u], vox + ((c + u) * 64 + lane) * VPL);
      process(r, c);
    }
  } else {
    u64 t = wave;
    if (t >= nfull) return;
    Vec16<T> a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; u++) load16<T, NT>(a[u], vox + ((t * U + u) * 64 + lane) * VPL);
    for (;;) {
      const u64 tn = t + nwaves;
      if (tn < nfull) {
#pragma unroll
        for (int u = 0; u < U; u++) load16<T, NT>(b[u], vox + ((tn * U + u) * 64 + lane) * VPL);
      }
      process(a, t * U);
      if (tn >= nfull) break;
      const u64 tnn = tn + nwaves;
      if (tnn < nfull) {
#pragma unroll
        for (int u = 0; u < U; u++) load16<T, NT>(a[u], vox + ((tnn * U + u) * 64 + lane) * VPL);
      }
      process(b, tn * U);
      if (tnn >= nfull) break;
      t = tnn;
    }
  }
  if (MODE == 1 && sinkAcc == 0x123456789abcdefull) bits[0] = sinkAcc;
}

// ---------------------------------------------------------------------------------------------------------
// v3: one pixel per lane per load (4-byte pixels): the compare's 64-bit lane mask IS the packed word -- no cross-lane
// work at all; 256 B per load instruction.
// ---------------------------------------------------------------------------------------------------------
template <int U, bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_classify_ballot(const float *__restrict__ vox, u64 *__restrict__ bits, u64 nwords, double isoD,
                                                           u32 *__restrict__ sliceOcc, int lgWordsPerSlice) {
  constexpr int WAVES = BLOCK / 64;
  const float iso = (float)isoD;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u64 wave = (u64)blockIdx.x * WAVES + wib;
  const u64 nwaves = (u64)gridDim.x * WAVES;
  const u64 ntrips = nwords / U;
  for (u64 t = wave; t < ntrips; t += nwaves) {
    const u64 w0 = t * U;
    float r[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float *p = vox + (w0 + u) * 64 + lane;
      r[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    u64 mine = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const u64 word = __ballot(!(r[u] < iso));
      if (lane == (u & 63)) mine = word;
    }
    if (lane < U) {
      bits[w0 + lane] = mine;
      if (lgWordsPerSlice >= 0 && mine) sliceOcc[(w0 + lane) >> lgWordsPerSlice] = 1u;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// v4: LDS-DMA stream (global_load_lds_dwordx4, 1 KiB per wave instruction straight into LDS, no VGPRs while in
// flight), ring of SLOTS trips per wave; the wave reads its own 16 bytes back with ds_read_b128.
// ---------------------------------------------------------------------------------------------------------
template <class T, int U, int SLOTS, int BLOCK, int AUX>
__global__ __launch_bounds__(BLOCK) void k_classify_dma(const T *__restrict__ vox, u64 *__restrict__ bits, u64 nchunks, double isoD,
                                                        u32 *__restrict__ sliceOcc, int lgWordsPerSlice) {
  constexpr int VPL = 16 / sizeof(T);
  constexpr int LPW = 64 / VPL;
  constexpr int WAVES = BLOCK / 64;
  __shared__ __attribute__((aligned(16))) unsigned char ring[WAVES][SLOTS][U][1024];
  const T iso = (T)isoD;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u64 wave = (u64)blockIdx.x * WAVES + wib;
  const u64 nwaves = (u64)gridDim.x * WAVES;
  const int sub = lane % LPW;
  const bool last = sub == LPW - 1;
  const u64 nfull = nchunks / U;
  auto issue = [&](u64 t, int slot) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const T *g = vox + ((t * U + u) * 64 + lane) * VPL;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                       (__attribute__((address_space(3))) void *)&ring[wib][slot][u][0], 16, 0, AUX);
    }
  };
  u64 t = wave;
  if (t >= nfull) return;
  // prologue: fill SLOTS-1 slots
  u64 tIssue = t;
  int sIssue = 0;
#pragma unroll
  for (int s = 0; s < SLOTS - 1; s++) {
    if (tIssue < nfull) issue(tIssue, s);
    tIssue += nwaves;
  }
  sIssue = SLOTS - 1;
  int sRead = 0;
  for (; t < nfull; t += nwaves) {
    if (tIssue < nfull) issue(tIssue, sIssue);
    tIssue += nwaves;
    sIssue = (sIssue + 1 == SLOTS) ? 0 : sIssue + 1;
    // wait until the oldest slot has landed: at most (SLOTS-1)*U loads may stay in flight
    if (SLOTS == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U) : "memory");
    else if (SLOTS == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * U) : "memory");
    else if (SLOTS == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * U) : "memory");
    const u64 c = t * U;
#pragma unroll
    for (int u = 0; u < U; u++) {
      Vec16<T> r;
      r.raw = *reinterpret_cast<const uint4 *>(&ring[wib][sRead][u][lane * 16]);
      const u32 m = inside_bits<T>(r, iso);
      const u64 word = group_or<LPW>((u64)m << (sub * VPL));
      if (last) {
        const u64 widx = (c + u) * VPL + lane / LPW;
        bits[widx] = word;
        if (lgWordsPerSlice >= 0 && word) sliceOcc[widx >> lgWordsPerSlice] = 1u;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot is re-filled next iteration: its reads must be done
    sRead = (sRead + 1 == SLOTS) ? 0 : sRead + 1;
  }
}


// ---------------------------------------------------------------------------------------------------------
// v5: a block owns SPAN consecutive trips per wave (4*SPAN*U KiB of voxels); the words go to LDS and are flushed
// once per span as whole 16-byte lanes (POLICY 0 plain, 1 nontemporal, 2 sc1 via agent-scope atomic store).
// ---------------------------------------------------------------------------------------------------------
template <int POLICY>
__device__ __forceinline__ void store_words16(u64 *dst, u64 a, u64 b) {
  if (POLICY == 1) { __builtin_nontemporal_store(a, dst); __builtin_nontemporal_store(b, dst + 1); }
  else if (POLICY == 2) { __hip_atomic_store(dst, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(dst + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  else { ulonglong2 v; v.x = a; v.y = b; *reinterpret_cast<ulonglong2 *>(dst) = v; }
}

template <class T, int U, int SPAN, int POLICY, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_span_variant(const T *__restrict__ vox, u64 *__restrict__ bits, u64 nchunks, double isoD, u64 wrapMask = ~0ull) {
  constexpr int VPL = 16 / sizeof(T);
  constexpr int LPW = 64 / VPL;
  constexpr int WAVES = BLOCK / 64;
  constexpr int WORDS = WAVES * SPAN * U * VPL;          // words per block span
  __shared__ __attribute__((aligned(16))) u64 stage[WORDS];
  const T iso = (T)isoD;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane % LPW;
  const bool last = sub == LPW - 1;
  const u64 tripsPerBlock = (u64)WAVES * SPAN;
  const u64 nspans = nchunks / (tripsPerBlock * U);       // whole spans only (harness sizes divide evenly)
  for (u64 sp = blockIdx.x; sp < nspans; sp += gridDim.x) {
    const u64 t0 = sp * tripsPerBlock;
#pragma unroll 1
    for (int i = 0; i < SPAN; i++) {
      const u64 tl = (u64)i * WAVES + wib;                // trip inside the span: waves interleave
      const u64 c = (t0 + tl) * U;
      Vec16<T> r[U];
#pragma unroll
      for (int u = 0; u < U; u++) load16<T, true>(r[u], vox + ((c + u) * 64 + lane) * VPL);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const u32 m = inside_bits<T>(r[u], iso);
        const u64 word = group_or<LPW>((u64)m << (sub * VPL));
        if (last) stage[(tl * U + u) * VPL + lane / LPW] = word;
      }
    }
    __syncthreads();
    u64 *dst = bits + ((t0 * U * VPL) & wrapMask);
    if constexpr (POLICY >= 3) {
      constexpr int AUX = POLICY == 3 ? 16 : (POLICY == 4 ? 17 : (POLICY == 5 ? 18 : 0));
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      auto rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, WORDS * 8, 0x00020000);
      for (int i = threadIdx.x * 2; i < WORDS; i += BLOCK * 2) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(&stage[i]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i * 8, 0, AUX);
      }
    } else
    for (int i = threadIdx.x * 2; i < WORDS; i += BLOCK * 2) store_words16<POLICY>(dst + i, stage[i], stage[i + 1]);
    __syncthreads();
  }
}

__global__ void k_compare(const u64 *a, const u64 *b, size_t n, u32 *diff) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && a[i] != b[i]) atomicAdd(diff, 1u);
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <class F> float run(F &&f, int reps) {
    f(); f();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < reps; i++) {
      CK(hipEventRecord(a, 0)); f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
  }
};

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const size_t nvox = (size_t)n * n * n;
  const size_t bytes = nvox * sizeof(float);
  float *vox; u64 *bitsRef, *bitsVar; u32 *occ, *diff;
  const size_t nwords = nvox / 64;
  CK(hipMalloc(&vox, bytes)); CK(hipMalloc(&bitsRef, nwords * 8)); CK(hipMalloc(&bitsVar, nwords * 8));
  CK(hipMalloc(&occ, n * 4)); CK(hipMalloc(&diff, 4));
  hipLaunchKernelGGL(k_fill_ml, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, 0, vox, n);
  CK(hipDeviceSynchronize());
  const u64 nchunks = nwords / 4;       // 1 KiB chunks (f32: 4 words each)
  int lg = 0; while ((1ull << lg) < (u64)n * (n / 64)) lg++;
  Timer T;
  auto report = [&](const char *name, float ms, bool check) {
    u32 d = 0;
    if (check) {
      CK(hipMemset(diff, 0, 4));
      hipLaunchKernelGGL(k_compare, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, 0, bitsRef, bitsVar, nwords, diff);
      CK(hipMemcpy(&d, diff, 4, hipMemcpyDeviceToHost));
      CK(hipMemset(bitsVar, 0xff, nwords * 8));
    }
    printf("%-44s %8.4f ms  %7.1f GB/s  %s\n", name, ms, bytes / (ms * 1e-3) / 1e9, check ? (d ? "MISMATCH" : "ok") : "");
    fflush(stdout);
  };
  // reference result with the product kernel
  hipLaunchKernelGGL((k_classify_flat<float, 8, true>), dim3(2048), dim3(256), 0, 0, vox, bitsRef, nchunks, 0.5, occ, lg, (u64)0);
  CK(hipDeviceSynchronize());

  report("read-only U8 nt grid 2048", T.run([&] { hipLaunchKernelGGL((k_read_only<8, true>), dim3(2048), dim3(256), 0, 0, (const uint4 *)vox, nchunks, diff); }, reps), false);
  report("product grid 2048", T.run([&] { hipLaunchKernelGGL((k_classify_flat<float, 8, true>), dim3(2048), dim3(256), 0, 0, vox, bitsVar, nchunks, 0.5, occ, lg, (u64)0); }, reps), true);
#define SPB(U, SPAN, BLOCK, GRID)                                                                                                \
  {                                                                                                                              \
    char nm[96];                                                                                                                 \
    snprintf(nm, sizeof nm, "span U%d span %d block %d grid %d (flush %d KiB)", U, SPAN, BLOCK, GRID, (BLOCK / 64) * SPAN * U * 4 * 8 / 1024); \
    report(nm, T.run([&] { hipLaunchKernelGGL((k_span_variant<float, U, SPAN, 3, BLOCK>), dim3(GRID), dim3(BLOCK), 0, 0, vox, bitsVar, nchunks, 0.5); }, reps), true); \
  }
  SPB(4, 64, 256, 512)
  SPB(4, 64, 256, 768)
  SPB(4, 64, 256, 1024)
  SPB(4, 32, 512, 256)
  SPB(4, 32, 512, 512)
  SPB(2, 128, 256, 512)
  SPB(2, 128, 256, 1024)
  SPB(2, 128, 256, 1536)
  SPB(2, 64, 512, 512)
  SPB(4, 16, 1024, 256)
  SPB(4, 128, 128, 1024)
  SPB(4, 128, 128, 768)
  SPB(8, 32, 256, 256)
  SPB(4, 64, 256, 512)
  SPB(6, 32, 256, 256)
  SPB(6, 32, 256, 512)
  SPB(3, 64, 256, 512)
  SPB(3, 64, 256, 768)
  SPB(5, 32, 256, 512)
  report("read-only U8 nt grid 2048 (again)", T.run([&] { hipLaunchKernelGGL((k_read_only<8, true>), dim3(2048), dim3(256), 0, 0, (const uint4 *)vox, nchunks, diff); }, reps), false);
  return 0;
}
