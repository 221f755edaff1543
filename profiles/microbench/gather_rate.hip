// Development harness (not part of the product library): how many vector-memory LOAD instructions a CU retires per cycle when
// the lanes of a wave read scattered addresses -- the rate that bounds k_emit_cells / k_emit_points_dense.
// Every lane issues R independent loads of WIDTH bytes per trip from a table of `span` bytes, addresses from a hash (scattered),
// lane-consecutive (coalesced) or one per wave (broadcast); 8 waves per SIMD, grid = 8 workgroups of 256 per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class V> __device__ __forceinline__ unsigned fold(const V &v);
template <> __device__ __forceinline__ unsigned fold(const unsigned &v) { return v; }
template <> __device__ __forceinline__ unsigned fold(const uint2 &v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ unsigned fold(const uint4 &v) { return v.x ^ v.y ^ v.z ^ v.w; }
struct u3 { unsigned x, y, z; };
template <> __device__ __forceinline__ unsigned fold(const u3 &v) { return v.x ^ v.y ^ v.z; }

// PATTERN 0 scattered, 1 coalesced, 2 one address per wave
template <class V, int PATTERN>
__global__ __launch_bounds__(256) void k_gather(const unsigned *__restrict__ table, unsigned spanWords, int trips, unsigned *out) {
  constexpr int R = 8;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  unsigned acc = 0, h = t * 2654435761u + 12345u;
  constexpr unsigned stepW = sizeof(V) / 4 < 1 ? 1 : (sizeof(V) == 12 ? 3 : sizeof(V) / 4);
  const unsigned slots = spanWords / stepW;
  for (int i = 0; i < trips; i++) {
    V v[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      h = h * 1664525u + 1013904223u;
      const unsigned hw = __builtin_amdgcn_readfirstlane(h);   // (the wave's: the other two patterns)
      const unsigned s = PATTERN == 0 ? (h >> 4) % slots : PATTERN == 1 ? (((hw >> 4) % (slots - 64)) & ~63u) + lane : (hw >> 4) % slots;
      v[r] = *reinterpret_cast<const V *>(table + (size_t)s * stepW);
    }
#pragma unroll
    for (int r = 0; r < R; r++) acc ^= fold(v[r]);
  }
  if (acc == 0x12345678u) out[t] = acc;
}

template <class V, int PATTERN>
static void run(const char *what, const unsigned *table, unsigned spanBytes, unsigned *out, int cus) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int trips = 200, blocks = cus * 8;
  hipLaunchKernelGGL((k_gather<V, PATTERN>), dim3(blocks), dim3(256), 0, 0, table, spanBytes / 4, 10, out);
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL((k_gather<V, PATTERN>), dim3(blocks), dim3(256), 0, 0, table, spanBytes / 4, trips, out);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double waveLoads = (double)blocks * 4 * trips * 8;                   // wave-level load instructions
  const double cyc = ms * 1e-3 * 2.4e9;                                     // at 2.4 GHz
  printf("%-44s %8.3f ms   %6.1f cycles per wave-load per CU   %7.1f GB/s of lanes\n", what, ms, cyc / (waveLoads / cus),
         waveLoads * 64 * sizeof(V) / (ms * 1e-3) / 1e9);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const size_t big = 1ull << 30;
  unsigned *table, *out;
  CK(hipMalloc(&table, big)); CK(hipMemset(table, 1, big));
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  printf("%d CUs; 8 waves per SIMD; 8 independent loads per lane per trip\n", cus);
  for (unsigned span : {16u << 10, 2u << 20, 64u << 20, 1u << 30}) {
    printf("-- table of %u KiB (%s)\n", span >> 10, span <= (16u << 10) ? "L1" : span <= (2u << 20) ? "L2" : span <= (64u << 20) ? "L2 of all XCDs / Infinity Cache" : "HBM");
    run<unsigned, 0>("scattered  4 B", table, span, out, cus);
    run<uint2, 0>("scattered  8 B", table, span, out, cus);
    run<u3, 0>("scattered 12 B", table, span, out, cus);
    run<uint4, 0>("scattered 16 B", table, span, out, cus);
    run<unsigned, 1>("coalesced  4 B", table, span, out, cus);
    run<uint2, 1>("coalesced  8 B", table, span, out, cus);
    run<uint4, 1>("coalesced 16 B", table, span, out, cus);
    run<uint2, 2>("one address per wave, 8 B", table, span, out, cus);
  }
  return 0;
}
