// Issue cost of the integer / boolean vector instructions the word classification (k_count, k_emit_points_dense) is made
// of, on gfx950 -- same method as f64_issue.hip: ITER x 16 independent instructions of one kind per wave, 1 / 2 / 4 waves
// per SIMD on every CU; (time x clock) / (ITER x 16 x waves) = cycles one SIMD spends per wave-instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o int_issue int_issue.hip ; run: ./int_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 4096;
#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed) {
  unsigned n[16];
  unsigned long long q[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { n[i] = seed * (i + 3) + threadIdx.x; q[i] = (unsigned long long)n[i] * 0x9E3779B97F4A7C15ull; }
  const unsigned a = seed * 7u + 1u, b = seed ^ 0x5bd1e995u;
  for (int it = 0; it < ITER; it++) {
#define S_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(a));
#define S_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(n[i]) : "v"(a), "v"(b));
#define S_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(a), "v"(b));
#define S_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(a), "v"(b));
#define S_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(n[i]) : "v"(a));
#define S_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(n[i]) : "v"(a));
#define S_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(n[i]) : "v"(a));
#define S_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i]));
#define S_LSHR64(i) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q[i]) : "v"(a));
#define S_MUL(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(a));
#define S_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(n[i]) : "v"(a), "v"(b));
#define S_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(a));
#define S_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(a) : "vcc");
#define S_FFBL(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(n[i]));
#define S_DPP(i) asm volatile("v_or_b32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(n[i]));
    if (OP == 0) { REP16(S_AND) }
    if (OP == 1) { REP16(S_BITOP3) }
    if (OP == 2) { REP16(S_ANDOR) }
    if (OP == 3) { REP16(S_OR3) }
    if (OP == 4) { REP16(S_BCNT) }
    if (OP == 5) { REP16(S_ALIGNBIT) }
    if (OP == 6) { REP16(S_LSHLOR) }
    if (OP == 7) { REP16(S_LSHL64) }
    if (OP == 8) { REP16(S_LSHR64) }
    if (OP == 9) { REP16(S_MUL) }
    if (OP == 10) { REP16(S_DOT4) }
    if (OP == 11) { REP16(S_ADD) }
    if (OP == 12) { REP16(S_CNDMASK) }
    if (OP == 13) { REP16(S_FFBL) }
    if (OP == 14) { REP16(S_DPP) }
  }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += n[i] + (unsigned)q[i] + (unsigned)(q[i] >> 32);
  if (s == 0x12345678u) out[0] = s;
}

template <int OP>
static void run(const char *name, unsigned *out, double mhz) {
  for (int wavesPerSimd : {1, 2, 4}) {
    const dim3 grid(256 * wavesPerSimd), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, out, 3u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, out, 3u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-16s waves/SIMD %d: %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", name, wavesPerSimd, ms,
           ms * 1e-3 * mhz * 1e6 / ((double)ITER * 16 * wavesPerSimd));
  }
}

int main() {
  unsigned *out;
  CHECK(hipMalloc(&out, 64));
  int khz = 0;
  CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
  const double mhz = khz / 1000.0;
  printf("clock %.0f MHz (nominal peak; cycles below assume it)\n", mhz);
  run<0>("v_and_b32", out, mhz);
  run<1>("v_bitop3_b32", out, mhz);
  run<2>("v_and_or_b32", out, mhz);
  run<3>("v_or3_b32", out, mhz);
  run<4>("v_bcnt_u32_b32", out, mhz);
  run<5>("v_alignbit_b32", out, mhz);
  run<6>("v_lshl_or_b32", out, mhz);
  run<7>("v_lshlrev_b64", out, mhz);
  run<8>("v_lshrrev_b64", out, mhz);
  run<9>("v_mul_lo_u32", out, mhz);
  run<10>("v_dot4_u32_u8", out, mhz);
  run<11>("v_add_u32", out, mhz);
  run<12>("v_cndmask_b32", out, mhz);
  run<13>("v_ffbl_b32", out, mhz);
  run<14>("v_or_b32_dpp", out, mhz);
  return 0;
}
