// Issue cost of the f64 vector instructions the projection walk is made of, on gfx950.
// Each kernel runs ITER x 16 independent instructions of one kind per wave; the grid is 256 CUs x 4 SIMDs x W waves,
// so (time x clock) / (ITER x 16 x W) is the cycles one SIMD spends per wave-instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o f64_issue f64_issue.hip ; run: ./f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 4096;

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed, float fseed) {
  double d[16];
  float f[16];
  int n[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { d[i] = seed + i + threadIdx.x; f[i] = fseed + i + threadIdx.x; n[i] = i; }
  const double a = seed * 0.5 + 1.0, b = seed + 0.25;
  for (int it = 0; it < ITER; it++) {
#define S_FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));
#define S_MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(a));
#define S_ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(b));
#define S_CVT_D_F(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
#define S_CVT_F_D(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
#define S_CVT_I_D(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(d[i]));
#define S_FLOOR(i) asm volatile("v_floor_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
#define S_FRACT(i) asm volatile("v_fract_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
#define S_RCP(i) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
#define S_RSQ(i) asm volatile("v_rsq_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
#define S_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fseed));
#define S_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(a));
#define S_CMP(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(a) : "vcc");
#define S_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(it) : "vcc");
#define S_MIN(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(a));
#define S_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(a));
#define S_LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(n[i]));
#define S_DIVFIX(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));
    if (OP == 0) { REP16(S_FMA) }
    if (OP == 1) { REP16(S_MUL) }
    if (OP == 2) { REP16(S_ADD) }
    if (OP == 3) { REP16(S_CVT_D_F) }
    if (OP == 4) { REP16(S_CVT_F_D) }
    if (OP == 5) { REP16(S_CVT_I_D) }
    if (OP == 6) { REP16(S_FLOOR) }
    if (OP == 7) { REP16(S_FRACT) }
    if (OP == 8) { REP16(S_RCP) }
    if (OP == 9) { REP16(S_RSQ) }
    if (OP == 10) { REP16(S_FMA32) }
    if (OP == 11) { REP16(S_PKFMA32) }
    if (OP == 12) { REP16(S_CMP) }
    if (OP == 13) { REP16(S_CNDMASK) }
    if (OP == 14) { REP16(S_MIN) }
    if (OP == 15) { REP16(S_MOV64) }
    if (OP == 16) { REP16(S_LDEXP) }
    if (OP == 17) { REP16(S_DIVFIX) }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += d[i] + (double)f[i] + (double)n[i];
  if (s == 12345.678) out[0] = s;
}

template <int OP>
static void run(const char *name, double *out, double mhz) {
  for (int wavesPerSimd : {1, 2, 4}) {
    const dim3 grid(256 * wavesPerSimd), block(256);      // 4 waves per block = one per SIMD, 256 CUs
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, out, 1.5, 2.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, out, 1.5, 2.5f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double cycles = ms * 1e-3 * mhz * 1e6;
    printf("%-16s waves/SIMD %d: %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", name, wavesPerSimd, ms,
           cycles / ((double)ITER * 16 * wavesPerSimd));
  }
}

int main() {
  double *out;
  CHECK(hipMalloc(&out, 64));
  int khz = 0;
  CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
  const double mhz = khz / 1000.0;
  printf("clock %.0f MHz (nominal peak; cycles below assume it)\n", mhz);
  run<0>("v_fma_f64", out, mhz);
  run<1>("v_mul_f64", out, mhz);
  run<2>("v_add_f64", out, mhz);
  run<3>("v_cvt_f64_f32", out, mhz);
  run<4>("v_cvt_f32_f64", out, mhz);
  run<5>("v_cvt_i32_f64", out, mhz);
  run<6>("v_floor_f64", out, mhz);
  run<7>("v_fract_f64", out, mhz);
  run<8>("v_rcp_f64", out, mhz);
  run<9>("v_rsq_f64", out, mhz);
  run<10>("v_fma_f32", out, mhz);
  run<11>("v_pk_fma_f32", out, mhz);
  run<12>("v_cmp_lt_f64", out, mhz);
  run<13>("v_cndmask_b32", out, mhz);
  run<14>("v_min_f64", out, mhz);
  run<15>("v_mov_b64", out, mhz);
  run<16>("v_ldexp_f64", out, mhz);
  run<17>("v_div_fixup_f64", out, mhz);
  return 0;
}
