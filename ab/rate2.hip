#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(unsigned *out, unsigned seed, int iters) {
  unsigned f[8]; unsigned long long d[8];
  for (int i = 0; i < 8; i++) { f[i] = seed + threadIdx.x + i; d[i] = (unsigned long long)seed * 77 + i + threadIdx.x; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) { asm volatile("v_lshlrev_b64 %0, 1, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7])); }
      if (OP == 1) { asm volatile("v_lshrrev_b64 %0, 1, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7])); }
      if (OP == 2) { asm volatile("v_alignbit_b32 %0, %1, %2, 31" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7])); }
      if (OP == 3) { asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7])); }
      if (OP == 4) { asm volatile("v_and_b32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7])); }
      if (OP == 5) { asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7])); }
      if (OP == 6) { asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7])); }
      if (OP == 7) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]) : "vcc"); }
      if (OP == 8) { asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7])); }
      if (OP == 9) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7])); }
      if (OP == 10) { asm volatile("v_ffbl_b32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 7])); }
      if (OP == 11) { asm volatile("ds_bpermute_b32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(f[i]) : "v"(f[(i + 1) & 7] & 252), "v"(f[(i + 2) & 7])); }
    }
  }
  unsigned s = 0; for (int i = 0; i < 8; i++) s += f[i] + (unsigned)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, unsigned *out) {
  const int iters = 4096, blocks = 256 * 4, threads = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 3u, 16);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 3u, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-18s %.3f ms  -> %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (4.0 * iters * 8));
}
int main() {
  unsigned *out; (void)hipMalloc(&out, 256 * 4 * 256 * sizeof(unsigned));
  run<4>("v_and_b32", out); run<0>("v_lshlrev_b64", out); run<1>("v_lshrrev_b64", out); run<2>("v_alignbit_b32", out);
  run<3>("v_bcnt_u32_b32", out); run<5>("v_and_or_b32", out); run<6>("v_lshl_add_u64", out); run<7>("v_mad_u64_u32", out);
  run<8>("v_mul_lo_u32", out); run<9>("v_cndmask_b32", out); run<10>("v_ffbl_b32", out); run<11>("ds_bpermute+wait", out);
  return 0;
}
