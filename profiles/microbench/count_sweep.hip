// Development harness (not part of the product library): the count + scan kernel on the bench volume, after either
// flavour of the sweep, timed alone (HIP events around the one launch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o count_sweep count_sweep.hip
#include "../../midas-journal-740_amd/csrc/cuberille_kernels.hip"

#include <cstdio>
#include <vector>
#include <algorithm>

using namespace cuberille;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

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

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const size_t nvox = (size_t)n * n * n;
  float *vox;
  CK(hipMalloc(&vox, nvox * 4));
  hipLaunchKernelGGL(k_fill_ml, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, 0, vox, n);
  CK(hipDeviceSynchronize());
  Grid g{};
  g.nx = g.ny = g.nzb = n; g.W = n / 64; g.lastpos = 63;
  g.wShift = g.yShift = -1;
  for (int b = 0; b < 31; b++) { if (g.W == (1 << b)) g.wShift = b; if (g.ny == (1 << b)) g.yShift = b; }
  g.gnz = n; g.zglob0 = 0; g.oz0 = 0; g.oz1 = n; g.cz0 = 0;
  const size_t nwords = nvox / 64, nseg = nwords / 64, nblk = (nwords + COUNT_WB - 1) / COUNT_WB;
  Workspace w{};
  w.vox = vox;
  CK(hipMalloc(&w.bits, nwords * 8)); CK(hipMalloc(&w.sliceOcc, n * 4)); CK(hipMalloc(&w.prefix, nwords * 4));
  CK(hipMalloc(&w.segPre, nseg * 8)); CK(hipMalloc(&w.blockTot, nblk * 8)); CK(hipMalloc(&w.blockBase, nblk * 16));
  CK(hipMalloc(&w.totals, sizeof(Totals))); CK(hipMalloc(&w.vqueue, nwords * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  Tuning tn;
  for (int variant = 0; variant < 2; variant++) {
    tn.classify_variant = variant;
    std::vector<float> tc, tk;
    Totals t{};
    for (int i = 0; i < reps + 2; i++) {
      CK(hipMemsetAsync(w.sliceOcc, 0, n * 4, 0));
      CK(hipMemsetAsync(w.totals, 0, sizeof(Totals), 0));
      CK(hipEventRecord(a, 0));
      CK(launch_classify(CUBERILLE_PIX_F32, w, g, 0.5, 0, n, tn, 0));
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (i >= 2) tc.push_back(ms);
      CK(hipEventRecord(a, 0));
      CK(launch_count(w, g, nwords, 1, 0));
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b));
      if (i >= 2) tk.push_back(ms);
      CK(hipMemcpy(&t, w.totals, sizeof(Totals), hipMemcpyDeviceToHost));
    }
    std::sort(tc.begin(), tc.end()); std::sort(tk.begin(), tk.end());
    printf("classify variant %d: classify %.4f ms  count %.4f ms   totV %llu totQ %llu vertexWords %u\n", variant,
           tc[tc.size() / 2], tk[tk.size() / 2], t.totV, t.totQ, t.nVertexWords);
  }
  // parts of the kernel switched off: 1 no last-block scan, 2 no publish either, 4 no corner logic (phase 2), 6 = 2 + 4
  auto timeMode = [&](auto tag, const char *what) {
    constexpr int MODE = decltype(tag)::value;
    std::vector<float> tk;
    const unsigned blocks = (unsigned)((nwords + COUNT_WB - 1) / COUNT_WB);
    for (int i = 0; i < reps; i++) {
      CK(hipMemsetAsync(w.totals, 0, sizeof(Totals), 0));
      CK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(k_count<MODE>, dim3(blocks), dim3(256), 0, 0, w.bits, w.sliceOcc, g, nwords, 1, w.prefix, w.segPre, w.blockTot,
                         w.vqueue, w.totals);
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      tk.push_back(ms);
    }
    std::sort(tk.begin(), tk.end());
    printf("count mode %d (%s): %.4f ms\n", MODE, what, tk[tk.size() / 2]);
  };
  timeMode(std::integral_constant<int, 0>(), "k_count alone");
  timeMode(std::integral_constant<int, 2>(), "no block scan");
  timeMode(std::integral_constant<int, 4>(), "no corner logic");
  timeMode(std::integral_constant<int, 6>(), "faces + wave scans + prefix stores only");
  // count again without anything in between (bits as the previous count left the caches)
  {
    std::vector<float> tk;
    for (int i = 0; i < reps; i++) {
      CK(hipMemsetAsync(w.totals, 0, sizeof(Totals), 0));
      CK(hipEventRecord(a, 0));
      CK(launch_count(w, g, nwords, 1, 0));
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      tk.push_back(ms);
    }
    std::sort(tk.begin(), tk.end());
    printf("count back to back: %.4f ms\n", tk[tk.size() / 2]);
  }
  return 0;
}
