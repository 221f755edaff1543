// Development harness (not part of the product library): the count + scan kernel alone (HIP events around the one
// launch) on a synthetic uint8 field whose surface density is a parameter, untiled and LDS-tiled, with parts of the
// kernel switched off (MODE bits: 2 stop before the block-level scans and the vertex-word queue, 4 no corner logic).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o count_sweep count_sweep.hip
//   ./count_sweep [n=1024] [wavelength=48] [reps=10]
#include "../../midas-journal-740_amd/csrc/cuberille_kernels.hip"

#include <cstdio>
#include <vector>
#include <algorithm>

using namespace cuberille;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// blobs of about `wl` voxels: 128 + 42 (sin + sin + sin), a little hashed roughness
__global__ void k_fill(unsigned char *vox, int n, float wl) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n * n * n;
  if (i >= total) return;
  const int x = (int)(i % n), y = (int)((i / n) % n), z = (int)(i / ((size_t)n * n));
  const float k = 6.2831853f / wl;
  unsigned h = (unsigned)i * 2654435761u;
  h ^= h >> 15;
  const float v = 128.f + 42.f * (sinf(k * x) + sinf(1.13f * k * y + 1.f) + sinf(0.91f * k * z + 2.f)) + (float)(h & 7u) - 3.5f;
  vox[i] = (unsigned char)fminf(fmaxf(v, 0.f), 255.f);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  const float wl = argc > 2 ? (float)atof(argv[2]) : 48.f;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const size_t nvox = (size_t)n * n * n;
  unsigned char *vox;
  CK(hipMalloc(&vox, nvox));
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, 0, vox, n, wl);
  CK(hipDeviceSynchronize());
  Grid g{};
  g.nx = g.ny = g.nzb = n; g.W = n / 64; g.lastpos = 63;
  g.wShift = g.yShift = -1;
  for (int b = 0; b < 31; b++) { if (g.W == (1 << b)) g.wShift = b; if (g.ny == (1 << b)) g.yShift = b; }
  g.gnz = n; g.zglob0 = 0; g.oz0 = 0; g.oz1 = n; g.cz0 = 0;
  const size_t nwords = nvox / 64, nseg = nwords / 64, nblk = (nwords + COUNT_WB - 1) / COUNT_WB;
  Workspace w{};
  w.vox = vox;
  CK(hipMalloc(&w.bits, (nwords + (size_t)n * g.W) * 8)); CK(hipMalloc(&w.sliceOcc, n * 4)); CK(hipMalloc(&w.prefix, nwords * 4));
  CK(hipMalloc(&w.segPre, nseg * 8)); CK(hipMalloc(&w.blockTot, nblk * 8)); CK(hipMalloc(&w.blockBase, nblk * 16 * 8));
  CK(hipMalloc(&w.totals, sizeof(Totals))); CK(hipMalloc(&w.vqueue, nwords * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  Tuning tn;
  Params prm{};
  prm.iso = 128.0;
  CK(hipMemsetAsync(w.sliceOcc, 0, n * 4, 0));
  CK(launch_classify(CUBERILLE_PIX_U8, w, g, prm, 0, n, tn, 0));
  CK(hipDeviceSynchronize());
  const unsigned blocks = (unsigned)nblk;
  auto timeIt = [&](auto launch, const char *what) {
    std::vector<float> tk;
    Totals t{};
    for (int i = 0; i < reps + 1; i++) {
      CK(hipMemsetAsync(w.totals, 0, sizeof(Totals), 0));
      CK(hipEventRecord(a, 0));
      launch();
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (i) tk.push_back(ms);
    }
    CK(hipMemcpy(&t, w.totals, sizeof(Totals), hipMemcpyDeviceToHost));
    std::sort(tk.begin(), tk.end());
    printf("%-58s %.4f ms   (vertex words %u of %zu)\n", what, tk[tk.size() / 2], t.nVertexWords, nwords);
    fflush(stdout);
  };
#define KC(MODE, TILED, NT, VQ) hipLaunchKernelGGL((k_count<MODE, TILED, NT>), dim3(blocks), dim3(NT), 0, 0, w.bits, w.sliceOcc, g, nwords, 1, \
                                                    w.prefix, w.segPre, w.blockTot, VQ, w.totals, 0, w.blockBase, Gate{}, 0)
  printf("n %d wavelength %.0f\n", n, wl);
  {
    // where a block's time goes (round-4 review: the sphere's 0.03 ms for 1024 blocks "unexplained"): thread 0's clock at the
    // phase boundaries of every block (MODE 16), medians over the blocks, in shader cycles
    CK(hipMemsetAsync(w.totals, 0, sizeof(Totals), 0));
    CK(hipMemsetAsync(w.blockBase, 0, nblk * 16 * 8, 0));
    KC(16, false, 256, w.vqueue);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(nblk * 16);
    CK(hipMemcpy(st.data(), w.blockBase, nblk * 16 * 8, hipMemcpyDeviceToHost));
    const char *names[8] = {"start -> first barrier", "phase 1 (faces, 8 words per thread)", "barrier", "phase 2 (corner logic of queued words)",
                            "barrier", "phase 3 (wave scans, prefix stores)", "barrier", "segment scan + queue flush"};
    unsigned long long t0min = ~0ull, t8max = 0;
    for (size_t b = 0; b < nblk; b++) { t0min = std::min(t0min, st[b * 16]); t8max = std::max(t8max, st[b * 16 + 8]); }
    printf("phase stamps, untiled 256 (cycles; median / 90th percentile over %zu blocks; kernel span %llu cycles)\n", nblk, t8max - t0min);
    for (int ph = 0; ph < 8; ph++) {
      std::vector<unsigned long long> d(nblk);
      for (size_t b = 0; b < nblk; b++) d[b] = st[b * 16 + ph + 1] - st[b * 16 + ph];
      std::sort(d.begin(), d.end());
      printf("  %-44s %8llu / %8llu\n", names[ph], d[nblk / 2], d[nblk * 9 / 10]);
    }
    std::vector<unsigned long long> q(nblk), life(nblk), start(nblk);
    for (size_t b = 0; b < nblk; b++) { q[b] = st[b * 16 + 9]; life[b] = st[b * 16 + 8] - st[b * 16]; start[b] = st[b * 16] - t0min; }
    std::sort(q.begin(), q.end()); std::sort(life.begin(), life.end()); std::sort(start.begin(), start.end());
    printf("  queued words per block: median %llu, max %llu; block lifetime median %llu, max %llu; block start after the first: median %llu, max %llu\n",
           q[nblk / 2], q[nblk - 1], life[nblk / 2], life[nblk - 1], start[nblk / 2], start[nblk - 1]);
  }
  timeIt([&] { KC(0, false, 256, w.vqueue); }, "untiled: everything");
  timeIt([&] { KC(0, false, 256, (u32 *)nullptr); }, "untiled: no vertex-word queue");
  timeIt([&] { KC(2, false, 256, w.vqueue); }, "untiled: no block-level scans / queue");
  timeIt([&] { KC(4, false, 256, w.vqueue); }, "untiled: no corner logic");
  timeIt([&] { KC(6, false, 256, w.vqueue); }, "untiled: faces + wave scans + prefix stores only");
  timeIt([&] { KC(0, true, 512, w.vqueue); }, "tiled 512: everything");
  timeIt([&] { KC(4, true, 512, w.vqueue); }, "tiled 512: no corner logic");
  timeIt([&] { KC(6, true, 512, w.vqueue); }, "tiled 512: tile + faces + wave scans + prefix stores only");
#define KF(MODE, ZRUN) hipLaunchKernelGGL((k_count_dense<MODE>), dim3(blocks), dim3(512), 0, 0, w.bits, w.sliceOcc, g, nwords, 1, \
                                                    w.prefix, w.segPre, w.blockTot, w.vqueue, w.totals, ZRUN)
#define KT(MODE, ZRUN) hipLaunchKernelGGL((k_count<MODE, true, 512>), dim3(blocks), dim3(512), 0, 0, w.bits, w.sliceOcc, g, nwords, 1, \
                                                    w.prefix, w.segPre, w.blockTot, w.vqueue, w.totals, ZRUN, w.blockBase, Gate{}, 0)
  timeIt([&] { KF(0, 0); }, "dense form: everything");
  timeIt([&] { KF(8, 0); }, "dense form: no virtual words");
  timeIt([&] { KF(4, 0); }, "dense form: no corner logic");
  timeIt([&] { KF(12, 0); }, "dense form: no corner logic, no virtual words");
    if (((size_t)n * g.W) % COUNT_WB == 0) {
    timeIt([&] { KT(0, 8); }, "tiled 512 in columns of 8: everything");
    timeIt([&] { KF(0, 8); }, "dense form in columns of 8: everything");
    timeIt([&] { KF(8, 8); }, "dense form in columns of 8: no virtual words");
    timeIt([&] { KF(12, 8); }, "dense form in columns of 8: no corner logic, no virtual words");
  }
  timeIt([&] { KC(0, true, 256, w.vqueue); }, "tiled 256: everything");
  timeIt([&] { KC(0, true, 1024, w.vqueue); }, "tiled 1024: everything");
  return 0;
}
