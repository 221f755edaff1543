// Development harness (not part of the product library): what a chain of N small dependent launches costs on one stream,
// launched one by one and as one captured hipGraph (wall time around launch + wait, and the device interval between two events).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o launch_chain launch_chain.hip && ./launch_chain [n=10] [reps=300]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_step(unsigned *p, int i) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  p[t] = p[t] * 3u + (unsigned)i;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 10, reps = argc > 2 ? atoi(argv[2]) : 300;
  unsigned *p;
  CK(hipMalloc(&p, 64 * 256 * 4));
  CK(hipMemset(p, 0, 64 * 256 * 4));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto chain = [&] {
    CK(hipMemsetAsync(p, 0, 256, s));
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_step, dim3(64), dim3(256), 0, s, p, i);
  };
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  chain();
  CK(hipStreamEndCapture(s, &graph));
  const double t0 = now_ms();
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  const double tInst = now_ms() - t0;
  auto measure = [&](auto run, const char *what) {
    std::vector<double> wall, dev;
    for (int r = 0; r < reps + 20; r++) {
      const double w0 = now_ms();
      CK(hipEventRecord(a, s));
      run();
      CK(hipEventRecord(b, s));
      CK(hipEventSynchronize(b));
      const double w1 = now_ms();
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (r >= 20) { wall.push_back(w1 - w0); dev.push_back(ms); }
    }
    std::sort(wall.begin(), wall.end()); std::sort(dev.begin(), dev.end());
    printf("%-28s wall %.4f ms   between the events %.4f ms\n", what, wall[wall.size() / 2], dev[dev.size() / 2]);
  };
  printf("memset + %d dependent launches of 64 x 256 threads; graph instantiation %.3f ms\n", n, tInst);
  measure(chain, "one by one");
  measure([&] { CK(hipGraphLaunch(exec, s)); }, "hipGraphLaunch");
  // capture + instantiate + launch every time (what a changing argument list would cost)
  measure([&] {
    hipGraph_t g2; hipGraphExec_t e2;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(s, &g2));
    CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
    CK(hipGraphLaunch(e2, s));
    CK(hipStreamSynchronize(s));
    CK(hipGraphExecDestroy(e2)); CK(hipGraphDestroy(g2));
  }, "capture + instantiate + launch");
  return 0;
}
