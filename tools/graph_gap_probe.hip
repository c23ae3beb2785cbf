// graph_gap_probe.hip — what does a dependent kernel boundary cost on this GPU?  (measurement tool, not product code)
// N dependent kernels of 256 workgroups x 512 threads that each spin for `us` microseconds, launched (a) eagerly on one
// stream, (b) as one hipGraph, (c) as two graphs on two streams; reports (wall - N*us)/N per mode.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_gap_probe tools/graph_gap_probe.hip && /tmp/graph_gap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(512) void spin(long ticks, int* sink) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (ticks < 0) *sink = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int* sink; CK(hipMalloc(&sink, 4));
  CK(hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));  // one workgroup per CU, like the conv kernels
  hipStream_t s[4]; for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  const int N = 46, REP = 200;
  for (double us : {0.0, 20.0, 45.0}) {
    const long ticks = (long)(us * 100);   // wall_clock64: 100 MHz
    auto body = [&](hipStream_t st) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(512), 100 * 1024, st, ticks, sink); };
    // (a) eager
    body(s[0]); CK(hipStreamSynchronize(s[0]));
    double t = now(); for (int r = 0; r < REP; ++r) body(s[0]); CK(hipStreamSynchronize(s[0]));
    const double eager = (now() - t) / REP / N * 1e6;
    // (b) graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal)); body(s[0]); CK(hipStreamEndCapture(s[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s[0])); CK(hipStreamSynchronize(s[0]));
    t = now(); for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s[0])); CK(hipStreamSynchronize(s[0]));
    const double graph = (now() - t) / REP / N * 1e6;
    // (c) the same graph body on K streams at once (K separately instantiated graphs): aggregate time per kernel
    double multi[3] = {0, 0, 0};
    int ki = 0;
    for (int K : {2, 4}) {
      std::vector<hipGraphExec_t> ges(K);
      for (int k = 0; k < K; ++k) {
        hipGraph_t gk; CK(hipStreamBeginCapture(s[k], hipStreamCaptureModeThreadLocal)); body(s[k]); CK(hipStreamEndCapture(s[k], &gk));
        CK(hipGraphInstantiate(&ges[k], gk, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ges[k], s[k]));
      }
      for (int k = 0; k < K; ++k) CK(hipStreamSynchronize(s[k]));
      t = now();
      for (int r = 0; r < REP; ++r) for (int k = 0; k < K; ++k) CK(hipGraphLaunch(ges[k], s[k]));
      for (int k = 0; k < K; ++k) CK(hipStreamSynchronize(s[k]));
      multi[ki++] = (now() - t) / REP / N / K * 1e6;
    }
    printf("spin %5.1f us x %d dependent kernels: per kernel  eager %.2f us  graph %.2f us  2 streams %.2f us  4 streams %.2f us   (overhead = value - spin)\n",
           us, N, eager, graph, multi[0], multi[1]);
  }
  return 0;
}
