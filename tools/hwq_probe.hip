// Which streams share a hardware queue?  N streams created one after the other; for every k a long single-workgroup kernel on stream 0
// and the same on stream k, started together: ~T if they run side by side (different queues), ~2T if one waits for the other (same queue).
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/hwq_probe tools/hwq_probe.hip && GPU_MAX_HW_QUEUES=6 tools/bin/hwq_probe [N] [destroy_every]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long ticks, unsigned long long* out) {
  const unsigned long long t0 = wall_clock64();
  while ((long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
  if (out) *out = wall_clock64() - t0;
}
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 14;
  const int spacer_mode = argc > 2 ? atoi(argv[2]) : 0;   // 1: also create (and keep) an unused stream between every two
  std::vector<hipStream_t> s(N);
  std::vector<hipStream_t> spacers;
  for (int i = 0; i < N; ++i) {
    (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    if (spacer_mode) { hipStream_t d; (void)hipStreamCreateWithFlags(&d, hipStreamNonBlocking); spacers.push_back(d); }
  }
  const long T = 30000;  // 300 us at 100 MHz
  for (int k = 0; k < N; ++k) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[k], 100, nullptr); }
  (void)hipDeviceSynchronize();
  printf("GPU_MAX_HW_QUEUES=%s, %d streams%s: pair time / single time (1 = side by side, 2 = same queue)\n", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)", N, spacer_mode ? " with spacers" : "");
  for (int a = 0; a < (N < 4 ? N : 4); ++a) {
    printf("stream %d vs:", a);
    for (int k = 0; k < N; ++k) {
      if (k == a) { printf("  - "); continue; }
      (void)hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[a], T, nullptr);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[k], T, nullptr);
      (void)hipStreamSynchronize(s[a]);
      (void)hipStreamSynchronize(s[k]);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf(" %3.1f", us / 300.0);
    }
    printf("\n");
  }
  return 0;
}
