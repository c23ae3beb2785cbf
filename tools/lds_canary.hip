// lds_canary.hip — does a co-resident workgroup of ANOTHER kernel ever see its LDS modified?
// Canary workgroups fill their LDS with a pattern, spin, and verify; meanwhile an engine runs frames.
// Build: hipcc --offload-arch=gfx950 -O2 tools/lds_canary.hip -Iinclude -Lcaffe_rtpose_amd -lrtpose_mi355x -Wl,-rpath,$PWD/caffe_rtpose_amd -lpthread -o /tmp/lds_canary
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "rtpose_mi355x.h"

__global__ __launch_bounds__(256) void canary(int words, int spins, unsigned* bad, unsigned* first_bad) {
  extern __shared__ unsigned lds[];
  const unsigned tag = 0xA5000000u | (blockIdx.x << 12);
  for (int i = threadIdx.x; i < words; i += 256) lds[i] = tag ^ i;
  __syncthreads();
  for (int s = 0; s < spins; ++s) {
    for (int i = threadIdx.x; i < words; i += 256) {
      const unsigned v = lds[i];
      if (v != (tag ^ i)) {
        const unsigned n = atomicAdd(bad, 1u);
        if (n < 8) { first_bad[4 * n] = blockIdx.x; first_bad[4 * n + 1] = i; first_bad[4 * n + 2] = v; first_bad[4 * n + 3] = s; }
        lds[i] = tag ^ i;
      }
    }
    __builtin_amdgcn_s_sleep(20);
  }
}

int main(int argc, char** argv) {
  const int lds_kb = argc > 1 ? atoi(argv[1]) : 16;
  const int with_load = argc > 2 ? atoi(argv[2]) : 1;
  rtp_config cfg;
  rtp_config_default(&cfg);
  cfg.net_w = 320; cfg.net_h = 176; cfg.num_scales = 2; cfg.scale_gap = 0.25f; cfg.disp_w = 640; cfg.disp_h = 360; cfg.frames_in_flight = 3;
  rtp_engine* e = nullptr;
  if (rtp_engine_create(&cfg, &e)) { fprintf(stderr, "%s\n", rtp_last_error(nullptr)); return 1; }
  std::vector<float> x((size_t)2 * 3 * 176 * 320, 0.1f);
  std::atomic<bool> stop{false};
  std::thread load([&] {
    int pend = 0;
    std::vector<float> joints(96 * 18 * 3);
    while (!stop && with_load) {
      rtp_submit(e, x.data(), 0);
      if (++pend == 3) { uint64_t t; int n; rtp_collect(e, &t, joints.data(), &n); --pend; }
    }
    while (pend--) { uint64_t t; int n; rtp_collect(e, &t, joints.data(), &n); }
  });
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  unsigned *bad, *fb;
  hipMalloc(&bad, 4); hipMalloc(&fb, 128);
  hipMemset(bad, 0, 4); hipMemset(fb, 0, 128);
  const int words = lds_kb * 256;
  for (int it = 0; it < 300; ++it) {
    hipLaunchKernelGGL(canary, dim3(256), dim3(256), words * 4, st, words, 40, bad, fb);
    hipStreamSynchronize(st);
  }
  stop = true;
  load.join();
  unsigned hb, hfb[32];
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  hipMemcpy(hfb, fb, 128, hipMemcpyDeviceToHost);
  printf("lds %d KiB, load %d: corrupted words: %u\n", lds_kb, with_load, hb);
  for (unsigned i = 0; i < (hb < 8 ? hb : 8); ++i) printf("  wg %u word %u (byte %u) value %08x spin %u\n", hfb[4 * i], hfb[4 * i + 1], hfb[4 * i + 1] * 4, hfb[4 * i + 2], hfb[4 * i + 3]);
  rtp_engine_destroy(e);
  return 0;
}
