// Are the 100 MHz wall clocks (s_memrealtime, wall_clock64()) of the 8 XCDs synchronised?  256+ workgroups spin on a global flag that the host
// sets, then each records wall_clock64() and its XCC id: the spread of the stamps across XCDs (beyond the ~1 us release jitter) is the offset.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/clock_sync_probe tools/clock_sync_probe.hip && tools/bin/clock_sync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void probe(volatile int* flag, unsigned long long* out, int* arrived) {
  if (threadIdx.x == 0) {
    atomicAdd_system(arrived, 1);
    long spins = 0;
    while (*flag == 0 && ++spins < 20000000L) { __builtin_amdgcn_s_sleep(1); }   // (bounded: a probe must never hang the box)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = wall_clock64();
    out[2 * blockIdx.x + 1] = xcc & 15;
  }
}
int main() {
  const int N = 256;
  int *flag, *arrived; unsigned long long* out;
  hipHostMalloc((void**)&flag, 4, hipHostMallocCoherent); *flag = 0;
  hipHostMalloc((void**)&arrived, 4, hipHostMallocCoherent); *arrived = 0;
  hipMalloc((void**)&out, N * 16);
  for (int rep = 0; rep < 3; ++rep) {
    *flag = 0; *arrived = 0; hipDeviceSynchronize();
    hipLaunchKernelGGL(probe, dim3(N), dim3(64), 0, 0, flag, out, arrived);
    int a = 0; for (long i = 0; i < 200000000L && a < N; ++i) a = *(volatile int*)arrived;   // (host-coherent memory: no stream operation behind the spinning kernel)
    *flag = 1; hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * N); hipMemcpy(h.data(), out, N * 16, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull; for (int i = 0; i < N; ++i) lo = std::min(lo, h[2 * i]);
    double mn[16], mx[16]; int cnt[16] = {0}; for (int x = 0; x < 16; ++x) { mn[x] = 1e30; mx[x] = -1; }
    for (int i = 0; i < N; ++i) { int x = (int)h[2 * i + 1]; double t = (double)(h[2 * i] - lo) * 0.01; mn[x] = std::min(mn[x], t); mx[x] = std::max(mx[x], t); ++cnt[x]; }
    printf("rep %d (arrived %d):", rep, a);
    for (int x = 0; x < 16; ++x) if (cnt[x]) printf("  xcc%d n=%d [%.2f, %.2f] us", x, cnt[x], mn[x], mx[x]);
    printf("\n"); fflush(stdout);
  }
  return 0;
}
