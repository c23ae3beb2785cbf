// Which way does v_mov_b32_dpp wave_shl:1 move data on gfx950?  (conv_ring.hip's interleaved A fragments need lane l <- lane l+1.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_probe tools/dpp_probe.hip && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int l = threadIdx.x;
  out[l] = __builtin_amdgcn_update_dpp(0, 100 + l, 0x130, 0xf, 0xf, true);
}
int main() {
  int* d; int h[64];
  hipMalloc((void**)&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("wave_shl:1  lane0<-%d lane1<-%d lane15<-%d lane16<-%d lane31<-%d lane32<-%d lane62<-%d lane63<-%d\n", h[0] - 100, h[1] - 100, h[15] - 100, h[16] - 100, h[31] - 100, h[32] - 100, h[62] - 100, h[63] - 100);
  bool ok = true;
  for (int l = 0; l < 63; ++l) ok &= h[l] == 100 + l + 1;
  printf("%s\n", ok ? "OK: lane l takes lane l+1 (lane 63 -> 0 with bound_ctrl)" : "DIFFERENT direction/semantics");
  return ok ? 0 : 1;
}
