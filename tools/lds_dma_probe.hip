// Probe: does an LDS-DMA issued from inline asm (M0 = workgroup-relative LDS address) land in the
// right workgroup's LDS when several workgroups share a CU?  Compares the builtin path and two asm
// forms (plain vaddr / saddr+voffset+inst_offset).  Build: hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ src, int* bad, int lds_words, int spin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // each block copies its own 4 KiB pattern (block-specific) into LDS at offset OFF, 4 waves x 1 KiB
  const unsigned char* g = (const unsigned char*)src + (size_t)blockIdx.x * 4096;
  const int OFF = lds_words * 4 - 4096;  // top of the allocation
  unsigned char* l = smem + OFF + wave * 1024;
  for (int it = 0; it < spin; ++it) {
    if (MODE == 0) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + wave * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    } else if (MODE == 1) {
      const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)l;
      const unsigned char* p = g + wave * 1024 + lane * 16;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(p), "s"(la) : "memory");
    } else {
      // saddr + voffset + instruction offset: lower half via offset:0, upper half via offset:512 (two dwordx2-sized pieces
      // would not be 16 B; instead issue with base-512 and offset:512 to exercise the offset path)
      const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)l - 512;
      const unsigned voff = (unsigned)(wave * 1024 + lane * 16);
      const unsigned char* gb = g - 512;  // the buffer has a 4 KiB pad in front
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:512\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(gb), "s"(la) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned* lw = (const unsigned*)(smem + OFF);
    int err = 0;
    for (int i = threadIdx.x; i < 1024; i += 256)
      if (lw[i] != src[(size_t)blockIdx.x * 1024 + i]) err++;
    if (err) atomicAdd(bad, err);
    __builtin_amdgcn_s_barrier();
    // scribble so that a stale/foreign image is detectable next iteration
    for (int i = threadIdx.x; i < 1024; i += 256) ((unsigned*)(smem + OFF))[i] = 0xdeadbeef;
    __builtin_amdgcn_s_barrier();
  }
}

int main() {
  const int blocks = 2048;
  std::vector<unsigned> h((size_t)blocks * 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
  unsigned* d0; unsigned* d; int* bad;
  (void)hipMalloc(&d0, h.size() * 4 + 4096); (void)hipMalloc(&bad, 4);
  d = d0 + 1024;
  (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int lds_kb : {16, 40, 56, 72, 104, 160}) {
    for (int mode = 0; mode < 3; ++mode) {
      (void)hipMemset(bad, 0, 4);
      const int lds = lds_kb * 1024;
      if (mode == 0) { (void)hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), lds, 0, d, bad, lds / 4, 8); }
      if (mode == 1) { (void)hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), lds, 0, d, bad, lds / 4, 8); }
      if (mode == 2) { (void)hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), lds, 0, d, bad, lds / 4, 8); }
      hipError_t e = hipDeviceSynchronize();
      int hb = -1;
      (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
      printf("lds %3d KB/block mode %d (%s): %s bad words %d\n", lds_kb, mode, mode == 0 ? "builtin" : mode == 1 ? "asm vaddr" : "asm saddr+offset",
             e == hipSuccess ? "ok" : hipGetErrorString(e), hb);
    }
  }
  return 0;
}
