// Compile ONE instantiation of the ring kernel to look at its ISA / register use (seconds instead of the whole TU):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-vectorize --cuda-device-only -S -DPROBE_ARGS="_Float16,128,64,2,1,2,7,256,4,1,true,100,true" -o /tmp/p.s tools/ring_probe.hip
#define RTP_RING_NO_LAUNCHERS
#include "../caffe_rtpose_amd/csrc/conv_ring.hip"
template __global__ void rtp::conv_ring_kernel<PROBE_ARGS>(rtp::ConvParams);
