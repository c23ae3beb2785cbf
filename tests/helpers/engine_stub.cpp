// Link-time stand-ins for the engine entry points rtpose_main.cpp references, so that the HOST side of rtpose.bin (producer pool, shared queue,
// workers in --dry_engine mode, re-orderer, JSON writers, JPEG encoders) plus the host-only sources of the library (host_util.cpp,
// preprocess.cpp, codecs.cpp) build with g++ -fsanitize=thread / address,undefined — no HIP runtime in the process
// (tests/test_dispatch_sanitizers.py; SURVEY section 5 "race detection").  Nothing here is ever called with --dry_engine: every stub fails loudly.
#include <cstdio>
#include <cstring>

#include "../../include/rtpose_mi355x.h"

extern "C" {
int rtp_config_default(rtp_config* cfg) { if (!cfg) return RTP_EINVAL; memset(cfg, 0, sizeof *cfg); cfg->num_scales = 1; cfg->start_scale = 1.f; cfg->scale_gap = 0.3f; return RTP_OK; }
int rtp_engine_create(const rtp_config*, rtp_engine** out) { if (out) *out = nullptr; return RTP_ENODEV; }
void rtp_engine_destroy(rtp_engine*) {}
int rtp_engine_info(const rtp_engine*, int*, int*, int*, int*, int*) { return RTP_ENODEV; }
int rtp_submit(rtp_engine*, const float*, uint64_t) { return RTP_ENODEV; }
int rtp_submit_frame(rtp_engine*, const unsigned char*, int, int, uint64_t, float*) { return RTP_ENODEV; }
int rtp_collect(rtp_engine*, uint64_t*, float*, int*) { return RTP_ENODEV; }
int rtp_collect_rendered(rtp_engine*, uint64_t*, float*, int*, unsigned char*) { return RTP_ENODEV; }
int rtp_copy_weights_from(rtp_engine*, rtp_engine*) { return RTP_ENODEV; }
int rtp_device_local_cpus(int, char*, size_t) { return 0; }
int rtp_get_split_layers(const rtp_engine*, char*, size_t, int*) { return RTP_ENODEV; }
const char* rtp_last_error(const rtp_engine*) { return "engine stub (sanitizer build of the host side): no device code in this binary"; }
}
