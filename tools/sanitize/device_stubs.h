// The device entry points the host front end (pt_host.cpp) refers to, stubbed for the CPU-only sanitizer builds.
#pragma once
#include "pathtrace_hip.h"
extern "C" pt_ctx *pt_create(const pt_scene_desc *, const pt_config *) { return nullptr; }
extern "C" void pt_destroy(pt_ctx *) {}
extern "C" int pt_render_async(pt_ctx *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t) { return -1; }
extern "C" int pt_poll(pt_ctx *, uint64_t *, uint64_t *) { return -1; }
extern "C" int pt_wait(pt_ctx *) { return -1; }
extern "C" int pt_read_framebuffer(pt_ctx *, float *) { return -1; }
extern "C" int pt_snapshot_framebuffer(pt_ctx *, float *, uint64_t *) { return -1; }
extern "C" int pt_get_counters(pt_ctx *, pt_counters *) { return -1; }
extern "C" int pt_device_count(void) { return 0; }
extern "C" pt_multi *pt_multi_create(const pt_scene_desc *, const pt_config *, int32_t, const int32_t *, int32_t, int32_t) { return nullptr; }
extern "C" void pt_multi_destroy(pt_multi *) {}
extern "C" int pt_multi_render_async(pt_multi *, int32_t, int32_t) { return -1; }
extern "C" int pt_multi_poll(pt_multi *, uint64_t *, uint64_t *) { return -1; }
extern "C" int pt_multi_wait(pt_multi *) { return -1; }
extern "C" int pt_multi_read_framebuffer(pt_multi *, float *) { return -1; }
extern "C" int pt_multi_snapshot_framebuffer(pt_multi *, float *, uint64_t *) { return -1; }
extern "C" int pt_multi_get_counters(pt_multi *, pt_counters *) { return -1; }
extern "C" int pt_multi_device_count(pt_multi *) { return 0; }
extern "C" int pt_multi_get_device_counters(pt_multi *, int32_t, pt_counters *) { return -1; }
extern "C" uint64_t pt_multi_exchange_bytes(pt_multi *) { return 0; }
extern "C" int pt_spec_info(pt_ctx *, char *, size_t) { return -1; }
// ABI v6 (the launch plan): referenced by the HipWavefront mirror in pt_host.cpp
extern "C" int pt_reserve(pt_ctx *, int64_t, int32_t) { return -1; }
extern "C" int pt_prime(pt_ctx *, int32_t, const int32_t *, int32_t) { return -1; }
extern "C" int pt_get_plan(pt_ctx *, pt_plan *) { return -1; }
extern "C" double pt_render_seconds(pt_ctx *) { return -1.0; }
extern "C" int pt_wait_for(pt_ctx *, int32_t) { return -1; }
extern "C" int pt_spec_wait(pt_ctx *) { return -1; }
extern "C" int pt_multi_reserve(pt_multi *, int32_t, int32_t) { return -1; }
extern "C" double pt_multi_render_seconds(pt_multi *) { return -1.0; }
extern "C" int pt_multi_wait_for(pt_multi *, int32_t) { return -1; }
