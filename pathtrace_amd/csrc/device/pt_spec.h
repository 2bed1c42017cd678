// Per-scene build of the traversal sweep: interface between pt_spec.cpp (hiprtc, module API) and the context / launchers.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "pt_device.h"

namespace ptd {
struct SpecJob;
// start the build of the module for `table` (the PT_SPEC_HEADER text); synchronous = on the calling thread
SpecJob *spec_start(const std::string &table, bool geom_all, bool textured, int connect_nr, int device, bool synchronous);
int spec_poll(SpecJob *j);    // 0 building, 1 module loaded (launch through it), -1 not available
int spec_wait(SpecJob *j);    // blocks until the build has ended, then as spec_poll
const char *spec_log(SpecJob *j);
std::string spec_info(SpecJob *j);   // one line of JSON: who compiled the module (helper / in-process, libhiprtc path, producer string)
void spec_destroy(SpecJob *j);
int spec_connect_nr(SpecJob *j);
int spec_launch_extend(SpecJob *j, bool b0, int grid, size_t lds, hipStream_t s, const DScene &S, const DStreams &st, const DBatch &b, int qi, int bounce);
int spec_launch_connect(SpecJob *j, int grid, size_t lds, hipStream_t s, const DScene &S, const DStreams &st, const DBatch &b, int bounce);
int spec_launch_trace(SpecJob *j, int nr, int grid, size_t lds, hipStream_t s, const DScene &S, const DStreams &st, long long n, const float *org,
                      const float *dir, uint32_t k0, uint32_t k1, uint32_t vol_dim, float *t_out, int *id_out);
long spec_build_check(const std::string &table, bool geom_all, bool textured, int connect_nr, std::string &log, std::string *info = nullptr);
}  // namespace ptd
