// pt_spec_cc REQUEST RESULT -- the per-scene build's compiler process (pt_spec.cpp starts it, pathtrace_amd/build.py builds it).
// It loads THIS toolchain's libhiprtc (the directory the library resolved, or the one it was built against) in a process that holds
// no other ROCm, runs the one compilation the request file describes and writes the code object, the lowered kernel names
// and the log to the result file.  It never touches a GPU: hiprtc compiles for the --offload-arch it is given.
//   exit code 0: a result file was written (its status says whether the compilation succeeded); 2: bad arguments / files.
#include <cstdlib>

#include "pt_rtc_core.h"

#ifndef PT_ROCM_LIB_DIR
#define PT_ROCM_LIB_DIR "/opt/rocm/lib"
#endif

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: pt_spec_cc REQUEST RESULT\n"); return 2; }
    ptrtc::Request q;
    if (!ptrtc::read_request(argv[1], q)) { fprintf(stderr, "pt_spec_cc: cannot read %s\n", argv[1]); return 2; }
    ptrtc::Result r;
    ptrtc::Rtc rtc;
    std::vector<std::string> candidates;
    // the directory the library resolved when it started this process (pt_spec.cpp rocm_lib_dir), then the one this helper was
    // built against, then the usual place; by soname last (the environment pt_spec.cpp gives this process has LD_LIBRARY_PATH =
    // that directory, so hiprtc's own dlopen of libamd_comgr finds the same toolchain's)
    if (const char *dir = getenv("PT_SPEC_ROCM_LIB_DIR")) candidates.push_back(std::string(dir) + "/libhiprtc.so");
    candidates.push_back(PT_ROCM_LIB_DIR "/libhiprtc.so");
    candidates.push_back("/opt/rocm/lib/libhiprtc.so");
    candidates.push_back("libhiprtc.so");
    if (!rtc.load(candidates)) r.log = "pt_spec_cc: libhiprtc.so could not be loaded";
    else ptrtc::run(rtc, q, r);
    return ptrtc::write_result(argv[2], r) ? 0 : 2;
}
