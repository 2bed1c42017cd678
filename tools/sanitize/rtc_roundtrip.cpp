// ASan / UBSan over the per-scene build's helper process and its file protocol (pt_rtc_core.h, pt_spec_cc.cpp): a request for a
// small kernel is written, the SANITIZED helper compiles it (hiprtc cross-compiles gfx950 without a GPU), the result is read back;
// then a request that cannot compile, and a truncated request file.
#include <cstdlib>
#include <string>

#include "../../pathtrace_amd/csrc/device/pt_rtc_core.h"

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: rtc_roundtrip HELPER WORKDIR\n"); return 2; }
    const std::string helper = argv[1], dir = argv[2];
    const std::string req = dir + "/q.bin", res = dir + "/r.bin";
    ptrtc::Request q;
    q.top = "#include \"k.h\"\n";
    q.top_name = "top.hip";
    q.headers = {{"k.h", "namespace t { template <int N> __global__ void k(float *p) { p[threadIdx.x] = (float)N; } }\n"}};
    q.exprs = {"t::k<3>", "t::k<4>"};
    q.opts = {"--offload-arch=gfx950", "-O3"};
    if (!ptrtc::write_request(req.c_str(), q)) return 1;
    if (system((helper + " " + req + " " + res).c_str()) != 0) { fprintf(stderr, "helper failed\n"); return 1; }
    ptrtc::Result r;
    if (!ptrtc::read_result(res.c_str(), r) || r.status != 0 || r.lowered.size() != 2 || r.code.size() < 1000) { fprintf(stderr, "bad result: %s\n", r.log.c_str()); return 1; }
    printf("ok: %zu bytes, %s, %s\n", r.code.size(), r.lowered[0].c_str(), r.lowered[1].c_str());
    // a compile error is a result, not a crash
    q.headers[0].second = "this is not HIP\n";
    if (!ptrtc::write_request(req.c_str(), q) || system((helper + " " + req + " " + res).c_str()) != 0) return 1;
    ptrtc::Result r2;
    if (!ptrtc::read_result(res.c_str(), r2) || r2.status == 0 || r2.log.empty()) { fprintf(stderr, "a broken program compiled?\n"); return 1; }
    printf("ok: compile error reported (%zu bytes of log)\n", r2.log.size());
    // a truncated request: exit code 2, no result needed
    if (FILE *f = fopen(req.c_str(), "wb")) { fwrite("\x07\0\0\0\0\0\0\0PTRT", 1, 12, f); fclose(f); }
    const int rc = system((helper + " " + req + " " + res).c_str());
    if (rc == 0) { fprintf(stderr, "a truncated request was accepted\n"); return 1; }
    printf("ok: truncated request refused\n");
    return 0;
}
