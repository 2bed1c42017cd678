// Per-scene build of the traversal sweep: at pt_create the scene's fast traversal program becomes a
// compile-time table (pt_context.cpp spec_header_text) and pt_kernels.hip is compiled once more, with hiprtc, for that
// table: world_hit_fast unrolls into straight-line code -- no op fetch, decode or dispatch, leaf constants as literals --
// which performs the generic sweep's arithmetic statement for statement, so a render is the same bit for bit whichever
// of the two ran.  The module holds k_extend, k_connect and k_trace for this scene; everything else (and every scene
// the build cannot serve) runs the generic kernels of the library.
//
//  * The compilation runs in a helper process (pt_spec_cc beside the library, see compile_in_child) that loads this
//    toolchain's hiprtc; without the helper, hiprtc is loaded in-process (dlopen).  A machine without either, a compile
//    error or PATHTRACE_HIP_SPEC=off leave the generic kernels in place, silently -- pt_spec_status tells.
//  * The build runs on a thread of its own (PATHTRACE_HIP_SPEC=async, the default): pt_create returns at once, launches
//    switch to the module when it is ready.  PATHTRACE_HIP_SPEC=sync builds inside pt_create; pt_spec_wait blocks.
//  * Code objects are cached per process by the hash of (table, flags): a second context of the same scene builds nothing.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#ifndef _GNU_SOURCE
#define _GNU_SOURCE   // dlmopen
#endif
#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include <errno.h>
#include <signal.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include "pt_rtc_core.h"
#include "pt_spec.h"
extern char **environ;
#ifndef PT_ROCM_LIB_DIR
#define PT_ROCM_LIB_DIR "/opt/rocm/lib"
#endif

namespace ptd {

// the device sources, embedded at build time (pathtrace_amd/build.py writes pt_kernel_src.inc from the files themselves)
static const char kSrcKernels[] =
#include "pt_kernel_src_kernels.inc"
    ;
static const char kSrcDevice[] =
#include "pt_kernel_src_device.inc"
    ;
static const char kSrcFdiv[] =
#include "pt_kernel_src_fdiv.inc"
    ;
// -D flags this library was built with beyond the defaults (A/B variants): the module must be built from the same kernels
static const char *const kBuildFlags[] = {
#include "pt_kernel_src_flags.inc"
};

namespace {
ptrtc::Rtc g_rtc;          // the in-process compiler (fallback)
std::mutex g_rtc_mutex;    // one compile at a time (comgr is heavy, and the cache below is filled under it)

struct CodeObject {
    std::vector<char> code;
    std::string name_extend[2], name_connect, name_trace[3];   // k_extend: [1] = the bounce-0 instantiation (forms its camera rays)
    int connect_nr = 2;
};
std::map<std::string, std::shared_ptr<CodeObject>> g_cache;   // key: table text + flags

// The directory this library was loaded from: its compile helper (pt_spec_cc) sits beside it.
std::string library_dir()
{
    Dl_info info;
    if (!dladdr((const void *)&library_dir, &info) || !info.dli_fname) return "";
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return k == std::string::npos ? "." : p.substr(0, k);
}

// The compilation in a process of its own.  Why: the compiler must be the toolchain this library was built with.  A host
// process may already hold ANOTHER ROCm's libhiprtc / libamd_comgr under the same sonames -- PyTorch wheels bundle theirs
// (roc-7.0 / LLVM 20 in this image, against /opt/rocm's roc-7.2 / LLVM 22) -- and dlopen("libhiprtc.so"), and hiprtc's own
// dlopen of comgr, then return THOSE: the module was being built by the older compiler, whose code for these kernels is
// different (generic k_connect: 24 spilled VGPRs and 76 B of scratch against none; 25.6 against 13.0 ms per 64 spp).  A child
// process holds nothing but what it loads itself.  (A second link-map namespace, dlmopen, does the same in-process and was
// tried first: it crashed after some twenty compilations beside a live HIP runtime.)  The helper never touches the GPU.
// Returns false when the helper could not be run at all (missing file, spawn failure): the caller then compiles in-process.
bool compile_in_child(const ptrtc::Request &q, ptrtc::Result &r, std::string &why)
{
    const char *forced = getenv("PATHTRACE_HIP_SPEC_CC");
    const std::string exe = forced ? std::string(forced) : library_dir() + "/pt_spec_cc";
    if (access(exe.c_str(), X_OK) != 0) { why = exe + " is not there"; return false; }
    const char *tmp = getenv("TMPDIR");
    char req[512], res[512];
    snprintf(req, sizeof req, "%s/pt_spec_req_XXXXXX", tmp && *tmp ? tmp : "/tmp");
    snprintf(res, sizeof res, "%s/pt_spec_res_XXXXXX", tmp && *tmp ? tmp : "/tmp");
    const int fq = mkstemp(req), fr = mkstemp(res);
    if (fq >= 0) close(fq);
    if (fr >= 0) close(fr);
    bool ran = false;
    if (fq >= 0 && fr >= 0 && ptrtc::write_request(req, q)) {
        char *const argv[] = {(char *)exe.c_str(), req, res, nullptr};
        pid_t pid = 0;
        if (posix_spawn(&pid, exe.c_str(), nullptr, nullptr, argv, environ) == 0) {
            // wait for it, at most two minutes (a compilation takes 1.5 - 3 s): a helper that hangs is killed, not waited for
            int status = 0, w = 0;
            bool gone = false;
            for (int tick = 0; tick < 6000 && !gone; tick++) {
                w = waitpid(pid, &status, WNOHANG);
                if (w == pid) gone = true;
                else if (w < 0 && errno != EINTR) {   // a host that ignores SIGCHLD reaps children itself (ECHILD): watch the pid
                    if (kill(pid, 0) != 0) { gone = true; status = 0; }
                    else usleep(20000);
                } else usleep(20000);
            }
            if (!gone) { kill(pid, SIGKILL); while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {} why = "pt_spec_cc did not finish in two minutes"; }
            else if (WIFEXITED(status) && WEXITSTATUS(status) == 0 && ptrtc::read_result(res, r)) ran = true;
            else why = "pt_spec_cc ended abnormally";
        } else why = "posix_spawn(pt_spec_cc) failed";
    } else why = "no temporary file for the compile request";
    if (fq >= 0) unlink(req);
    if (fr >= 0) unlink(res);
    return ran;
}
}  // namespace

// One build.  `table` is the PT_SPEC_HEADER text.  Returns the code object or nullptr with `log` set.
static std::shared_ptr<CodeObject> compile(const std::string &table, bool geom_all, bool textured, int connect_nr, std::string &log)
{
    char flags[256];
    snprintf(flags, sizeof flags, "ga%d tex%d nr%d w%s g%d", geom_all ? 1 : 0, textured ? 1 : 0, connect_nr, getenv("PATHTRACE_HIP_SPEC_WAVES") ? getenv("PATHTRACE_HIP_SPEC_WAVES") : "5",
             getenv("PATHTRACE_HIP_SPEC_GENERIC") ? 1 : 0);
    const std::string key = table + flags + (getenv("PATHTRACE_HIP_SPEC_FLAGS") ? getenv("PATHTRACE_HIP_SPEC_FLAGS") : "") + (getenv("PATHTRACE_HIP_SPEC_PF") ? getenv("PATHTRACE_HIP_SPEC_PF") : "");
    std::lock_guard<std::mutex> lock(g_rtc_mutex);
    if (getenv("PATHTRACE_HIP_SPEC_BREAK")) { log = "PATHTRACE_HIP_SPEC_BREAK is set: the per-scene build fails on purpose (fallback test)"; return nullptr; }
    auto hit = g_cache.find(key);
    if (hit != g_cache.end()) return hit->second;
    ptrtc::Request q;
    // PATHTRACE_HIP_SPEC_GENERIC (measurement): the module holds the GENERIC kernels -- the library's own code through the module path
    q.top = getenv("PATHTRACE_HIP_SPEC_GENERIC") ? "#define PT_SPEC_BUILD 1\n#include \"pt_kernels.hip\"\n"
                                                 : "#define PT_SPEC_BUILD 1\n#define PT_SPEC_HEADER \"pt_spec_table.h\"\n#include \"pt_kernels.hip\"\n";
    q.top_name = "pt_spec_top.hip";
    q.headers = {{"pt_kernels.hip", kSrcKernels}, {"pt_device.h", kSrcDevice}, {"pt_fdiv.h", kSrcFdiv}, {"pt_spec_table.h", table}};
    const char *ga = geom_all ? "true" : "false", *tex = textured ? "true" : "false";
    char e[128];
    snprintf(e, sizeof e, "ptd::k_extend<%s, false, false>", ga); q.exprs.push_back(e);
    snprintf(e, sizeof e, "ptd::k_extend<%s, false, true>", ga); q.exprs.push_back(e);
    snprintf(e, sizeof e, "ptd::k_connect<%d, %s, %s, false>", connect_nr, tex, ga); q.exprs.push_back(e);
    for (int nr : {1, 2, 4}) { snprintf(e, sizeof e, "ptd::k_trace<%d, %s, false>", nr, ga); q.exprs.push_back(e); }
    // the product's flags (pathtrace_amd/build.py): no FMA contraction, IEEE division and square root; the specialised k_connect
    // is compiled for 5 waves per SIMD (96 VGPRs: at 6 it spills 17) and without the early radiance request, which at this
    // register budget is spilled the moment it arrives
    const char *waves = "-DPT_CONNECT_WAVES=5";
    if (const char *w = getenv("PATHTRACE_HIP_SPEC_WAVES")) { if (!strcmp(w, "4")) waves = "-DPT_CONNECT_WAVES=4"; else if (!strcmp(w, "6")) waves = "-DPT_CONNECT_WAVES=6"; }
    q.opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"};
    if (!getenv("PATHTRACE_HIP_SPEC_GENERIC")) { q.opts.push_back(waves); const char *pf = getenv("PATHTRACE_HIP_SPEC_PF"); q.opts.push_back(pf && pf[0] == '1' ? "-DPT_CONNECT_PREFETCH=1" : (pf && pf[0] == '2' ? "-DPT_CONNECT_PREFETCH=2" : "-DPT_CONNECT_PREFETCH=0")); }
    for (const char *const *f = kBuildFlags; *f; f++) q.opts.push_back(*f);
    std::string extra = getenv("PATHTRACE_HIP_SPEC_FLAGS") ? getenv("PATHTRACE_HIP_SPEC_FLAGS") : "";   // measurement: more compiler options, space separated
    for (size_t i = 0; i < extra.size();) {
        const size_t j = extra.find(' ', i);
        if (j != i) q.opts.push_back(extra.substr(i, j == std::string::npos ? j : j - i));
        if (j == std::string::npos) break;
        i = j + 1;
    }
    ptrtc::Result r;
    std::string why;
    // PATHTRACE_HIP_RTC_SHARED=1 (the A/B): compile in-process with whatever libhiprtc the process resolves
    if (getenv("PATHTRACE_HIP_RTC_SHARED") || !compile_in_child(q, r, why)) {
        if (!g_rtc.load({"libhiprtc.so", "libhiprtc.so.7", PT_ROCM_LIB_DIR "/libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"})) { log = "no compiler: " + (why.empty() ? std::string() : why + "; ") + "libhiprtc.so could not be loaded"; return nullptr; }
        ptrtc::run(g_rtc, q, r);
    }
    if (r.status != 0 || r.lowered.size() != 6 || r.code.empty()) { log = r.log.empty() ? "the per-scene build produced no code object" : r.log; return nullptr; }
    auto obj = std::make_shared<CodeObject>();
    obj->connect_nr = connect_nr;
    obj->name_extend[0] = r.lowered[0]; obj->name_extend[1] = r.lowered[1]; obj->name_connect = r.lowered[2];
    for (int i = 0; i < 3; i++) obj->name_trace[i] = r.lowered[3 + i];
    obj->code = std::move(r.code);
    if (const char *dump = getenv("PATHTRACE_HIP_SPEC_DUMP")) {   // the code object, for llvm-objdump / tools/isa_stats.py
        if (FILE *fh = fopen(dump, "wb")) { fwrite(obj->code.data(), 1, obj->code.size(), fh); fclose(fh); }
    }
    g_cache[key] = obj;
    return obj;
}

struct SpecJob {
    std::string table;
    bool geom_all = false, textured = false;
    int connect_nr = 2, device = 0;
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    std::shared_ptr<CodeObject> obj;
    std::string log;
    // loaded on the context's device by whoever first sees `done` (hipModuleLoadData needs the device current)
    hipModule_t module = nullptr;
    hipFunction_t f_extend[2] = {nullptr, nullptr}, f_connect = nullptr, f_trace[3] = {nullptr, nullptr, nullptr};
    std::atomic<int> state{0};   // 0 building, 1 module loaded, -1 failed
};

SpecJob *spec_start(const std::string &table, bool geom_all, bool textured, int connect_nr, int device, bool synchronous)
{
    SpecJob *j = new SpecJob();
    j->table = table; j->geom_all = geom_all; j->textured = textured; j->connect_nr = connect_nr; j->device = device;
    auto work = [j]() {
        std::string log;
        auto obj = compile(j->table, j->geom_all, j->textured, j->connect_nr, log);
        std::lock_guard<std::mutex> lock(j->m);
        j->obj = obj; j->log = log; j->done = true;
        j->cv.notify_all();
    };
    if (synchronous) work();
    else j->worker = std::thread(work);
    return j;
}

// the build has ended: load the module once.  Called from the context's host thread with its device current.
static void finish(SpecJob *j)
{
    if (j->state.load() != 0) return;
    if (!j->obj) { j->state.store(-1); return; }
    bool ok = hipModuleLoadData(&j->module, j->obj->code.data()) == hipSuccess;
    for (int i = 0; i < 2; i++) ok = ok && hipModuleGetFunction(&j->f_extend[i], j->module, j->obj->name_extend[i].c_str()) == hipSuccess;
    ok = ok && hipModuleGetFunction(&j->f_connect, j->module, j->obj->name_connect.c_str()) == hipSuccess;
    for (int i = 0; i < 3; i++) ok = ok && hipModuleGetFunction(&j->f_trace[i], j->module, j->obj->name_trace[i].c_str()) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); j->log = "hipModuleLoadData / hipModuleGetFunction failed for the per-scene module"; }
    j->state.store(ok ? 1 : -1);
}

int spec_poll(SpecJob *j)
{
    if (!j) return -1;
    if (j->state.load() != 0) return j->state.load();
    {
        std::lock_guard<std::mutex> lock(j->m);
        if (!j->done) return 0;
    }
    finish(j);
    return j->state.load();
}

int spec_wait(SpecJob *j)
{
    if (!j) return -1;
    {
        std::unique_lock<std::mutex> lock(j->m);
        j->cv.wait(lock, [j] { return j->done; });
    }
    finish(j);
    return j->state.load();
}

const char *spec_log(SpecJob *j) { return j ? j->log.c_str() : ""; }

void spec_destroy(SpecJob *j)
{
    if (!j) return;
    if (j->worker.joinable()) j->worker.join();   // a build in flight is waited for: its thread must not outlive the library
    if (j->module) (void)hipModuleUnload(j->module);
    delete j;
}

static int launch_(hipFunction_t f, int grid, size_t lds, hipStream_t s, void **args);
static int launch(hipFunction_t f, int grid, size_t lds, hipStream_t s, void **args)
{
    static const bool timed = getenv("PATHTRACE_HIP_SPEC_TIME") != nullptr;   // measurement: host time spent inside the module launch call
    if (!timed) return launch_(f, grid, lds, s, args);
    static double total_us = 0.0, max_us = 0.0;
    static long calls = 0;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = launch_(f, grid, lds, s, args);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    total_us += us; calls++; if (us > max_us) max_us = us;
    if (calls % 200 == 0) fprintf(stderr, "[pt spec] %ld module launches, mean %.1f us, max %.1f us inside the call\n", calls, total_us / calls, max_us);
    return rc;
}
static int launch_(hipFunction_t f, int grid, size_t lds, hipStream_t s, void **args)
{
    static const char *how = getenv("PATHTRACE_HIP_SPEC_LAUNCH");
    if (how && !strcmp(how, "ext"))
        return hipExtModuleLaunchKernel(f, (unsigned)grid * 256u, 1, 1, 256, 1, 1, lds, s, args, nullptr, nullptr, nullptr, 0) == hipSuccess ? 0 : -1;
    return hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, 256, 1, 1, (unsigned)lds, s, args, nullptr) == hipSuccess ? 0 : -1;
}

// the module's kernels, launched with the generic kernels' argument lists (pt_kernels.hip k_extend / k_connect / k_trace)
int spec_launch_extend(SpecJob *j, bool b0, int grid, size_t lds, hipStream_t s, const DScene &S, const DStreams &st, const DBatch &b, int qi, int bounce)
{
    void *args[] = {(void *)&S, (void *)&S.ops, (void *)&S.insts, (void *)&S.prims, (void *)&S.mats, (void *)&S.lights, (void *)&S.emit,
                    (void *)&st, (void *)&b, (void *)&qi, (void *)&bounce};
    return launch(j->f_extend[b0 ? 1 : 0], grid, lds, s, args);
}
int spec_launch_connect(SpecJob *j, int grid, size_t lds, hipStream_t s, const DScene &S, const DStreams &st, const DBatch &b, int bounce)
{
    void *args[] = {(void *)&S, (void *)&S.ops, (void *)&S.insts, (void *)&S.prims, (void *)&S.mats, (void *)&S.lights, (void *)&S.emit,
                    (void *)&st, (void *)&b, (void *)&bounce};
    return launch(j->f_connect, grid, lds, s, args);
}
int spec_launch_trace(SpecJob *j, int nr, int grid, size_t lds, hipStream_t s, const DScene &S, const DStreams &st, long long n, const float *org,
                      const float *dir, uint32_t k0, uint32_t k1, uint32_t vol_dim, float *t_out, int *id_out)
{
    const int i = nr == 4 ? 2 : (nr == 2 ? 1 : 0);
    void *args[] = {(void *)&S, (void *)&S.ops, (void *)&st, (void *)&n, (void *)&org, (void *)&dir, (void *)&k0, (void *)&k1, (void *)&vol_dim,
                    (void *)&t_out, (void *)&id_out};
    return launch(j->f_trace[i], grid, lds, s, args);
}
int spec_connect_nr(SpecJob *j) { return j ? j->connect_nr : 0; }

// host-only check (no device): build the module of a table and report the size of its code object, < 0 with the log on failure
long spec_build_check(const std::string &table, bool geom_all, bool textured, int connect_nr, std::string &log)
{
    auto obj = compile(table, geom_all, textured, connect_nr, log);
    return obj ? (long)obj->code.size() : -1;
}

}  // namespace ptd
