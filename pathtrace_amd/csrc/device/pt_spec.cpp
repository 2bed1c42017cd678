// Per-scene build of the traversal sweep: at pt_create the scene's fast traversal program becomes a
// compile-time table (pt_context.cpp spec_header_text) and pt_kernels.hip is compiled once more, with hiprtc, for that
// table: world_hit_fast unrolls into straight-line code -- no op fetch, decode or dispatch, leaf constants as literals --
// which performs the generic sweep's arithmetic statement for statement, so a render is the same bit for bit whichever
// of the two ran.  The module holds k_extend, k_connect and k_trace for this scene; everything else (and every scene
// the build cannot serve) runs the generic kernels of the library.
//
//  * hiprtc is loaded on demand (dlopen): a machine without it, a compile error or PATHTRACE_HIP_SPEC=off leave the
//    generic kernels in place, silently -- pt_spec_status tells.
//  * The build runs on a thread of its own (PATHTRACE_HIP_SPEC=async, the default): pt_create returns at once, launches
//    switch to the module when it is ready.  PATHTRACE_HIP_SPEC=sync builds inside pt_create; pt_spec_wait blocks.
//  * Code objects are cached per process by the hash of (table, flags): a second context of the same scene builds nothing.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

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
#include <thread>
#include <vector>

#include "pt_spec.h"

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
// ---- hiprtc through dlopen ----
struct Rtc {
    void *lib = nullptr;
    int (*CreateProgram)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*CompileProgram)(void *, int, const char **) = nullptr;
    int (*AddNameExpression)(void *, const char *) = nullptr;
    int (*GetLoweredName)(void *, const char *, const char **) = nullptr;
    int (*GetCodeSize)(void *, size_t *) = nullptr;
    int (*GetCode)(void *, char *) = nullptr;
    int (*GetProgramLogSize)(void *, size_t *) = nullptr;
    int (*GetProgramLog)(void *, char *) = nullptr;
    int (*DestroyProgram)(void **) = nullptr;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
#define SYM(f, n) f = (decltype(f))dlsym(lib, n)
        SYM(CreateProgram, "hiprtcCreateProgram"); SYM(CompileProgram, "hiprtcCompileProgram");
        SYM(AddNameExpression, "hiprtcAddNameExpression"); SYM(GetLoweredName, "hiprtcGetLoweredName");
        SYM(GetCodeSize, "hiprtcGetCodeSize"); SYM(GetCode, "hiprtcGetCode");
        SYM(GetProgramLogSize, "hiprtcGetProgramLogSize"); SYM(GetProgramLog, "hiprtcGetProgramLog");
        SYM(DestroyProgram, "hiprtcDestroyProgram");
#undef SYM
        return CreateProgram && CompileProgram && AddNameExpression && GetLoweredName && GetCodeSize && GetCode && DestroyProgram;
    }
};
Rtc g_rtc;
std::mutex g_rtc_mutex;   // one compile at a time (comgr is heavy, and the cache below is filled under it)

struct CodeObject {
    std::vector<char> code;
    std::string name_extend[2], name_connect, name_trace[3];   // k_extend: [1] = the bounce-0 instantiation (forms its camera rays)
    int connect_nr = 2;
};
std::map<std::string, std::shared_ptr<CodeObject>> g_cache;   // key: table text + flags

uint64_t fnv1a(const std::string &s)
{
    uint64_t h = 1469598103934665603ull;
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
    return h;
}
}  // namespace

// One build.  `table` is the PT_SPEC_HEADER text.  Returns the code object or nullptr with `log` set.
static std::shared_ptr<CodeObject> compile(const std::string &table, bool geom_all, bool textured, int connect_nr, std::string &log)
{
    char flags[256];
    snprintf(flags, sizeof flags, "ga%d tex%d nr%d w%s", geom_all ? 1 : 0, textured ? 1 : 0, connect_nr, getenv("PATHTRACE_HIP_SPEC_WAVES") ? getenv("PATHTRACE_HIP_SPEC_WAVES") : "5");
    const std::string key = table + flags;
    std::lock_guard<std::mutex> lock(g_rtc_mutex);
    if (getenv("PATHTRACE_HIP_SPEC_BREAK")) { log = "PATHTRACE_HIP_SPEC_BREAK is set: the per-scene build fails on purpose (fallback test)"; return nullptr; }
    auto hit = g_cache.find(key);
    if (hit != g_cache.end()) return hit->second;
    if (!g_rtc.load()) { log = "libhiprtc.so could not be loaded"; return nullptr; }
    const std::string top = "#define PT_SPEC_BUILD 1\n#define PT_SPEC_HEADER \"pt_spec_table.h\"\n#include \"pt_kernels.hip\"\n";
    const char *headers[] = {kSrcKernels, kSrcDevice, kSrcFdiv, table.c_str()};
    const char *names[] = {"pt_kernels.hip", "pt_device.h", "pt_fdiv.h", "pt_spec_table.h"};
    void *prog = nullptr;
    if (g_rtc.CreateProgram(&prog, top.c_str(), "pt_spec_top.hip", 4, headers, names) != 0) { log = "hiprtcCreateProgram failed"; return nullptr; }
    auto obj = std::make_shared<CodeObject>();
    obj->connect_nr = connect_nr;
    const char *ga = geom_all ? "true" : "false", *tex = textured ? "true" : "false";
    char e_ext[2][128], e_con[128], e_tr[3][128];
    snprintf(e_ext[0], sizeof e_ext[0], "ptd::k_extend<%s, false, false>", ga);
    snprintf(e_ext[1], sizeof e_ext[1], "ptd::k_extend<%s, false, true>", ga);
    snprintf(e_con, sizeof e_con, "ptd::k_connect<%d, %s, %s, false>", connect_nr, tex, ga);
    const int trs[3] = {1, 2, 4};
    for (int i = 0; i < 3; i++) snprintf(e_tr[i], sizeof e_tr[i], "ptd::k_trace<%d, %s, false>", trs[i], ga);
    bool ok = g_rtc.AddNameExpression(prog, e_ext[0]) == 0 && g_rtc.AddNameExpression(prog, e_ext[1]) == 0 && g_rtc.AddNameExpression(prog, e_con) == 0;
    for (int i = 0; i < 3; i++) ok = ok && g_rtc.AddNameExpression(prog, e_tr[i]) == 0;
    // the product's flags (pathtrace_amd/build.py): no FMA contraction, IEEE division and square root; the specialised k_connect
    // is compiled for 5 waves per SIMD (96 VGPRs: at 6 it spills 17, measured +10 % instead of -4 %) and without the early
    // radiance request, which at this register budget is spilled the moment it arrives (k_connect 20.5 against 14.3 ms)
    const char *waves = "-DPT_CONNECT_WAVES=5";
    if (const char *e = getenv("PATHTRACE_HIP_SPEC_WAVES")) { if (!strcmp(e, "4")) waves = "-DPT_CONNECT_WAVES=4"; else if (!strcmp(e, "6")) waves = "-DPT_CONNECT_WAVES=6"; }
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", waves, "-DPT_CONNECT_PREFETCH=0"};
    for (const char *const *f = kBuildFlags; *f; f++) opts.push_back(*f);
    const int rc = ok ? g_rtc.CompileProgram(prog, (int)opts.size(), opts.data()) : -1;
    if (rc != 0) {
        size_t n = 0;
        if (g_rtc.GetProgramLogSize && g_rtc.GetProgramLogSize(prog, &n) == 0 && n > 1) { log.resize(n); g_rtc.GetProgramLog(prog, &log[0]); }
        else log = "hiprtcCompileProgram failed";
        g_rtc.DestroyProgram(&prog);
        return nullptr;
    }
    const char *lowered = nullptr;
    ok = g_rtc.GetLoweredName(prog, e_ext[0], &lowered) == 0 && lowered;
    if (ok) obj->name_extend[0] = lowered;
    ok = ok && g_rtc.GetLoweredName(prog, e_ext[1], &lowered) == 0 && lowered;
    if (ok) obj->name_extend[1] = lowered;
    ok = ok && g_rtc.GetLoweredName(prog, e_con, &lowered) == 0 && lowered;
    if (ok) obj->name_connect = lowered;
    for (int i = 0; i < 3 && ok; i++) { ok = g_rtc.GetLoweredName(prog, e_tr[i], &lowered) == 0 && lowered; if (ok) obj->name_trace[i] = lowered; }
    size_t sz = 0;
    ok = ok && g_rtc.GetCodeSize(prog, &sz) == 0 && sz > 0;
    if (ok) { obj->code.resize(sz); ok = g_rtc.GetCode(prog, obj->code.data()) == 0; }
    g_rtc.DestroyProgram(&prog);
    if (!ok) { log = "hiprtc: no code object / lowered names"; return nullptr; }
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

static int launch(hipFunction_t f, int grid, size_t lds, hipStream_t s, void **args)
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
