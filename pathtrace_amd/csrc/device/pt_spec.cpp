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
#define _GNU_SOURCE   // dladdr
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
#ifndef PT_ROCM_LIB_DIR
#define PT_ROCM_LIB_DIR "/opt/rocm/lib"
#endif
#ifndef PT_HIPCC_PRODUCER
#define PT_HIPCC_PRODUCER ""   // "clang version X.Y.Z" of the hipcc that built this library (pathtrace_amd/build.py), "" = unknown
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
// Process-wide state of the build.  Heap-allocated and never destroyed on purpose: an asynchronous build may still run when
// the process exits or the library is unloaded (a Renderer never closed, an exception path, interpreter shutdown), and a
// static destructor running under the worker's feet -- the cache map, the mutex it holds -- crashed at exit.
struct CodeObject {
    std::vector<char> code;
    std::string name_extend[2], name_connect, name_trace[3];   // k_extend: [1] = the bounce-0 instantiation (forms its camera rays)
    int connect_nr = 2;
    // provenance (pt_spec_info): who compiled it
    bool by_helper = false;
    std::string rtc_path, producer, note;   // note: why the helper did not compile it, when it did not
};
struct Globals {
    ptrtc::Rtc rtc;          // the in-process compiler (fallback)
    std::mutex rtc_mutex;    // one compile at a time (comgr is heavy, and the cache below is filled under it)
    std::map<std::string, std::shared_ptr<CodeObject>> cache;   // key: table text + flags
};
Globals &G() { static Globals *g = new Globals(); return *g; }

// Every environment knob of the build, read ONCE on the thread that asks for the build (spec_start / spec_build_check): the
// worker thread must not call getenv while the host thread may call setenv (Python's os.environ, a test's monkeypatch).
struct SpecEnv {
    std::string flags, brk, dump, cc, tmpdir, path;   // cc = "in-process": no helper, whatever libhiprtc the process resolves (the A/B of round 4)
    static std::string get(const char *n) { const char *v = getenv(n); return v ? v : ""; }
    static SpecEnv snapshot()
    {
        SpecEnv e;
        e.flags = get("PATHTRACE_HIP_SPEC_FLAGS"); e.brk = get("PATHTRACE_HIP_SPEC_BREAK"); e.dump = get("PATHTRACE_HIP_SPEC_DUMP");
        e.cc = get("PATHTRACE_HIP_SPEC_CC");
        e.tmpdir = get("TMPDIR"); e.path = get("PATH");
        return e;
    }
};

// The directory this library was loaded from: its compile helper (pt_spec_cc) sits beside it.
std::string library_dir()
{
    Dl_info info;
    if (!dladdr((const void *)&library_dir, &info) || !info.dli_fname) return "";
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return k == std::string::npos ? "." : p.substr(0, k);
}

// Where this toolchain's libhiprtc / libamd_comgr live, resolved when the build runs: the directory the library was built
// against as hipcc was named (normally /opt/rocm/lib: the symlink, not the versioned directory behind it) if it is there,
// else /opt/rocm/lib.
std::string rocm_lib_dir()
{
    for (const char *d : {PT_ROCM_LIB_DIR, "/opt/rocm/lib"}) {
        const std::string f = std::string(d) + "/libhiprtc.so";
        if (access(f.c_str(), R_OK) == 0) return d;
    }
    return PT_ROCM_LIB_DIR;
}

// The compilation in a process of its own.  Why: the compiler must be the toolchain this library was built with.  A host
// process may already hold ANOTHER ROCm's libhiprtc / libamd_comgr under the same sonames -- PyTorch wheels bundle theirs
// (roc-7.0 / LLVM 20 in this image, against /opt/rocm's roc-7.2 / LLVM 22) -- and dlopen("libhiprtc.so"), and hiprtc's own
// dlopen of comgr, then return THOSE: the module was being built by the older compiler, whose code for these kernels is
// different (generic k_connect: 24 spilled VGPRs and 76 B of scratch against none; 25.6 against 13.0 ms per 64 spp).  A child
// process holds nothing but what it loads itself -- and is given an environment of its own making: the host's
// LD_LIBRARY_PATH (hiprtc opens comgr by soname, and LD_LIBRARY_PATH beats RUNPATH: a torch/lib in front would bring the
// other compiler back), LD_PRELOAD and the profilers' ROCP_* / HSA_TOOLS_LIB variables (under rocprofv3 the tool library
// would load in a process that must never touch the GPU) do not reach it.  The helper never touches the GPU.
// Returns false when the helper could not be run at all (missing file, spawn failure): the caller then compiles in-process.
bool compile_in_child(const ptrtc::Request &q, ptrtc::Result &r, std::string &why, const SpecEnv &env, std::atomic<int> *child_pid,
                      const std::atomic<bool> *cancel)
{
    const std::string exe = !env.cc.empty() ? env.cc : library_dir() + "/pt_spec_cc";
    if (access(exe.c_str(), X_OK) != 0) { why = exe + " is not there"; return false; }
    const std::string tmp = env.tmpdir.empty() ? "/tmp" : env.tmpdir;
    char req[512], res[512];
    snprintf(req, sizeof req, "%s/pt_spec_req_XXXXXX", tmp.c_str());
    snprintf(res, sizeof res, "%s/pt_spec_res_XXXXXX", tmp.c_str());
    const int fq = mkstemp(req), fr = mkstemp(res);
    if (fq >= 0) close(fq);
    if (fr >= 0) close(fr);
    bool ran = false;
    if (fq >= 0 && fr >= 0 && ptrtc::write_request(req, q)) {
        char *const argv[] = {(char *)exe.c_str(), req, res, nullptr};
        const std::string libdir = rocm_lib_dir();
        std::vector<std::string> ev = {"PATH=" + (env.path.empty() ? std::string("/usr/bin:/bin") : env.path), "TMPDIR=" + tmp,
                                       "LD_LIBRARY_PATH=" + libdir, "PT_SPEC_ROCM_LIB_DIR=" + libdir};
        std::vector<char *> envp;
        for (auto &e : ev) envp.push_back(&e[0]);
        envp.push_back(nullptr);
        pid_t pid = 0;
        if (cancel && cancel->load()) { why = "the context was destroyed before its module was built"; ran = true; r.status = -1; r.log = why; }
        else if (posix_spawn(&pid, exe.c_str(), nullptr, nullptr, argv, envp.data()) == 0) {
            if (child_pid) child_pid->store((int)pid);
            // wait for it, at most two minutes (a compilation takes 1.5 - 3 s): a helper that hangs is killed, not waited for;
            // a context destroyed meanwhile (cancel) kills it too
            int status = 0, w = 0;
            bool gone = false, cancelled = false;
            for (int tick = 0; tick < 6000 && !gone; tick++) {
                if (cancel && cancel->load()) { cancelled = true; break; }
                w = waitpid(pid, &status, WNOHANG);
                if (w == pid) gone = true;
                else if (w < 0 && errno != EINTR) {   // a host that ignores SIGCHLD reaps children itself (ECHILD): watch the pid
                    if (kill(pid, 0) != 0) { gone = true; status = 0; }
                    else usleep(20000);
                } else usleep(20000);
            }
            if (child_pid) child_pid->store(0);
            if (!gone) {
                kill(pid, SIGKILL);
                while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
                why = cancelled ? "the context was destroyed while its module was being built" : "pt_spec_cc did not finish in two minutes";
                if (cancelled) ran = true, r.status = -1, r.log = why;   // not a reason to compile in-process
            }
            else if (WIFEXITED(status) && WEXITSTATUS(status) == 0 && ptrtc::read_result(res, r)) ran = true;
            else why = "pt_spec_cc ended abnormally";
        } else why = "posix_spawn(pt_spec_cc) failed";
    } else why = "no temporary file for the compile request";
    if (fq >= 0) unlink(req);
    if (fr >= 0) unlink(res);
    return ran;
}

// the producer string of a code object: the "... clang version X.Y.Z (...)" text of its .comment section
std::string producer_of(const std::vector<char> &code)
{
    static const char key[] = "clang version ";
    const size_t kl = sizeof key - 1;
    for (size_t i = 0; i + kl < code.size(); i++) {
        if (memcmp(&code[i], key, kl) != 0) continue;
        size_t a = i, b = i;
        while (a > 0 && code[a - 1] >= 32 && code[a - 1] < 127) a--;
        while (b < code.size() && code[b] >= 32 && code[b] < 127) b++;
        return std::string(&code[a], b - a);
    }
    return "";
}
}  // namespace

// One build.  `table` is the PT_SPEC_HEADER text.  Returns the code object or nullptr with `log` set.
static std::shared_ptr<CodeObject> compile(const std::string &table, bool geom_all, bool textured, int connect_nr, std::string &log, const SpecEnv &env,
                                           std::atomic<int> *child_pid = nullptr, const std::atomic<bool> *cancel = nullptr)
{
    char flags[256];
    snprintf(flags, sizeof flags, "ga%d tex%d nr%d", geom_all ? 1 : 0, textured ? 1 : 0, connect_nr);
    const std::string key = table + flags + env.flags + "|" + env.cc;
    Globals &g = G();
    std::lock_guard<std::mutex> lock(g.rtc_mutex);
    // a context destroyed while its job waited for the compiler: nothing to build
    if (cancel && cancel->load()) { log = "the context was destroyed before its module was built"; return nullptr; }
    if (!env.brk.empty()) { log = "PATHTRACE_HIP_SPEC_BREAK is set: the per-scene build fails on purpose (fallback test)"; return nullptr; }
    auto hit = g.cache.find(key);
    if (hit != g.cache.end()) return hit->second;
    ptrtc::Request q;
    q.top = "#define PT_SPEC_BUILD 1\n#define PT_SPEC_HEADER \"pt_spec_table.h\"\n#include \"pt_kernels.hip\"\n";
    q.top_name = "pt_spec_top.hip";
    q.headers = {{"pt_kernels.hip", kSrcKernels}, {"pt_device.h", kSrcDevice}, {"pt_fdiv.h", kSrcFdiv}, {"pt_spec_table.h", table}};
    const char *ga = geom_all ? "true" : "false", *tex = textured ? "true" : "false";
    char e[128];
    snprintf(e, sizeof e, "ptd::k_extend<%s, false, false>", ga); q.exprs.push_back(e);
    snprintf(e, sizeof e, "ptd::k_extend<%s, false, true>", ga); q.exprs.push_back(e);
    snprintf(e, sizeof e, "ptd::k_connect<%d, %s, %s, false>", connect_nr, tex, ga); q.exprs.push_back(e);
    for (int nr : {1, 2, 4}) { snprintf(e, sizeof e, "ptd::k_trace<%d, %s, false>", nr, ga); q.exprs.push_back(e); }
    // the product's flags (pathtrace_amd/build.py): no FMA contraction, IEEE division and square root; the specialised k_connect
    // is compiled for AT LEAST 5 waves per SIMD (it reaches 6, see PT_CONNECT_NOHOIST below) and without the early radiance request
    const char *waves = "-DPT_CONNECT_WAVES=5";   // (4 and 6 measured in round 3: PATHTRACE_HIP_SPEC_FLAGS carries such options for an A/B)
    // -pragma-unroll-threshold: the sweep's op loop must unroll COMPLETELY for the table to fold into the code (kind, shapes and
    // constants are only compile-time values per unrolled iteration).  LLVM sizes the unrolled loop before it folds the per-kind
    // dispatch away -- every body of every leaf kind times PT_SPEC_N -- and past 16384 it quietly keeps a run-time loop that
    // fetches the table from constant memory and carries all bodies (seen on the volume scene: k_connect 18.7 against 16.2 ms).
    q.opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"};
    // (the option may occur once: PATHTRACE_HIP_SPEC_FLAGS that bring their own replace it -- the negative control of
    // tests/test_spec_build.py builds with a low threshold to see the check for a surviving table trip)
    if (env.flags.find("pragma-unroll-threshold") == std::string::npos) { q.opts.push_back("-mllvm"); q.opts.push_back("-pragma-unroll-threshold=4000000"); }
    if (env.flags.find("PT_CONNECT_WAVES") == std::string::npos) q.opts.push_back(waves);
    if (env.flags.find("PT_CONNECT_PREFETCH") == std::string::npos) q.opts.push_back("-DPT_CONNECT_PREFETCH=0");
    // Round 5: the origin's per-leaf terms (bound - origin for every face of the unrolled program: ~30 VGPRs) are NOT kept across a hit's
    // groups of rays but formed again per group (+80 instructions per group): 96 VGPRs with 3 spilled -> 79 with none, six waves per
    // SIMD instead of five, k_connect 11.5 -> 11.0 ms per 64 spp (+1.3 %); with sphere / medium leaves 128 VGPRs -> 96, five waves
    // instead of four: with_volume +2.1 % (profiles/experiments/r05_ab_runs.json r05_nohoist*)
    if (env.flags.find("PT_CONNECT_NOHOIST") == std::string::npos) q.opts.push_back("-DPT_CONNECT_NOHOIST=1");
    if (env.flags.find("PT_CONNECT_WAVES_GA") == std::string::npos) q.opts.push_back("-DPT_CONNECT_WAVES_GA=5");
    for (const char *const *f = kBuildFlags; *f; f++) q.opts.push_back(*f);
    const std::string &extra = env.flags;   // measurement: more compiler options, space separated
    for (size_t i = 0; i < extra.size();) {
        const size_t j = extra.find(' ', i);
        if (j != i) q.opts.push_back(extra.substr(i, j == std::string::npos ? j : j - i));
        if (j == std::string::npos) break;
        i = j + 1;
    }
    ptrtc::Result r;
    std::string why;
    bool by_helper = true;
    // PATHTRACE_HIP_SPEC_CC=in-process (the A/B): compile in-process with whatever libhiprtc the process resolves
    if (env.cc == "in-process" || !compile_in_child(q, r, why, env, child_pid, cancel)) {
        by_helper = false;
        const std::string dir = rocm_lib_dir();
        if (!g.rtc.load({dir + "/libhiprtc.so", "libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"})) { log = "no compiler: " + (why.empty() ? std::string() : why + "; ") + "libhiprtc.so could not be loaded"; return nullptr; }
        ptrtc::run(g.rtc, q, r);
    }
    if (r.status != 0 || r.lowered.size() != 6 || r.code.empty()) { log = r.log.empty() ? "the per-scene build produced no code object" : r.log; return nullptr; }
    auto obj = std::make_shared<CodeObject>();
    obj->connect_nr = connect_nr;
    obj->name_extend[0] = r.lowered[0]; obj->name_extend[1] = r.lowered[1]; obj->name_connect = r.lowered[2];
    for (int i = 0; i < 3; i++) obj->name_trace[i] = r.lowered[3 + i];
    obj->code = std::move(r.code);
    obj->by_helper = by_helper;
    obj->rtc_path = r.rtc_path;
    obj->producer = producer_of(obj->code);
    if (!by_helper && !why.empty()) obj->note = "compiled in-process: " + why;
    if (!env.dump.empty()) {   // the code object, for llvm-objdump / tools/isa_stats.py
        if (FILE *fh = fopen(env.dump.c_str(), "wb")) { fwrite(obj->code.data(), 1, obj->code.size(), fh); fclose(fh); }
    }
    g.cache[key] = obj;
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
    SpecEnv env;                 // the build's environment knobs as the creating thread saw them
    std::atomic<int> child_pid{0};
    std::atomic<bool> cancel{false};   // spec_destroy: the context is going away, a compile helper still running is killed
};

SpecJob *spec_start(const std::string &table, bool geom_all, bool textured, int connect_nr, int device, bool synchronous)
{
    SpecJob *j = new SpecJob();
    j->table = table; j->geom_all = geom_all; j->textured = textured; j->connect_nr = connect_nr; j->device = device;
    j->env = SpecEnv::snapshot();
    auto work = [j]() {
        std::string log;
        auto obj = compile(j->table, j->geom_all, j->textured, j->connect_nr, log, j->env, &j->child_pid, &j->cancel);
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

// Who compiled a module (pt_spec_info): one line of JSON.  own_compiler: the code object's producer string is the hipcc's
// that built this library (PT_HIPCC_PRODUCER, recorded by build.py) -- false means a FOREIGN compiler built the kernels the
// context launches (measured: PyTorch's bundled clang 20 makes k_connect 2x slower), null = unknown.
static std::string json_escape(const std::string &t)
{
    std::string o;
    // ASCII only: a compiler log cut at 300 bytes may end inside a UTF-8 sequence (its quotes are U+2018 / U+2019), and the line
    // must decode whatever it holds
    for (char ch : t) { if (ch == '"' || ch == '\\') o += '\\'; if ((unsigned char)ch >= 32 && (unsigned char)ch < 127) o += ch; else if ((unsigned char)ch >= 127) o += '?'; }
    return o;
}
static std::string info_of(const CodeObject *obj, int status, const std::string &note)
{
    if (!obj) return "{\"status\": " + std::to_string(status) + ", \"built_by\": null" + (note.empty() ? "" : ", \"note\": \"" + json_escape(note.substr(0, 300)) + "\"") + "}";
    const std::string own = PT_HIPCC_PRODUCER;
    std::string o = "{\"status\": " + std::to_string(status) + ", \"built_by\": \"" + (obj->by_helper ? "helper" : "in-process") + "\", \"rtc_lib\": \"" + json_escape(obj->rtc_path) +
                    "\", \"producer\": \"" + json_escape(obj->producer) + "\", \"library_producer\": \"" + json_escape(own) + "\", \"own_compiler\": ";
    o += own.empty() || obj->producer.empty() ? "null" : (obj->producer.find(own) != std::string::npos ? "true" : "false");
    const std::string all = obj->note.empty() ? note : (note.empty() ? obj->note : obj->note + "; " + note);
    if (!all.empty()) o += ", \"note\": \"" + json_escape(all.substr(0, 300)) + "\"";
    return o + "}";
}
std::string spec_info(SpecJob *j)
{
    if (!j) return info_of(nullptr, -1, "no per-scene build");
    std::lock_guard<std::mutex> lock(j->m);
    if (!j->done) return info_of(nullptr, 0, "");
    return info_of(j->obj.get(), j->obj ? j->state.load() : -1, j->log);
}

void spec_destroy(SpecJob *j)
{
    if (!j) return;
    j->cancel.store(true);                        // a compile helper still running is killed by its waiter: a short-lived context
    if (j->worker.joinable()) j->worker.join();   // does not sit out the compilation; the thread itself is joined (it must not outlive the job)
    if (j->module) (void)hipModuleUnload(j->module);
    delete j;
}

static int launch(hipFunction_t f, int grid, size_t lds, hipStream_t s, void **args)
{
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
long spec_build_check(const std::string &table, bool geom_all, bool textured, int connect_nr, std::string &log, std::string *info)
{
    auto obj = compile(table, geom_all, textured, connect_nr, log, SpecEnv::snapshot());
    if (info) *info = info_of(obj.get(), obj ? 1 : -1, log);
    return obj ? (long)obj->code.size() : -1;
}

}  // namespace ptd
