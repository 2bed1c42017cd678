// One hiprtc compilation, as a value: request (top source, headers, name expressions, options) -> result (code object, lowered
// names, log).  Shared by the library (pt_spec.cpp) and by its compile helper (pt_spec_cc.cpp), which runs the same request in
// a process of its own so that the compiler is this toolchain's (see pt_spec.cpp).  Host C++ only; no HIP headers.
#pragma once
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace ptrtc {

struct Rtc {   // hiprtc through dlopen
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
    bool load(const std::vector<std::string> &candidates)
    {
        if (lib) return true;
        for (const std::string &name : candidates) {
            lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
#define PTRTC_SYM(f, n) f = (decltype(f))dlsym(lib, n)
        PTRTC_SYM(CreateProgram, "hiprtcCreateProgram"); PTRTC_SYM(CompileProgram, "hiprtcCompileProgram");
        PTRTC_SYM(AddNameExpression, "hiprtcAddNameExpression"); PTRTC_SYM(GetLoweredName, "hiprtcGetLoweredName");
        PTRTC_SYM(GetCodeSize, "hiprtcGetCodeSize"); PTRTC_SYM(GetCode, "hiprtcGetCode");
        PTRTC_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize"); PTRTC_SYM(GetProgramLog, "hiprtcGetProgramLog");
        PTRTC_SYM(DestroyProgram, "hiprtcDestroyProgram");
#undef PTRTC_SYM
        return CreateProgram && CompileProgram && AddNameExpression && GetLoweredName && GetCodeSize && GetCode && DestroyProgram;
    }
};

struct Request {
    std::string top, top_name;
    std::vector<std::pair<std::string, std::string>> headers;   // (include name, text)
    std::vector<std::string> exprs, opts;
};
struct Result {
    int status = -1;                    // 0: code and lowered names are valid
    std::string log;
    std::string rtc_path;               // the libhiprtc file that compiled (dladdr of hiprtcCompileProgram), "" when unknown
    std::vector<std::string> lowered;   // one per name expression
    std::vector<char> code;
};

inline void run(Rtc &rtc, const Request &q, Result &r)
{
    r = Result();
    std::vector<const char *> htext, hname, opts;
    for (const auto &h : q.headers) { hname.push_back(h.first.c_str()); htext.push_back(h.second.c_str()); }
    for (const auto &o : q.opts) opts.push_back(o.c_str());
    {   // which file the compiler is: the path dlopen resolved (a wheel's bundled libhiprtc answers to the same soname)
        Dl_info info;
        if (rtc.CompileProgram && dladdr((const void *)rtc.CompileProgram, &info) && info.dli_fname) r.rtc_path = info.dli_fname;
    }
    void *prog = nullptr;
    if (rtc.CreateProgram(&prog, q.top.c_str(), q.top_name.c_str(), (int)htext.size(), htext.data(), hname.data()) != 0) { r.log = "hiprtcCreateProgram failed"; return; }
    bool ok = true;
    for (const auto &e : q.exprs) ok = ok && rtc.AddNameExpression(prog, e.c_str()) == 0;
    const int rc = ok ? rtc.CompileProgram(prog, (int)opts.size(), opts.data()) : -1;
    if (rc != 0) {
        size_t n = 0;
        if (rtc.GetProgramLogSize && rtc.GetProgramLogSize(prog, &n) == 0 && n > 1) { r.log.resize(n); rtc.GetProgramLog(prog, &r.log[0]); }
        else r.log = "hiprtcCompileProgram failed";
        rtc.DestroyProgram(&prog);
        return;
    }
    for (const auto &e : q.exprs) {
        const char *lowered = nullptr;
        ok = ok && rtc.GetLoweredName(prog, e.c_str(), &lowered) == 0 && lowered;
        if (ok) r.lowered.push_back(lowered);
    }
    size_t sz = 0;
    ok = ok && rtc.GetCodeSize(prog, &sz) == 0 && sz > 0;
    if (ok) { r.code.resize(sz); ok = rtc.GetCode(prog, r.code.data()) == 0; }
    rtc.DestroyProgram(&prog);
    if (!ok) { r.log = "hiprtc: no code object / lowered names"; r.lowered.clear(); r.code.clear(); return; }
    r.status = 0;
}

// ---- files between the library and its helper: length-prefixed fields, native byte order (same machine) ----
inline void put(FILE *f, const void *p, size_t n) { uint64_t len = n; fwrite(&len, 8, 1, f); if (n) fwrite(p, 1, n, f); }
inline void put(FILE *f, const std::string &s) { put(f, s.data(), s.size()); }
inline bool get(FILE *f, std::string &s)
{
    uint64_t len = 0;
    if (fread(&len, 8, 1, f) != 1 || len > (1ull << 30)) return false;
    s.resize((size_t)len);
    return len == 0 || fread(&s[0], 1, (size_t)len, f) == len;
}
inline bool write_request(const char *path, const Request &q)
{
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    put(f, std::string("PTRTCQ1")); put(f, q.top); put(f, q.top_name);
    put(f, std::to_string(q.headers.size()));
    for (const auto &h : q.headers) { put(f, h.first); put(f, h.second); }
    put(f, std::to_string(q.exprs.size()));
    for (const auto &e : q.exprs) put(f, e);
    put(f, std::to_string(q.opts.size()));
    for (const auto &o : q.opts) put(f, o);
    return fclose(f) == 0;
}
inline bool read_request(const char *path, Request &q)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    std::string magic, n;
    bool ok = get(f, magic) && magic == "PTRTCQ1" && get(f, q.top) && get(f, q.top_name) && get(f, n);
    for (int i = 0, c = ok ? atoi(n.c_str()) : 0; ok && i < c; i++) { std::pair<std::string, std::string> h; ok = get(f, h.first) && get(f, h.second); q.headers.push_back(h); }
    ok = ok && get(f, n);
    for (int i = 0, c = ok ? atoi(n.c_str()) : 0; ok && i < c; i++) { std::string e; ok = get(f, e); q.exprs.push_back(e); }
    ok = ok && get(f, n);
    for (int i = 0, c = ok ? atoi(n.c_str()) : 0; ok && i < c; i++) { std::string o; ok = get(f, o); q.opts.push_back(o); }
    fclose(f);
    return ok;
}
inline bool write_result(const char *path, const Result &r)
{
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    put(f, std::string("PTRTCR2")); put(f, std::to_string(r.status)); put(f, r.log); put(f, r.rtc_path);
    put(f, std::to_string(r.lowered.size()));
    for (const auto &l : r.lowered) put(f, l);
    put(f, r.code.data(), r.code.size());
    return fclose(f) == 0;
}
inline bool read_result(const char *path, Result &r)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    std::string magic, n, code;
    bool ok = get(f, magic) && magic == "PTRTCR2" && get(f, n);
    if (ok) r.status = atoi(n.c_str());
    ok = ok && get(f, r.log) && get(f, r.rtc_path) && get(f, n);
    for (int i = 0, c = ok ? atoi(n.c_str()) : 0; ok && i < c; i++) { std::string l; ok = get(f, l); r.lowered.push_back(l); }
    ok = ok && get(f, code);
    if (ok) r.code.assign(code.begin(), code.end());
    fclose(f);
    return ok;
}

}  // namespace ptrtc
