"""Build libpathtrace_hip.so (HIP kernels for gfx950 + C-ABI + C++ host front end) in-tree.

    python -m pathtrace_amd.build [--force]

hipcc cross-compiles gfx950 without a GPU.  Flags that matter for parity: -ffp-contract=off (no FMA
contraction on host or device) and the default correctly-rounded f32 divide/sqrt (no -ffast-math).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libpathtrace_hip.so")
SPEC_CC = os.path.join(LIB_DIR, "pt_spec_cc")   # the per-scene build's compiler process (csrc/device/pt_spec_cc.cpp), beside the library
SOURCES = [
    os.path.join(CSRC, "device", "pt_kernels.hip"),
    os.path.join(CSRC, "device", "pt_context.cpp"),
    os.path.join(CSRC, "device", "pt_spec.cpp"),
    os.path.join(CSRC, "device", "pt_multi.cpp"),
    os.path.join(CSRC, "host", "pt_host.cpp"),
]
HEADERS = [
    os.path.join(CSRC, "device", "pt_device.h"),
    os.path.join(CSRC, "device", "pt_fdiv.h"),
    os.path.join(CSRC, "device", "pt_spec.h"),
    os.path.join(CSRC, "device", "pt_rtc_core.h"),
    os.path.join(CSRC, "device", "pt_spec_cc.cpp"),
    os.path.join(CSRC, "host", "json_min.h"),
    os.path.join(HERE, "..", "include", "pathtrace_hip.h"),
]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc otherwise packs adjacent scalar f32 mul/add into v_pk_*_f32, whose SGPR operands must be
# aligned pairs -- every packed op then costs two s_mov on the (single per CU) scalar unit and extra VGPRs.
# PT_ROCM_LIB_DIR: where this toolchain's libhiprtc / libamd_comgr live -- the per-scene build (pt_spec.cpp) loads THEM, not
# whatever ROCm the host process may carry (a PyTorch wheel bundles its own, older, compiler)
# As hipcc was NAMED (normally /opt/rocm/lib, the symlink): the versioned directory behind it need not exist on the machine
# that runs the library; pt_spec.cpp checks the directory at run time and falls back to /opt/rocm/lib.
ROCM_LIB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(HIPCC))), "lib")
if not os.path.exists(os.path.join(ROCM_LIB_DIR, "libhiprtc.so")):
    ROCM_LIB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))), "lib")


def hipcc_producer() -> str:
    """'clang version X.Y.Z (repository build-id)' of the hipcc that builds the library: the text a code object's producer string must contain for
    the per-scene module to count as built by this toolchain (pt_spec_info's own_compiler)."""
    import re
    try:
        out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    except OSError:
        return ""
    m = re.search(r"clang version [^\n\"\\]+", out)   # version and the build id in parentheses: "clang version 22.0.0git (... roc-7.2.0 ...)"
    return m.group(0).strip() if m else ""


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-ldl", f'-DPT_ROCM_LIB_DIR="{ROCM_LIB_DIR}"']


def embed_sources(defs=()) -> None:
    """The device sources as string literals for pt_spec.cpp (the per-scene hiprtc build compiles them again at pt_create):
    generated files, never edited and not tracked."""
    for name, src in (("kernels", "pt_kernels.hip"), ("device", "pt_device.h"), ("fdiv", "pt_fdiv.h")):
        text = open(os.path.join(CSRC, "device", src)).read()
        assert ")PTSRC\"" not in text
        # a string literal is limited to 64 KiB by some compilers: emit adjacent raw literals of at most 16 KiB
        parts, step = [], 16000
        for i in range(0, len(text), step):
            parts.append('R"PTSRC(' + text[i:i + step] + ')PTSRC"')
        out = os.path.join(CSRC, "device", f"pt_kernel_src_{name}.inc")
        new = "\n".join(parts) + "\n"
        if not os.path.exists(out) or open(out).read() != new:
            with open(out, "w") as f:
                f.write(new)
    # the -D flags of an A/B variant reach the per-scene hiprtc build too: the module and the library must be the same kernels
    # (a module built with the default flags beside a library built without the live-count bound faulted: one side never
    # wrote the words the other read).  The specialisation's own knobs are the module's business.
    keep = [d for d in defs if d.startswith("-DPT_") and not d.startswith(("-DPT_SPEC_HEADER", "-DPT_CONNECT_WAVES", "-DPT_CONNECT_PREFETCH", "-DPT_CONNECT_NOHOIST"))]
    out = os.path.join(CSRC, "device", "pt_kernel_src_flags.inc")
    new = "".join('"%s",\n' % d.replace("\\", "\\\\").replace('"', '\\"') for d in keep) + "nullptr\n"
    if not os.path.exists(out) or open(out).read() != new:
        with open(out, "w") as f:
            f.write(new)


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(SPEC_CC):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = False, defs=(), out: str = None) -> str:
    """`defs` / `out`: an A/B variant with extra -D flags under another file name (loaded through PATHTRACE_HIP_LIB)."""
    if out is None and not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    embed_sources(defs)
    target = os.path.join(LIB_DIR, out) if out else LIB
    cmd = [HIPCC] + FLAGS + [f'-DPT_HIPCC_PRODUCER="{hipcc_producer()}"'] + list(defs) + SOURCES + ["-o", target]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    # the helper is plain host C++ (it loads hiprtc itself): one file, no HIP
    cc = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", f'-DPT_ROCM_LIB_DIR="{ROCM_LIB_DIR}"',
          os.path.join(CSRC, "device", "pt_spec_cc.cpp"), "-o", SPEC_CC, "-ldl"]
    if verbose:
        print(" ".join(cc), flush=True)
    subprocess.run(cc, check=True)
    return target


if __name__ == "__main__":
    # python -m pathtrace_amd.build [--force] [--variant NAME -DFOO ...]  ->  lib/libpathtrace_hip_NAME.so
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build(force=True, verbose=True, defs=[a for a in sys.argv[i + 2:]], out=f"libpathtrace_hip_{sys.argv[i + 1]}.so"))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
