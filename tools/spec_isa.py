#!/usr/bin/env python3
"""Static instruction statistics of a scene's PER-SCENE MODULE (the kernels pt_spec.cpp builds with hiprtc at pt_create),
compiled here with hipcc -- the same compiler -- from the same sources and flags (CPU only).

    python tools/spec_isa.py [scene.json] [--light-samples L] [--flags "..."] [--blocks] [--keep out.s]

hiprtc instantiates the module's kernels through name expressions; here a wrapper source instantiates them explicitly.
"""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

ARGS_EXT = ("ptd::DScene, const ptd::DOp *, const ptd::DInst *, const ptd::DPrim *, const ptd::DMat *, const int32_t *, const float4 *, "
            "ptd::DStreams, ptd::DBatch, int, int")
ARGS_CON = ("ptd::DScene, const ptd::DOp *, const ptd::DInst *, const ptd::DPrim *, const ptd::DMat *, const int32_t *, const float4 *, "
            "ptd::DStreams, ptd::DBatch, int")


def analyze(scene_file, light_samples=4, flags="", keep="", waves=5):
    import isa_stats
    import pathtrace_amd as pt
    sc = pt.Scene(scene_file, 1920, 1080)
    d = tempfile.mkdtemp()
    table = os.path.join(d, "pt_spec_table.h")
    open(table, "w").write(pt.spec_header(sc))
    ga = "true" if "#define PT_SPEC_GA 1" in open(table).read() else "false"
    nr = 2 if light_samples % 2 == 0 else 1
    src = os.path.join(ROOT, "pathtrace_amd", "csrc", "device", "pt_kernels.hip")
    top = os.path.join(d, "top.hip")
    with open(top, "w") as f:
        f.write(f'#define PT_SPEC_BUILD 1\n#define PT_SPEC_HEADER "{table}"\n#include "{src}"\n')
        f.write(f"template __global__ void ptd::k_extend<{ga}, false, false>({ARGS_EXT});\n")
        f.write(f"template __global__ void ptd::k_extend<{ga}, false, true>({ARGS_EXT});\n")
        f.write(f"template __global__ void ptd::k_connect<{nr}, false, {ga}, false>({ARGS_CON});\n")
    fl = f"-DPT_CONNECT_WAVES={waves} -DPT_CONNECT_PREFETCH=0 -DPT_CONNECT_NOHOIST=1 -DPT_CONNECT_WAVES_GA=5 -mllvm -pragma-unroll-threshold=4000000 -I{os.path.dirname(src)} " + flags   # pt_spec.cpp's options
    return isa_stats.analyze(flags=fl, src=top, keep=keep)


def main():
    import isa_stats
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", nargs="?", default=os.path.join(ROOT, "scenes", "cornell_box.json"))
    ap.add_argument("--light-samples", type=int, default=4)
    ap.add_argument("--flags", default="")
    ap.add_argument("--keep", default="")
    ap.add_argument("--blocks", action="store_true")
    a = ap.parse_args()
    for k in analyze(a.scene, a.light_samples, a.flags, a.keep):
        valu, cyc = isa_stats.issue_cycles(k["total"])
        print(f"{k['name']}\n   vgpr {k['vgpr']} sgpr {k['sgpr']} scratch {k['scratch']} occupancy {k['occupancy']} "
              f"sgpr_spills {k['sgpr_spills']} vgpr_spills {k['vgpr_spills']} lds {k['lds']}  VALU {valu}, {cyc:.2f} issue cycles each (static mix)")
        print("   static: " + "  ".join(f"{c} {k['total'].get(c, 0)}" for c in isa_stats.KEYS))
        if a.blocks:
            for lab, d in k["blocks"]:
                n = sum(v for c, v in d.items() if c != "div*")
                if n >= 12:
                    print(f"      {lab:14s} {n:5d}: " + "  ".join(f"{c} {d.get(c, 0)}" for c in isa_stats.KEYS if d.get(c, 0)))


if __name__ == "__main__":
    main()
