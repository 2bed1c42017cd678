#!/usr/bin/env python3
"""Static instruction-class statistics of the gfx950 kernels (CPU only: hipcc cross-compiles).

    python tools/isa_stats.py [--kernel SUBSTR] [--blocks] [--flags "..."]

Compiles pathtrace_amd/csrc/device/pt_kernels.hip to assembly with the product's flags and prints, per kernel:
registers, spills, occupancy, and the instruction mix split by the issue classes measured on MI355X:
  fp2   v_mul/add/sub_f32 whose operands are VGPRs / constants only       (2.3 - 2.6 cycles per wave64 instruction)
  fp2s  the same with an SGPR operand                                     (3.9)
  fma   v_fma / v_fmac / v_mac / v_mad_f32 (three register reads)         (3.8 - 4.0; round 4's microbench: it is NOT a 2-cycle class)
  v4    every other 32-bit VALU instruction (compare, select, min/max, integer, convert, div scaffolding)  (4)
  pk    v_pk_*_f32: two results per lane                                  (4.1)
  f64   f64 arithmetic and conversions                                    (4.0 - 4.5)
  trans v_rcp/rsq/sqrt/exp/log/sin/cos_f32                                (7.8), v_rcp_f64 16
(tools/microbench/valu_rates.hip; ISSUE_CYCLES below are the figures bench.py's issue model uses)
--blocks lists the basic blocks of the selected kernels with their mixes (the traversal sweep's per-op bodies).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FP2 = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_legacy_f32"}
FMA = {"v_fma_f32", "v_mac_f32", "v_fmac_f32", "v_mad_f32", "v_fmaak_f32", "v_fmamk_f32", "v_madak_f32", "v_madmk_f32"}
ISSUE_CYCLES = {"fp2": 2.45, "fp2s": 3.9, "fma": 3.9, "v4": 4.0, "pk": 4.1, "f64": 4.3, "trans": 7.8}
VALU_CLASSES = ("fp2", "fp2s", "fma", "v4", "pk", "f64", "trans")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def classify(op, args):
    if op.startswith("v_"):
        base = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)
        if base.startswith(TRANS):
            return "trans"
        if base.endswith("_f64") or "_f64_" in base:
            return "f64"
        if base in FMA:
            return "fma"
        if base in FP2:
            # an SGPR source operand (s12, s[4:5], vcc, ...) puts the instruction in the 4-cycle class
            srcs = args.split(",")[1:]
            if any(re.match(r"\s*-?\|?(s\d+|s\[|vcc|ttmp|m0|exec)", s) for s in srcs):
                return "fp2s"
            return "fp2"
        if base.startswith("v_pk_"):
            return "pk"
        return "v4"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


KEYS = ("fp2", "fp2s", "fma", "v4", "pk", "f64", "trans", "salu", "smem", "vmem", "lds", "div*")


def issue_cycles(total):
    """(vector instructions, class-weighted issue cycles per vector instruction) of a static mix {class: n}."""
    n = sum(total.get(c, 0) for c in VALU_CLASSES)
    return n, (sum(total.get(c, 0) * ISSUE_CYCLES[c] for c in VALU_CLASSES) / n if n else 0.0)


def analyze(flags="", src=None, keep="", kernel=""):
    """Compile `src` to gfx950 assembly with the product's flags (+ `flags`) and return one dict per kernel whose demangled
    name contains `kernel`: name, vgpr, sgpr, scratch, occupancy, sgpr_spills, vgpr_spills, lds, total {class: n},
    blocks [(label, {class: n})]."""
    from pathtrace_amd import build as ptb
    src = src or os.path.join(ROOT, "pathtrace_amd", "csrc", "device", "pt_kernels.hip")
    fl = [f for f in ptb.FLAGS if f not in ("-shared", "-fPIC", "-ldl")] + flags.split()
    out = keep or os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.run([ptb.HIPCC] + fl + ["-S", "--cuda-device-only", src, "-o", out], check=True)
    text = open(out).read()
    res = []
    # kernels: from "<sym>:" (a .globl function) to ".Lfunc_end"
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        sym, body = m.group(1), m.group(2)
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().split("(")[0]
        if kernel not in name:
            continue
        tail = text[m.end(): m.end() + 3000]
        info = {k: (re.search(r"; " + k + r": (\S+)", tail) or [None, "?"])[1] for k in
                ("NumSgprs", "NumVgprs", "ScratchSize", "Occupancy", "LDSByteSize")}
        spills = {k: (re.search(r"\." + k + r":\s+(\d+)", text[text.find(sym, text.find("amdhsa.kernels")):][:3000]) or [None, "?"])[1]
                  for k in ("sgpr_spill_count", "vgpr_spill_count")}
        blocks = []
        cur = ["entry", {}]
        tot = {}
        for line in body.splitlines():
            line = line.split(";")[0].rstrip()
            if not line.strip():
                continue
            lab = re.match(r"^(\.LBB\w+):", line)
            if lab:
                blocks.append(cur)
                cur = [lab.group(1), {}]
                continue
            s = line.strip()
            if s.startswith("."):
                continue
            parts = s.split(None, 1)
            op, a = parts[0], (parts[1] if len(parts) > 1 else "")
            c = classify(op, a)
            cur[1][c] = cur[1].get(c, 0) + 1
            tot[c] = tot.get(c, 0) + 1
            if op.startswith("v_div_") or op.startswith("v_rcp_"):
                cur[1]["div*"] = cur[1].get("div*", 0) + 1
                tot["div*"] = tot.get("div*", 0) + 1
        blocks.append(cur)

        def num(v):
            return int(v) if str(v).isdigit() else v
        res.append({"name": name, "vgpr": num(info["NumVgprs"]), "sgpr": num(info["NumSgprs"]), "scratch": num(info["ScratchSize"]),
                    "occupancy": num(info["Occupancy"]), "sgpr_spills": num(spills["sgpr_spill_count"]),
                    "vgpr_spills": num(spills["vgpr_spill_count"]), "lds": num(info["LDSByteSize"]), "total": tot,
                    "blocks": [(lab, d) for lab, d in blocks]})
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="")
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--flags", default="")
    ap.add_argument("--src", default=None)
    ap.add_argument("--keep", default="")
    args = ap.parse_args()
    for k in analyze(args.flags, args.src, args.keep, args.kernel):
        print(f"{k['name']}\n   vgpr {k['vgpr']} sgpr {k['sgpr']} scratch {k['scratch']} occupancy {k['occupancy']} "
              f"sgpr_spills {k['sgpr_spills']} vgpr_spills {k['vgpr_spills']} lds {k['lds']}")
        print("   static: " + "  ".join(f"{c} {k['total'].get(c, 0)}" for c in KEYS))
        if args.blocks:
            for lab, d in k["blocks"]:
                n = sum(v for c, v in d.items() if c != "div*")
                if n >= 12:
                    print(f"      {lab:14s} {n:5d}: " + "  ".join(f"{c} {d.get(c, 0)}" for c in KEYS if d.get(c, 0)))


if __name__ == "__main__":
    main()
