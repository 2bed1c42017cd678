"""Condense the rocprofv3 output of tools/profile_gpu.sh into the small files kept under profiles/.

    python tools/summarize_profiles.py <tag> <dir with <tag>_prof/, <tag>_pmc_fetch/, <tag>_pmc_write/>

Writes (into <dir>/summary_<tag>/, to be copied to profiles/):
  <tag>_kernel_stats.csv   rocprofv3's own kernel_stats table (name, calls, total/avg/min/max ns, %)
  hbm_traffic.json         per kernel: FETCH_SIZE / WRITE_SIZE (KB, summed over XCDs) per launch and
                           hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: the gfx950 correction of
                           /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3) -- wide coalesced reads are
                           under-reported by 2x, writes are not.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

KERNELS = ("generate", "extend", "shade", "connect", "accumulate")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha16():
    """The hash bench.py compares before it quotes a PMC summary: these counters describe THIS build of the kernels."""
    import hashlib
    h = hashlib.sha256()
    for f in ("pt_kernels.hip", "pt_device.h", "pt_fdiv.h"):
        with open(os.path.join(ROOT, "pathtrace_amd", "csrc", "device", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def kname(full):
    for k in KERNELS:
        if "k_" + k in full:
            return k
    return None


def pmc_per_launch(d, counter):
    f = find(d, "counter_collection.csv")
    if not f:
        return {}
    per = defaultdict(lambda: defaultdict(float))   # kernel -> dispatch id -> value
    dur = defaultdict(dict)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            k = kname(row.get("Kernel_Name", ""))
            if k:
                per[k][row.get("Dispatch_Id")] += float(row.get("Counter_Value", 0))
                dur[k][row.get("Dispatch_Id")] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
    return {k: (sum(v.values()) / len(v), len(v), sum(dur[k].values()) / len(v)) for k, v in per.items() if v}


def insts_per_kernel(d):
    """Per kernel, over all its dispatches of the counter pass (dispatches are serialised there): instruction counts per
    wave and the share of the vector issue roof the kernel sustained.  A wave64 FP32 mul/add/fma occupies its SIMD for 2
    cycles (MI355X_MICROARCH.md; every other class 4 or more), so a chip of 256 CUs x 4 SIMDs issues at most
    1024 * f / 2 vector instructions per second; f is taken from GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch
    time, i.e. the clock the dispatch actually ran at."""
    f = find(d, "counter_collection.csv")
    if not f:
        return {}
    acc = defaultdict(lambda: defaultdict(float))
    dur = defaultdict(dict)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = kname(row.get("Kernel_Name", ""))
            if not k:
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            dur[k][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
    out = {}
    for k, c in acc.items():
        t = sum(dur[k].values())
        waves = max(c.get("SQ_WAVES", 0.0), 1.0)
        clock = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / t if t > 0 else 0.0
        peak = 1024.0 * clock / 2.0
        out[k] = {"dispatches": len(dur[k]), "seconds": round(t, 6), "waves": int(waves),
                  "valu_per_wave": round(c.get("SQ_INSTS_VALU", 0) / waves, 1),
                  "salu_per_wave": round(c.get("SQ_INSTS_SALU", 0) / waves, 1),
                  "smem_per_wave": round(c.get("SQ_INSTS_SMEM", 0) / waves, 1),
                  "lds_per_wave": round(c.get("SQ_INSTS_LDS", 0) / waves, 1),
                  "effective_clock_GHz": round(clock / 1e9, 3),
                  "valu_insts_per_s": round(c.get("SQ_INSTS_VALU", 0) / t, 1) if t > 0 else 0.0,
                  "valu_issue_fraction": round(c.get("SQ_INSTS_VALU", 0) / t / peak, 4) if t > 0 and peak > 0 else None,
                  "salu_over_valu": round(c.get("SQ_INSTS_SALU", 0) / max(c.get("SQ_INSTS_VALU", 0), 1.0), 3)}
    return out


def wait_states(d):
    """Pass 5 (PATHTRACE_HIP_LANES=1: every kernel alone on the chip): where a wave's life goes.  SQ_WAVE_CYCLES, SQ_WAIT_*
    and SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); WAIT_ANY (parked on s_waitcnt / barrier)
    + WAIT_INST_ANY (stalled at issue) + ACTIVE_INST_ANY ~ WAVE_CYCLES.  simd_cycles_per_valu_inst = dispatch time x 1024
    SIMDs x clock / vector instructions: the class-weighted cost of the instruction mix is ~3.5 cycles."""
    f = find(d, "counter_collection.csv")
    if not f:
        return {}
    acc = defaultdict(lambda: defaultdict(float))
    dur = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        if not k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    out = {}
    for k, c in acc.items():
        t = sum(dur[k].values())
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        waves = max(c.get("SQ_WAVES", 0.0), 1.0)
        clock = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / t if t > 0 else 0.0
        out[k] = {"dispatches": len(dur[k]), "seconds": round(t, 6),
                  "wave_life_cycles": round(4 * wc / waves),
                  "parked_on_waitcnt_or_barrier": round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
                  "stalled_at_issue": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                  "executing": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                  "executing_valu": round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                  "resident_waves_per_simd": round(4 * wc / (t * clock) / 1024, 2) if clock > 0 else None,
                  "simd_cycles_per_valu_inst": round(t * 1024 * clock / max(c.get("SQ_INSTS_VALU", 0), 1.0), 2)}
    return out


def main():
    tag, d = sys.argv[1], sys.argv[2]
    out = os.path.join(d, "summary_" + tag)
    os.makedirs(out, exist_ok=True)
    ks1 = find(os.path.join(d, tag + "_prof1"), "kernel_stats.csv")
    if ks1:
        shutil.copy(ks1, os.path.join(out, tag + "_kernel_stats_one_lane.csv"))
    ks = find(os.path.join(d, tag + "_prof"), "kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(out, tag + "_kernel_stats.csv"))
    b = os.path.join(d, tag + "_bench_default.json")
    if os.path.exists(b):
        shutil.copy(b, os.path.join(out, tag + "_bench_default.json"))
    fetch = pmc_per_launch(os.path.join(d, tag + "_pmc_fetch"), "FETCH_SIZE")
    write = pmc_per_launch(os.path.join(d, tag + "_pmc_write"), "WRITE_SIZE")
    sha = kernel_source_sha16()
    res = {"kernel_source_sha16": sha,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_gpu.sh " + tag + "), "
                   "one lane (every dispatch alone on the chip); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                   "the gfx950 correction for wide coalesced reads; averages over all launches of all bounces; "
                   "seconds_per_launch / TBps from the FETCH pass's own dispatch timestamps"}
    for k in KERNELS:
        if k in fetch and k in write:
            fk, n, sec = fetch[k]
            wk, _, _ = write[k]
            b = int((2 * fk + wk) * 1024)
            res[k] = {"launches_sampled": n, "FETCH_SIZE_KB_per_launch": round(fk, 1), "WRITE_SIZE_KB_per_launch": round(wk, 1),
                      "hbm_bytes_per_launch": b, "seconds_per_launch": round(sec, 9), "TBps": round(b / sec / 1e12, 3) if sec > 0 else None}
    json.dump(res, open(os.path.join(out, tag + "_hbm_traffic.json"), "w"), indent=1)
    ins = insts_per_kernel(os.path.join(d, tag + "_pmc_insts"))
    if ins:
        # What a vector instruction of each kernel costs to issue: the class-weighted mean over the STATIC mix of the instantiation
        # that ran (tools/isa_stats.py classes and their measured cycles, tools/microbench/valu_rates.hip) -- k_shade from the
        # library, k_extend / k_connect from the scene's per-scene module (tools/spec_isa.py compiles it with hipcc).  bench.py's
        # issue_frac = wave instructions per second x this / (1024 SIMDs x clock).  A static mix weighs cold blocks like hot ones:
        # an estimate, stated as such.
        try:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import isa_stats
            import spec_isa
            scene_file = os.environ.get("PT_PROFILE_SCENE") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "cornell_box.json")
            stat = {}
            for k in isa_stats.analyze(kernel="k_shade<false, 1, true, false>"):
                stat["shade"] = (k["name"], isa_stats.issue_cycles(k["total"]))
            for k in spec_isa.analyze(scene_file, 4):
                if "k_extend" in k["name"] and k["name"].rstrip(">").endswith("false"):
                    stat["extend"] = (k["name"], isa_stats.issue_cycles(k["total"]))
                if "k_connect" in k["name"]:
                    stat["connect"] = (k["name"], isa_stats.issue_cycles(k["total"]))
            for kk, (name, (n, cyc)) in stat.items():
                if kk in ins:
                    ins[kk]["issue_cycles_per_valu"] = round(cyc, 3)
                    ins[kk]["issue_cycles_from"] = f"static mix of {name} ({n} vector instructions)"
                    ins[kk]["issue_frac"] = round(ins[kk]["valu_insts_per_s"] * cyc / (1024 * 2.4e9), 4)
        except Exception as e:   # the counters stand without it
            ins["issue_cycles_error"] = str(e)[:200]
        ins["kernel_source_sha16"] = sha
        ins["note"] = ("rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS "
                       "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (tools/profile_gpu.sh " + tag + "), bench.py --steps 4 "
                       "--warmup 1; valu_issue_fraction = wave64 VALU instructions per second / (1024 SIMDs * clock / 2), the FP32 mul/add/fma rate")
        json.dump(ins, open(os.path.join(out, tag + "_instruction_mix.json"), "w"), indent=1)
        print(json.dumps({k: v for k, v in ins.items() if isinstance(v, dict)}))
    ws = wait_states(os.path.join(d, tag + "_pmc_wait"))
    if ws:
        ws["kernel_source_sha16"] = sha
        ws["note"] = ("rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                      "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE with PATHTRACE_HIP_LANES=1 (tools/profile_gpu.sh " + tag + " pass 5): every kernel "
                      "alone on the chip; fractions are of SQ_WAVE_CYCLES, i.e. of a wave's life")
        json.dump(ws, open(os.path.join(out, tag + "_wait_states.json"), "w"), indent=1)
        print(json.dumps({k: v for k, v in ws.items() if isinstance(v, dict)}))
    if ins:   # instruction totals of the run: the first thing to compare with the previous profile (same workload every time)
        print("total vector instructions (1e9):", {k: round(v["valu_per_wave"] * v["waves"] / 1e9, 3) for k, v in ins.items() if isinstance(v, dict)})
    print(json.dumps({k: v for k, v in res.items() if isinstance(v, dict)}))
    if ks:
        print(open(ks).read()[:1500])


if __name__ == "__main__":
    main()
