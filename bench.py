#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the per-pixel NEE path-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torch.distributed environment, this process starts the N ranks itself as child processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...`, one per GPU) BEFORE anything here
touches the GPU, relays rank 0's JSON line and exits with the children's return code; launched under
torch.distributed.run by someone else it runs as one rank (RANK / LOCAL_RANK / WORLD_SIZE from the environment).

Workload (BASELINE.json configs[1]): scenes/cornell_box.json, 1920x1080, max_bounces 10, light_samples 4, russian
roulette on, normal_offset 1e-4.  One *step* = one pass of the hot path over the whole frame at 16 samples per pixel
(33 M camera samples); the default K = 64 steps is exactly the 1024 spp of configs[1].

"ray" = one World::hit query, extension + shadow, the reference's own unit (integrator.h:192,247).  `value` counts the
queries the device PERFORMED (pt_counters::rays_traced).  The reference-equivalent count (`rays`: light_samples shadow
rays per hit, which is what the reference and the CPU baseline perform for the same image) is ~8 % higher on
cornell_box, because hits none of whose light samples can contribute get no shadow rays here; its rate is reported
beside it as `value_reference_equivalent` and is the figure to compare with `cpu_baseline`.

Launch plan: the LIBRARY's (include/pathtrace_hip.h "the launch plan", ABI v6).  Contexts are created with
max_paths_in_flight = 0, `pt_reserve(pixels, spp)` sizes their streams outside the timed region (set-up, like the scene upload)
and every render is ONE `pt_render_tiles_async` call over all its samples, which the library cuts into as few equal batches as
199 M paths each allow, at least two, a multiple of its lanes -- the same calls the reference-side plugin makes.  `plugin_path`
times that plugin surface itself: `pth_main` (main.cpp:108-168 for render_type hip_wavefront, a child process reading a
config.json) and, when the build container left it, `oracle/_ref/plugin_driver render` (the reference's own Renderer protocol
over tools/integration/hip_wavefront.h), the rates they print beside `value`.

N > 1 (strong scaling: the frame and its K*16 spp are fixed, the metric is "1080p@1024spp at 1/2/4/8 GPU"): the image is
partitioned by 128x128 tile in NaiveSpiral order, ownership balanced over per-tile ray counts measured
in one pass (pt_measure_tile_costs; PT_BENCH_ROUND_ROBIN=1 = tile k -> rank k mod N).  Every rank renders ITS tiles
for all K steps with no communication and ONE collective ends the timed region: a gather of every rank's OWN tiles
(1/N of the frame each, pathtrace_amd.distributed.OwnedTileExchange; RCCL over xGMI) into rank 0's framebuffer --
(N-1)/N of one frame over the links instead of the N-1 whole frames a sum-reduce moves (PT_BENCH_EXCHANGE=reduce: that form).
PT_BENCH_SCALING=weak keeps per-GPU work fixed instead (every step renders 16*N spp).

The timed region holds only GPU work on device-resident data (scene tables + streams live in HBM; there are no host
buffers on the path) and runs with the library's per-launch event profiling OFF.  Rank 0 also reports:
  roofline       from a SERIALISED pass of this very run (one lane: every kernel alone on the chip, HIP events around
                 every launch on its stream): per kernel the SURVEY.md 8d model bytes / its time against 8 TB/s; top level
                 = the dominant kernel.  `overlapped` holds the same events with three batches in flight (shared-chip
                 figures); `pmc` the HBM traffic and vector-instruction counts of rocprofv3 passes under profiles/ when
                 they were taken on this build of the kernels (same source hash), else null
  scaling_proxy  N = 1 only: for N' in 2, 4, 8 every rank's tile list is rendered for the same K steps on THIS GPU (the
                 render has no communication, so rank r's time here is rank r's time at N'): max-over-ranks time,
                 predicted efficiency T1 / (N' x max T_r), ray imbalance.  The framebuffer reduce is not in it.
  configs        32-step sub-runs of BASELINE configs 3, 4 and config 5's 4K frame on this GPU
  cpu_baseline   the oracle (CPU restatement, stream mode, all host cores) on a bounded sample of the same workload
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 1920, 1080
if os.environ.get("PT_BENCH_SIZE"):   # e.g. 3840x2160 for BASELINE config 5's frame; the default is the headline 1080p
    WIDTH, HEIGHT = (int(v) for v in os.environ["PT_BENCH_SIZE"].lower().split("x"))
SCENE = os.path.join(ROOT, "scenes", "cornell_box.json")
SPP_PER_STEP = int(os.environ.get("PT_BENCH_SPP_PER_STEP", "16"))
TILE = 128
LIGHT_SAMPLES = 4              # BASELINE configs: light_samples 4 (pt.Renderer's default)
N_LANES = 3                    # stream lanes of a context (pt_context.cpp; PATHTRACE_HIP_LANES)
# Measurement knob only: an EXPLICIT max_paths_in_flight for every context of this run (e.g. 1920*1080*32 = 32 spp per launch for
# the PMC passes of tools/profile_gpu.sh).  Unset = 0 = the library's launch plan sizes the contexts (PT_PLAN_MAX_PATHS = 199 M
# slots x 288 B x 3 lanes = 172 GB of the 288 GB; same box, K = 64, 66 M / 133 M / 199 M slots: 39.33 / 39.69 / 39.99 Grays/s).
EXPLICIT_PATHS = int(os.environ.get("PT_BENCH_MAX_PATHS", "0"))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# vector ALU roof: 256 CUs x 4 SIMDs, one wave64 FP32 mul/add/fma every 2 cycles at 2.4 GHz (MI355X_MICROARCH.md:
# v_fma_f32 2 cycles per wave64 = 157.3 TFLOP/s); other vector instruction classes issue at 4 or more cycles
VALU_PEAK_WAVE_INSTS = 1024 * 2.4e9 / 2
PIPELINE_BYTES_PER_RAY_16x9 = 153.0         # whole pipeline, cornell_box 16:9


# What a vector instruction costs to issue on one SIMD (tools/microbench/valu_rates.hip): the class table of
# tools/isa_stats.py, which is the one the *_instruction_mix.json summaries' issue_cycles_per_valu were weighted with
from tools.isa_stats import ISSUE_CYCLES  # noqa: E402
SIMDS, CLOCK_HZ = 1024, 2.4e9


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def reference_baseline(scene_file):
    """The REAL reference (oracle/_ref/ref_driver: the reference's own headers compiled in place with its own flags, built in
    the build container, travels as a binary) timed on this box on BASELINE config 1 (200x200x16, threads = 1).  The rate is
    the one the reference prints itself (renderer.h:700-706).  One thread by necessity: with threads > 1 its workers share
    one unsynchronised mt19937 (random.h:9-15) and the image differs from run to run."""
    import re
    import tempfile
    from oracle import pt_oracle, scene_params
    if not pt_oracle.ref_available():
        return None
    try:
        params = scene_params.load_scene_params(scene_file)
        cfg = pt_oracle.make_config(200, 200, 16)
        with tempfile.TemporaryDirectory() as d:
            t0 = time.perf_counter()
            txt = pt_oracle.ref_run(params, "render", pt_oracle._cfg_args(cfg) + [os.path.join(d, "fb.f32")], d)
            wall = time.perf_counter() - t0
        m = re.search(r"computed (\d+) rays, at ([0-9.eE+-]+) rays per second", txt)
        t = re.search(r"time taken to compute ([0-9.eE+-]+)", txt)
        rays = int(m.group(1))
        return {"kind": "reference", "value": round(float(m.group(2)) / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                "cpu_model": cpu_model_name(), "nproc": os.cpu_count(),
                "sample": f"scenes/{os.path.basename(scene_file)} 200x200x16 (BASELINE configs[0]), {rays} rays, the reference's own Tiled + NEEIterative "
                          f"(oracle/_ref/ref_driver), render phase {float(t.group(1)) if t else wall:.2f} s, process {wall:.2f} s",
                "rate_is": "the figure the reference prints (renderer.h:700-706: rays / time taken to compute)",
                "single_threaded_because": "threads > 1 share one unsynchronised mt19937 (random.h:9-15): racy, the image differs per run, "
                                           "and it scales 2.4x on 8 threads (SURVEY 8d)"}
    except Exception as e:   # the headline number does not depend on it
        return {"kind": "reference", "error": str(e)[:200]}


def model_bytes(kernel, d):
    """SURVEY.md 8(d) byte model per kernel for the counter deltas `d` of a pass (N camera samples, E extension rays,
    H extension hits, S shadow rays TRACED -- hits without a shadow record move no record bytes): generate writes ray + state 80 per N; extend reads the ray 32
    and writes the hit 16 per E; shade reads hit + ray + state 96 per E, writes state 48 per H and a shadow record 48 per S;
    connect reads the record 48 and updates the radiance 24 per S; accumulate is the framebuffer update, 32 per N."""
    N, E, H, S = d["camera_samples"], d["extension_rays"], d["extension_hits"], d["shadow_rays_traced"]   # records written / read
    return {"generate": 80 * N, "extend": 48 * E, "shade": 96 * E + 48 * H + 48 * S, "connect": 72 * S, "accumulate": 32 * N}[kernel]


def stream_bytes(kernel, d, light_samples, n_launches, pixels, generate_launches):
    """What THIS implementation's records move per kernel, from the same counter deltas.  Camera rays are not
    stored (bounce 0 forms them itself; no k_generate launch): hit 8 B per extension ray; path record 64 B written per
    continuing path and read back by the next bounce (extend reads its 32-byte ray half); shadow record 16 + 24 L B per hit
    that has one; radiance: 16 B initialised per camera sample, 32 B per update.  Smaller than the model where records were
    shrunk or dropped, larger where the model's 48-byte shadow record per RAY stands against 24 B per ray + 16 B per hit here."""
    N, E, H, S = d["camera_samples"], d["extension_rays"], d["extension_hits"], d["shadow_rays_traced"]
    lit = S / max(light_samples, 1)
    ends = d["term_miss"] + d["term_emitter"]
    fused = generate_launches == 0
    return {"generate": 0 if fused else 48 * N,
            "extend": (32 * (E - N) if fused else 32 * E) + 8 * E,
            "shade": 8 * E + (64 * (E - N) if fused else 32 * E + 32 * (E - N)) + (16 * N if fused else 0) + 64 * (E - N)
                     + (16 + 24 * light_samples) * lit + 32 * ends,
            "connect": (16 + 24 * light_samples) * lit + 32 * lit,
            "accumulate": 16 * N + 32 * pixels * n_launches}[kernel]


def self_launch(n, argv):
    """Start the N ranks as fresh child processes (this parent never touches the GPU) and relay their exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def kernel_source_sha16():
    """Hash of the device sources: a PMC summary under profiles/ describes this build iff it carries the same hash."""
    import hashlib
    h = hashlib.sha256()
    for f in ("pt_kernels.hip", "pt_device.h", "pt_fdiv.h"):
        with open(os.path.join(ROOT, "pathtrace_amd", "csrc", "device", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_profile(dom):
    """HBM traffic and vector-instruction counts of the dominant kernel from the newest committed rocprofv3 PMC summaries
    (tools/profile_gpu.sh writes them).  They are not measured by this run: they are reported only when they were taken on
    this build of the kernels (kernel_source_sha16 recorded in the file), and under their own key."""
    import glob
    sha = kernel_source_sha16()
    out = {"kernel_source_sha16": sha, "traffic_bytes_per_launch": None, "traffic_TBps": None, "traffic_source": None, "valu": None, "wait_states": None,
           "note": "rocprofv3 --pmc passes of tools/profile_gpu.sh (serialised dispatches), not this run; null = no summary of this build"}
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))   # rNNx tags sort by round
    for f in reversed(tfiles):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel_source_sha16") == sha and dom in d:
            out["traffic_bytes_per_launch"] = d[dom].get("hbm_bytes_per_launch")
            out["traffic_seconds_per_launch"] = d[dom].get("seconds_per_launch")
            out["traffic_TBps"] = d[dom].get("TBps")   # bytes and seconds of the SAME pass (its launches hold 32 spp: PT_BENCH_MAX_PATHS = 1920*1080*32)
            out["traffic_source"] = "profiles/" + os.path.basename(f)
            break
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_wait_states.json")))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel_source_sha16") == sha and dom in d and "cfg" not in os.path.basename(f) and "textured" not in os.path.basename(f):
            w = d[dom]
            out["wait_states"] = {"parked": w.get("parked_on_waitcnt_or_barrier"), "stalled_at_issue": w.get("stalled_at_issue"), "executing": w.get("executing"),
                                  "resident_waves_per_simd": w.get("resident_waves_per_simd"), "source": "profiles/" + os.path.basename(f)}
            break
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_instruction_mix.json")))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        mk = d.get(dom)
        if d.get("kernel_source_sha16") == sha and mk:
            out["valu"] = {"valu_per_wave": mk["valu_per_wave"], "salu_over_valu": mk["salu_over_valu"], "issue_cycles_per_valu": mk.get("issue_cycles_per_valu"),
                           "wave_insts_per_s": mk["valu_insts_per_s"], "peak_wave_insts_per_s": VALU_PEAK_WAVE_INSTS,
                           "frac": round(mk["valu_insts_per_s"] / VALU_PEAK_WAVE_INSTS, 4),
                           "source": "profiles/" + os.path.basename(f)}
            break
    return out


def sweep_label(spec_state, info):
    """Which kernels traced, and who compiled them (pt_spec_info): a module built by a compiler other than the library's own
    (an in-process fallback onto a PyTorch wheel's bundled hiprtc / comgr) is correct and measurably slower -- the line says so."""
    if spec_state != 1:
        return {"sweep": "generic kernels", "info": info}
    foreign = info.get("own_compiler") is False
    return {"sweep": "per-scene build" + (" (FOREIGN COMPILER)" if foreign else ""), "built_by": info.get("built_by"),
            "rtc_lib": info.get("rtc_lib"), "producer": info.get("producer"), "own_compiler": info.get("own_compiler"),
            "note": info.get("note")}


def plugin_path(scene_file, w, h, spp, fb_timed, value_traced, with_reference_side):
    """The drop-in surface timed as a user of the reference gets it: config.json in, rays per second out.
    (a) `pth_main` (pt_host.cpp: main.cpp:108-168 for render_type "hip_wavefront") in a child process of its own;
    (b) `oracle/_ref/plugin_driver render`: the REFERENCE's Renderer protocol (its constructor, start_render -> sync_progress*
        -> finalize, its film output) over tools/integration/hip_wavefront.h, built in the build container against the reference's
        headers (the binary travels, the sources do not), when present; its framebuffer is held against this run's, bit for bit.
    Both pass max_paths_in_flight = 0 and make ONE pt_render_async call: the launch plan is the library's.  The rates are the ones
    the programs print themselves (renderer.h:700-706's lines), on the library's clock of the render call."""
    import re
    import tempfile
    import numpy as np
    out = {"workload": f"scenes/{os.path.basename(scene_file)} {w}x{h}x{spp}spp, config.json -> render_type hip_wavefront, max_paths_in_flight 0 (the library's launch plan)",
           "value_of_this_run": value_traced}

    def rates(txt):
        d = {}
        m = re.search(r"time taken to compute ([0-9.eE+-]+)", txt)
        d["seconds"] = float(m.group(1)) if m else None
        m = re.search(r"polling included: ([0-9.eE+-]+)", txt)
        d["seconds_process_clock"] = float(m.group(1)) if m else None
        m = re.search(r"computed (\d+) rays, at ([0-9.eE+-]+) rays per second", txt)
        d["rays"], d["Mrays_per_s_reference_count"] = (int(m.group(1)), round(float(m.group(2)) / 1e6, 2)) if m else (None, None)
        m = re.search(r"traced (\d+) rays, at ([0-9.eE+-]+) rays per second", txt)
        d["rays_traced"], d["value"] = (int(m.group(1)), round(float(m.group(2)) / 1e6, 2)) if m else (None, None)
        m = re.search(r"launch plan: (\d+) batches of (\d+) spp", txt)
        d["launch_plan"] = {"batches": int(m.group(1)), "spp_per_batch": int(m.group(2))} if m else None
        d["vs_value"] = round(d["value"] / value_traced, 4) if d["value"] and value_traced else None
        return d

    with tempfile.TemporaryDirectory() as wd:
        os.makedirs(os.path.join(wd, "output"))
        cfg = {"film": {"width": w, "height": h, "gamma": 2.2, "exposure": 0.0}, "ppm_output_path": "output/render.ppm", "png_output_path": "output/render.png",
               "traced_paths_output_path": "output/out.txt", "traced_paths_2d_output_path": "output/out_2d.txt", "scene": os.path.abspath(scene_file),
               "should_trace_paths": False, "block_width": 128, "block_height": 128, "render_type": "hip_wavefront",
               "integrator_type": "iterative nee path tracing", "max_bounces": 10, "samples": spp, "threads": 1, "normal_offset": 0.0001,
               "light_samples": LIGHT_SAMPLES, "russian_roulette": True, "only_direct_illumination": False}
        with open(os.path.join(wd, "config.json"), "w") as f:
            json.dump(cfg, f)
        code = "import sys; sys.path.insert(0, %r); import pathtrace_amd as pt; rc = pt.lib().pth_main(%r.encode()); sys.stdout.flush(); print('ERR ' + pt.last_error()) if rc else None; sys.exit(1 if rc else 0)" % (ROOT, wd)
        try:
            t0 = time.perf_counter()
            cp = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
            out["pth_main"] = dict(rates(cp.stdout), process_s=round(time.perf_counter() - t0, 2), rc=cp.returncode)
            if cp.returncode:
                out["pth_main"]["error"] = (cp.stdout + cp.stderr)[-400:]
        except Exception as e:
            out["pth_main"] = {"error": str(e)[:300]}
        drv = os.path.join(ROOT, "oracle", "_ref", "plugin_driver")
        if with_reference_side and os.path.exists(drv):
            try:
                from oracle import scene_params
                params = os.path.join(wd, "params.txt")
                with open(params, "w") as f:
                    f.write(scene_params.to_text(scene_params.load_scene_params(scene_file)))
                fbf = os.path.join(wd, "fb.f32")
                t0 = time.perf_counter()
                cp = subprocess.run([drv, params, "render", str(w), str(h), str(spp), "10", str(LIGHT_SAMPLES), "1", "0.0001", "0", "128", "128", fbf],
                                    capture_output=True, text=True, timeout=600, cwd=wd)
                d = dict(rates(cp.stdout), process_s=round(time.perf_counter() - t0, 2), rc=cp.returncode)
                if cp.returncode == 0 and fb_timed is not None:
                    pfb = np.fromfile(fbf, np.float32).reshape(h, w, 3)
                    same = (pfb.view(np.uint32) == fb_timed.view(np.uint32)) | (pfb == fb_timed)
                    d["framebuffer_vs_this_run"] = {"pixel_channels": int(same.size), "mismatched": int((~same).sum()), "tolerance_ulp": 0}
                elif cp.returncode:
                    d["error"] = (cp.stdout + cp.stderr)[-400:]
                out["reference_side_plugin"] = d
            except Exception as e:
                out["reference_side_plugin"] = {"error": str(e)[:300]}
        else:
            out["reference_side_plugin"] = None
    best = [v for v in (out.get("pth_main", {}).get("value"), (out.get("reference_side_plugin") or {}).get("value")) if v]
    out["value"] = min(best) if best else None   # the slower of the two surfaces
    out["vs_value"] = round(out["value"] / value_traced, 4) if best and value_traced else None
    out["unit"] = "Mrays/s traced"
    return out


def sub_config(pt, name, scene_file, w, h, spp_step, steps, device, oracle=None):
    """Another BASELINE configuration on this GPU: Mrays/s (unprofiled, `steps` steps), rays per camera sample, and the
    kernel times of a short serialised pass (one lane)."""
    scene = pt.Scene(os.path.join(ROOT, "scenes", scene_file), w, h)
    total = spp_step * steps
    r = pt.Renderer(scene, device=device, seed=0, max_paths_in_flight=EXPLICIT_PATHS)
    r.reserve(w * h, total)   # the library's launch plan sizes the streams (set-up, outside the timed region)
    spec = r.spec_wait()   # the scene's own build of the traversal kernels (hiprtc at pt_create); -1 = generic kernels

    def run(s0, s1):
        r.render_async(s0, s1)   # ONE call: the library cuts it into batches
    run(0, total)
    r.wait()
    plan = r.plan()
    spp_launch = plan["spp_per_batch"]
    r.clear()
    t0 = time.perf_counter()
    run(0, total)
    r.wait()
    dt = time.perf_counter() - t0
    c = r.counters()
    r.set_lanes(1)
    r.set_profiling(True)
    run(total, total + spp_launch)
    r.wait()
    kt = r.kernel_times()
    dom = max(("extend", "shade", "connect"), key=lambda k: kt[k]["ms"])
    r.set_profiling(False)
    # parity with the metric, on the kernels that were just timed (this context's per-scene module): an 8 x 8 window in the
    # middle of the frame at 16 spp against the oracle in stream mode, bit for bit, with its ray count
    parity = None
    if oracle is not None:
        import numpy as np
        wx, wy, pspp = (w // 2 - 4) & ~1, (h // 2 - 4) & ~1, 16
        rect = (wx, wy, wx + 8, wy + 8)
        r.clear()
        r.render_async(0, pspp, rect)
        gfb = r.framebuffer()[wy:wy + 8, wx:wx + 8]
        gc = r.counters()
        osc = oracle.Scene.from_json(os.path.join(ROOT, "scenes", scene_file))
        ofb = np.zeros((h, w, 3), np.float32)
        _, oc = osc.render_stream(oracle.make_config(w, h, pspp), seed=0, rect=rect, threads=2, fb=ofb)
        ofb = ofb[wy:wy + 8, wx:wx + 8]
        same = (gfb.view(np.uint32) == ofb.view(np.uint32)) | (gfb == ofb)
        parity = {"window": list(rect), "spp": pspp, "mismatched": int((~same).sum()), "pixel_channels": int(same.size),
                  "rays_equal": gc["rays"] == oc["rays"], "tolerance_ulp": 0}
    info = r.spec_info()
    r.close()
    scene.close()
    return {"config": name, "parity": parity, "module": sweep_label(spec, info), "workload": f"scenes/{scene_file} {w}x{h}, {steps} steps of {spp_step} spp, {spp_launch} spp per launch",
            "plan": plan,
            "value": round(c["rays_traced"] / dt / 1e6, 2), "value_reference_equivalent": round(c["rays"] / dt / 1e6, 2), "unit": "Mrays/s",
            "rays_per_sample": round(c["rays"] / max(c["camera_samples"], 1), 4),
            "traced_share": round(c["rays_traced"] / max(c["rays"], 1), 4), "ms_per_step": round(dt / steps * 1e3, 4),
            "dominant_kernel": "k_" + dom, "sweep": sweep_label(spec, info)["sweep"],
            "kernel_ms_serialised_pass": dict({k: round(v["ms"], 3) for k, v in kt.items()}, spp=spp_launch, lanes=1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true")
    ap.add_argument("--no-scaling-proxy", action="store_true")
    ap.add_argument("--no-plugin-path", action="store_true")
    ap.add_argument("--scene", default=SCENE)
    args = ap.parse_args()

    n = args.gpus
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        if n > 1:
            sys.exit(self_launch(n, sys.argv[1:]))   # nothing above this line has touched torch or HIP
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        n = world

    import torch  # device memory for the framebuffer + torch.distributed (RCCL) only
    import pathtrace_amd as pt

    if not torch.cuda.is_available() or pt.device_count() < 1:
        sys.exit("bench.py: no MI355X visible -- the hot path has no CPU fallback")
    # PT_BENCH_REHEARSAL=1: every rank on GPU 0, gloo instead of RCCL, the reduce staged through host memory.  Only for
    # exercising the N > 1 code path on a one-GPU box; its numbers mean nothing.
    rehearsal = os.environ.get("PT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if n > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=n)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=n, device_id=torch.device("cuda", local_rank))

    weak = os.environ.get("PT_BENCH_SCALING") == "weak"
    scene = pt.Scene(args.scene, WIDTH, HEIGHT)
    from pathtrace_amd.distributed import OwnedTileExchange, measure_tile_costs, reduce_framebuffer, tiles_for_rank
    spiral = pt.spiral_tiles(WIDTH, HEIGHT, TILE, TILE)
    costs = None

    def tile_costs():
        # scheduling set-up, like the scene upload outside the timed region: one pass of one sample per pixel over the
        # frame counts the rays of every tile (identical integers on every rank -> the same ownership map everywhere)
        planner = pt.Renderer(scene, device=local_rank, seed=0, max_paths_in_flight=EXPLICIT_PATHS)
        c = measure_tile_costs(planner, spiral)
        planner.close()
        return c

    setup_t0 = time.perf_counter()
    if n == 1:
        my_tiles = [(0, 0, WIDTH, HEIGHT)]
    else:
        if os.environ.get("PT_BENCH_ROUND_ROBIN") != "1":
            costs = tile_costs()
        my_tiles = tiles_for_rank(WIDTH, HEIGHT, TILE, TILE, rank, n, costs)
    setup_ms = (time.perf_counter() - setup_t0) * 1e3
    # the one collective of the path: every rank's OWN tiles gathered into rank 0's framebuffer (index tensors and staging
    # buffers built here, outside the timed region); PT_BENCH_EXCHANGE=reduce = a sum-reduce of whole frames instead
    exchange = None
    if n > 1 and os.environ.get("PT_BENCH_EXCHANGE", "gather") != "reduce":
        lists = [tiles_for_rank(WIDTH, HEIGHT, TILE, TILE, q, n, costs) for q in range(n)]
        exchange = OwnedTileExchange(lists, WIDTH, HEIGHT, rank, n, torch.device("cpu") if rehearsal else torch.device("cuda", local_rank))
    #   strong (default): total work fixed -- K steps of 16 spp over the frame; each rank renders its 1/N of the pixels
    #   weak:             per-GPU work fixed -- every step renders 16*N spp over the frame
    total_spp = SPP_PER_STEP * args.steps * (n if weak else 1)
    my_pixels = sum((x1 - x0) * (y1 - y0) for (x0, y0, x1, y1) in my_tiles)
    # max_paths_in_flight = 0: the library's launch plan sizes the context (what the plugin surface does); pt_reserve here is
    # set-up like the scene upload -- the first render call would do it otherwise, inside the timed region
    r = pt.Renderer(scene, device=local_rank, seed=0, max_paths_in_flight=EXPLICIT_PATHS)
    r.reserve(my_pixels, total_spp)
    plan0 = r.plan()
    spp_launch = plan0["spp_per_batch"]
    # pt_create started the per-scene build of the traversal kernels (hiprtc, ~2 s, like the scene upload outside the timed
    # region); wait for it so that every timed launch runs the same kernels.  -1: not available, the generic kernels run.
    spec_state = r.spec_wait()
    spec_info = r.spec_info()
    # render straight into a torch tensor so that the final reduce needs no copy
    fb = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    r.set_device_framebuffer(fb.data_ptr(), fb.numel() * 4)

    def render_range(rr, tiles, s0, s1):
        rr.render_tiles_async(tiles, s0, s1)   # ONE call over all the samples: the library's plan cuts it into wavefront batches

    def sync():
        r.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def exchange_fb():
        if dist is None:
            return
        if rehearsal:
            fb_host = fb.cpu()
            exchange.run(fb_host) if exchange else reduce_framebuffer(fb_host, dst=0)
            fb.copy_(fb_host)
        elif exchange:
            exchange.run(fb)
        else:
            reduce_framebuffer(fb, dst=0)

    def timed_pass(spp):
        r.clear()
        sync()
        t0 = time.perf_counter()
        render_range(r, my_tiles, 0, spp)
        r.wait()
        exchange_fb()   # the one exchange of the path; no-op at N = 1
        sync()
        return time.perf_counter() - t0

    warm_spp = SPP_PER_STEP * args.warmup * (n if weak else 1)
    if warm_spp:
        render_range(r, my_tiles, 0, warm_spp)
    sync()
    if dist is not None and not rehearsal and args.warmup > 0:
        exchange_fb()   # untimed: RCCL sets its rings and kernels up on the first collective of this shape
        sync()

    # ---- the timed region: exactly K steps, per-launch profiling off ----
    r.set_profiling(False)
    dt = timed_pass(total_spp)
    ctr = r.counters()
    plan = r.plan()   # how the library cut the timed call
    lib_seconds = r.render_seconds()   # the library's own clock of that call: entry -> the device finishing its last batch
    fb_sum = float(fb[..., :3].double().sum().item()) if rank == 0 else 0.0
    fb_timed = fb[..., :3].cpu().numpy().copy() if (rank == 0 and n == 1) else None   # held against the plugin surface's framebuffer below
    dt_again = timed_pass(total_spp)   # a second pass of the same K steps: the scaling proxy's T1 uses the better of the two, like its ranks
    # ---- the same K steps again with HIP events around every launch, three batches in flight (shared-chip figures) ----
    r.set_profiling(True)
    dt_prof = timed_pass(total_spp)
    kt = r.kernel_times()
    # ---- a SERIALISED pass for the roofline block: one lane, so every kernel runs alone on the chip and its HIP-event time is
    # its own; 4 steps (or K if smaller), the same launches as above ----
    ser_spp = SPP_PER_STEP * min(4, args.steps) * (n if weak else 1)
    r.set_lanes(1)
    r.set_profiling(True)
    dt_ser = timed_pass(ser_spp)
    kt_ser = r.kernel_times()
    ctr_ser = r.counters()
    r.set_profiling(False)
    r.set_lanes(N_LANES if not os.environ.get("PATHTRACE_HIP_LANES") else int(os.environ["PATHTRACE_HIP_LANES"]))

    if n == 1:
        # the side blocks below (plugin surface, scaling proxy, other configs, parity) bring contexts of their own, each sized by the
        # library for its job (up to 199 M slots x 288 B x 3 lanes): this one (172 GB at the default) goes first
        r.close()
        r = None
    red_dev = "cpu" if rehearsal else f"cuda:{local_rank}"
    rays = torch.tensor([ctr["rays"], ctr["camera_samples"], ctr["extension_rays"], ctr["extension_hits"],
                         ctr["shadow_rays"], ctr["rays_traced"], ctr["shadow_rays_traced"]], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt, dt_prof], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_rays, total_samples, E, H, S, traced_rays, traced_shadow = [float(x) for x in rays.tolist()]
    dt, dt_prof = [float(x) for x in tmax.tolist()]

    if rank == 0:
        # ---- roofline: rank 0's own launches of the serialised pass ----
        ser = {}
        for k in ("generate", "extend", "shade", "connect", "accumulate"):
            ms, launches = kt_ser[k]["ms"], kt_ser[k]["launches"]
            sb = stream_bytes(k, ctr_ser, LIGHT_SAMPLES, max(kt_ser["accumulate"]["launches"], 1), my_pixels, kt_ser["generate"]["launches"])
            mb = model_bytes(k, ctr_ser) if launches else 0   # no launch (k_generate: bounce 0 forms the camera rays), no bytes
            if k == "accumulate":
                mb = sb   # the 8(d) model's 32 B framebuffer update per camera sample is per PASS here: one per pixel and launch, 16 B read per sample
            gbs = mb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            ser[k] = {"ms": round(ms, 4), "launches": launches, "model_bytes": int(mb), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                      "stream_bytes": int(sb), "stream_GBps": round(sb / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0}
        dom = max(("extend", "shade", "connect"), key=lambda k: kt_ser[k]["ms"])
        kernel_ms_sum = sum(v["ms"] for v in kt_ser.values())
        pmc = pmc_profile(dom)
        ser_rays = ctr_ser["rays_traced"]
        avg_ms = kt_ser[dom]["ms"] / max(ser[dom]["launches"], 1)
        roofline = {"bound": "hbm", "kernel": "k_" + dom,
                    "achieved": ser[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ser[dom]["frac"],
                    "traffic": pmc["traffic_bytes_per_launch"],
                    "avg_launch_ms": round(avg_ms, 5), "launches": ser[dom]["launches"],
                    "bytes_per_launch_model": round(ser[dom]["model_bytes"] / max(ser[dom]["launches"], 1), 1),
                    "how": "achieved / frac = SURVEY 8d ALGORITHMIC bytes of the dominant kernel over its launches / their summed HIP-event time "
                           "(one-lane pass below: the kernel has the chip to itself).  The records this implementation moves are smaller than the "
                           "model's (hbm_frac_stream), the counters agree with them (hbm_frac_counters), and the kernel is bound by vector-"
                           "instruction issue, not by HBM: `bound` names the largest of issue_frac / hbm_frac_counters (or _stream)",
                    "hbm_frac_model": ser[dom]["frac"],
                    "hbm_frac_stream": round(ser[dom]["stream_GBps"] / HBM_PEAK_GBS, 4),
                    "hbm_frac_counters": None, "issue_frac": None,
                    "serialised": {"steps": ser_spp // SPP_PER_STEP, "spp": ser_spp, "lanes": 1, "wall_ms": round(dt_ser * 1e3, 3),
                                   "kernel_ms_sum": round(kernel_ms_sum, 3), "kernels": ser,
                                   "pipeline_model_GBps": round(sum(v["model_bytes"] for v in ser.values()) / (kernel_ms_sum * 1e-3) / 1e9, 1) if kernel_ms_sum > 0 else None,
                                   "pipeline_stream_GBps": round(sum(v["stream_bytes"] for v in ser.values()) / (kernel_ms_sum * 1e-3) / 1e9, 1) if kernel_ms_sum > 0 else None,
                                   "Mrays_per_s_traced": round(ser_rays / dt_ser / 1e6, 1)},
                    "overlapped": {"note": f"{N_LANES} batches in flight: launches share the chip, their event times overlap and sum to more than the wall time",
                                   "kernel_ms": {k: round(v["ms"], 3) for k, v in kt.items()},
                                   "kernel_launches": {k: v["launches"] for k, v in kt.items()},
                                   "profiled_pass_ms_per_step": round(dt_prof / args.steps * 1e3, 4),
                                   "pipeline_model_GBps": round(total_rays / dt * PIPELINE_BYTES_PER_RAY_16x9 / 1e9, 2),
                                   "pipeline_model_frac": round(total_rays / dt * PIPELINE_BYTES_PER_RAY_16x9 / 1e9 / (HBM_PEAK_GBS * n), 5)},
                    "pmc": pmc}
        if pmc["traffic_TBps"]:
            # the counters' bytes over the dispatch times of the SAME rocprofv3 pass (same build of the kernels, one lane; its launches
            # hold 32 spp where this run's may hold more: a rate, not a per-launch figure of this run)
            g = pmc["traffic_TBps"] * 1e3
            roofline["hbm_by_counters"] = {"GBps": round(g, 1), "frac": round(g / HBM_PEAK_GBS, 4), "bytes_per_launch_of_that_pass": pmc["traffic_bytes_per_launch"],
                                           "source": pmc["traffic_source"]}
            roofline["hbm_frac_counters"] = round(g / HBM_PEAK_GBS, 4)
        if pmc["valu"]:
            # class-weighted issue model: the kernel's vector instructions per second (SQ_INSTS_VALU of the same-build
            # PMC pass over its own time) x what an instruction of its static mix costs to issue / the chip's SIMD cycles per second
            cyc = pmc["valu"].get("issue_cycles_per_valu")
            if cyc:
                roofline["issue_frac"] = round(pmc["valu"]["wave_insts_per_s"] * cyc / (SIMDS * CLOCK_HZ), 4)
                roofline["issue_model"] = {"wave_insts_per_s": pmc["valu"]["wave_insts_per_s"], "issue_cycles_per_valu_inst": cyc,
                                           "class_cycles": ISSUE_CYCLES, "simd_cycles_per_s": SIMDS * CLOCK_HZ, "source": pmc["valu"]["source"]}
        hbm_real = roofline["hbm_frac_counters"] if roofline["hbm_frac_counters"] is not None else roofline["hbm_frac_stream"]
        ws = pmc.get("wait_states")
        if ws:
            roofline["parked"] = ws["parked"]
            roofline["wave_life"] = ws
        if ws and ws["parked"] is not None and ws["parked"] > (ws["stalled_at_issue"] or 0) + (ws["executing"] or 0):
            # a wave of this kernel spends more of its life parked on s_waitcnt (memory / atomic / LDS round trips in front of the
            # next dependent instruction) than stalled at issue and executing together: bound by its latency chains, whatever
            # the issue and byte fractions read (removing a quarter of k_shade's vector instructions did not shorten it)
            roofline["bound"] = "latency"
            roofline["bound_basis"] = "parked on s_waitcnt %.2f of a wave's life against %.2f stalled at issue + %.2f executing (%s); issue_frac %s, %.2f of the HBM roof" % (
                ws["parked"], ws["stalled_at_issue"], ws["executing"], ws["source"], roofline["issue_frac"], hbm_real)
        elif roofline["issue_frac"] is not None:
            roofline["bound"] = "valu-issue" if roofline["issue_frac"] >= hbm_real else "hbm"
            roofline["bound_basis"] = "issue_frac %.2f against %.2f of the HBM roof (%s)" % (
                roofline["issue_frac"], hbm_real, "counters" if roofline["hbm_frac_counters"] is not None else "this implementation's record bytes")
        else:
            # no PMC pass of this build: the kernel moves hbm_real of the HBM roof by its own records; below 0.6 of it nothing is
            # bandwidth-bound on this chip (a copy sustains 0.79 of spec) -- the remaining candidate is instruction issue
            roofline["bound"] = "hbm" if hbm_real >= 0.6 else "valu-issue"
            roofline["bound_basis"] = "inferred (no PMC pass of this build of the kernels): the kernel's own records move %.2f of the HBM roof" % hbm_real

        # ---- the plugin surface itself, same frame and samples as the timed region ----
        plug = None
        if n == 1 and not weak and not args.no_plugin_path:
            try:
                plug = plugin_path(args.scene, WIDTH, HEIGHT, total_spp, fb_timed, round(traced_rays / dt / 1e6, 2), not args.no_cpu_baseline)
            except Exception as e:   # the headline number does not depend on it
                plug = {"error": str(e)[:300]}
        fb_timed = None

        # ---- strong-scaling proxy on this one GPU (the render has no communication: rank r's time here is rank r's time at N) ----
        proxy = None
        if n == 1 and not args.no_scaling_proxy and not weak:
            proxy = {"note": "every rank's tile list of an N'-rank run rendered on this GPU for the same K steps with that run's launch plan; "
                             "T1 and a rank's time = the better of two passes each (same estimator); the exchange of the owned tiles and host-side launch contention "
                             "between processes are not in it", "T1_ms": round(min(dt, dt_again) * 1e3, 3), "T1_passes_ms": [round(dt * 1e3, 3), round(dt_again * 1e3, 3)], "by_n": []}
            p0 = time.perf_counter()
            costs = costs or tile_costs()
            proxy["planner_ms"] = round((time.perf_counter() - p0) * 1e3, 1)
            if os.environ.get("PT_BENCH_PROXY_DETAIL"):
                proxy["tile_costs"] = costs
                proxy["tiles"] = spiral
            for np_ in (2, 4, 8):
                strat = os.environ.get("PT_BENCH_PARTITION", "lpt")
                lists = [tiles_for_rank(WIDTH, HEIGHT, TILE, TILE, q, np_, None if strat == "rr" else costs, strat) for q in range(np_)]
                pix = [sum((x1 - x0) * (y1 - y0) for (x0, y0, x1, y1) in tl) for tl in lists]
                rp = pt.Renderer(scene, device=local_rank, seed=0, max_paths_in_flight=EXPLICIT_PATHS)
                rp.reserve(max(pix), total_spp)   # every "rank" gets the library's plan for its own pixels; the streams fit the largest
                rp.spec_wait()
                plans = []
                # warm-up: rank 0's whole pass once, so that EVERY lane's streams have been touched before anything is timed (one
                # launch warmed lane 0 only and rank 0, timed first, paid the first use of the other lanes' fresh allocations:
                # +0.6 ms on its 17 ms at K = 20, the slowest "rank" of every round-3 proxy at that K)
                render_range(rp, lists[0], 0, total_spp)
                rp.wait()
                times, rr_, cs_, tr_, full_ = [], [], [], [], []
                for q in range(np_):
                    best = None
                    for _rep in range(2):   # the better of two: one stray 3 ms on one "rank" of eight reads as 0.79 instead of 0.94
                        rp.clear()
                        t0 = time.perf_counter()
                        render_range(rp, lists[q], 0, total_spp)
                        rp.wait()
                        t1 = time.perf_counter() - t0
                        best = t1 if best is None else min(best, t1)
                    times.append(best)
                    plans.append(rp.plan()["spp_per_batch"])
                    rr_.append(rp.counters()["rays"])
                    cs_.append(rp.counters()["camera_samples"])
                    tr_.append(rp.counters()["rays_traced"])
                    if os.environ.get("PT_BENCH_PROXY_DETAIL"):
                        full_.append(dict(rp.counters(), tiles=len(lists[q]), pixels=pix[q]))
                rp.close()
                proxy["by_n"].append({"n": np_, "max_rank_ms": round(max(times) * 1e3, 3), "min_rank_ms": round(min(times) * 1e3, 3),
                                      "predicted_efficiency": round(min(dt, dt_again) / (np_ * max(times)), 4),
                                      "rays_traced_max_over_mean": round(max(tr_) / (sum(tr_) / np_), 4),
                                      "spp_per_launch": plans, "batches_per_rank": [-(-total_spp // sl) for sl in plans],
                                      "rank_ms": [round(t * 1e3, 3) for t in times], "rank_rays": rr_, "rank_camera_samples": cs_, "rank_rays_traced": tr_, "rank_detail": full_ or None,
                                      "rays_all_ranks": int(sum(rr_))})

        configs = None
        if n == 1 and not args.no_configs and not os.environ.get("PT_BENCH_SIZE"):
            configs = []
            cfg_oracle = None
            if not args.no_cpu_baseline:
                from oracle import pt_oracle as cfg_oracle   # the checker of the sub-runs' parity windows
            for name, sf, w, h, spp, st in (("configs[2]", "cornell_box_small_lights.json", 1920, 1080, 16, 32),
                                            ("configs[3]", "cornell_box_with_volume.json", 1920, 1080, 16, 32),
                                            ("configs[4] frame, 1 GPU", "cornell_box.json", 3840, 2160, 4, 32)):
                try:
                    configs.append(sub_config(pt, name, sf, w, h, spp, st, local_rank, cfg_oracle))
                except Exception as e:   # the headline number does not depend on these
                    configs.append({"config": name, "error": str(e)})

        cpu = None
        parity = None
        if not args.no_cpu_baseline:
            from oracle import pt_oracle
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = min(avail, 16)   # one GPU's share of the host (the GPU box guideline), all of them used
            osc = pt_oracle.Scene.from_json(args.scene)
            spp_cpu = 64   # ~1.1e9 rays: 10-15 s on 16 host threads
            cfg = pt_oracle.make_config(WIDTH, HEIGHT, spp_cpu)
            c0 = time.perf_counter()
            ofb, octr = osc.render_stream(cfg, seed=0, threads=cores)
            cdt = time.perf_counter() - c0
            # the reference's own order on one thread (global mt19937, config 1 = 200x200x16): ties the port back to the
            # reference's single-thread rate; the literal reference binary cannot travel to this box
            mcfg = pt_oracle.make_config(200, 200, 16)
            osc_mt = pt_oracle.Scene.from_json(args.scene)
            m0 = time.perf_counter()
            _, mctr = osc_mt.render_mt(mcfg)
            mdt = time.perf_counter() - m0
            cpu = {"value": round(octr["rays"] / cdt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_name(), "nproc": os.cpu_count(),
                   "reference": reference_baseline(args.scene),
                   "sample": f"cornell_box {WIDTH}x{HEIGHT} x {spp_cpu} spp ({octr['rays']} rays, all of them traced: compare with "
                             f"value_reference_equivalent), oracle stream mode, {cores} threads, {cdt:.2f} s wall",
                   "reference_order_1thread": {"value": round(mctr["rays"] / mdt / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                                               "sample": f"cornell_box 200x200x16 ({mctr['rays']} rays), oracle mt19937 mode, "
                                                         f"{mdt:.2f} s wall"}}
            if n == 1:
                # parity reported with the metric, at the bench's full frame size: the same 64 spp on the
                # GPU must be the oracle's framebuffer bit for bit, with equal path counters
                import numpy as np
                rpar = pt.Renderer(scene, device=local_rank, seed=0, max_paths_in_flight=EXPLICIT_PATHS)
                par_module = sweep_label(rpar.spec_wait(), rpar.spec_info())["sweep"]   # the kernels the timed region ran (same scene, cached module)
                rpar.render_async(0, spp_cpu)
                gfb = rpar.framebuffer()
                gctr = rpar.counters()
                rpar.close()
                same = (gfb.view(np.uint32) == ofb.view(np.uint32)) | (gfb == ofb)
                parity = {"against": "oracle stream mode, same seed", "size": f"{WIDTH}x{HEIGHT}x{spp_cpu}", "kernels": par_module,
                          "pixel_channels": int(same.size), "mismatched": int((~same).sum()), "tolerance_ulp": 0,
                          "counters_equal": all(gctr[g] == octr[o] for g, o in (
                              ("rays", "rays"), ("extension_rays", "ext_rays"), ("extension_hits", "ext_hits"),
                              ("shadow_rays", "shadow_rays"), ("term_miss", "term_miss"), ("term_rr", "term_rr"),
                              ("term_emitter", "term_emitter"), ("term_pdf", "term_pdf"),
                              ("term_bounce_limit", "term_bounce_limit")))}

        knobs = {k: v for k, v in os.environ.items() if k.startswith("PATHTRACE_HIP_") or k.startswith("PT_BENCH_")}
        out = {
            "metric": "Mrays/sec + achieved HBM GB/s %peak, cornell_box 1080p@1024spp, 1/2/4/8 GPU",
            "value": round(traced_rays / dt / 1e6, 2),
            "unit": "Mrays/s",
            "value_is": "World::hit queries performed by the device (pt_counters.rays_traced) / wall time of the K steps",
            "value_reference_equivalent": round(total_rays / dt / 1e6, 2),
            "value_reference_equivalent_is": "the reference's count for the same image (light_samples shadow rays per hit, integrator.h:246-247) / the same time; "
                                             "hits whose light samples cannot contribute get no shadow rays here",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"scenes/{os.path.basename(args.scene)} {WIDTH}x{HEIGHT}, iterative NEE path tracing, max_bounces 10, "
                                   f"light_samples 4, russian roulette; step = {SPP_PER_STEP} spp over the frame"
                                   + (f" x {n} (weak)" if weak else "") + f" (K=64 is 1024 spp), {spp_launch} spp per launch on rank 0 "
                                   f"({plan['batches']} batches, cut by the library)",
                       "launch_plan": dict(plan, decided_by="libpathtrace_hip.so (pt_render_tiles_async; include/pathtrace_hip.h, ABI v6)" if not EXPLICIT_PATHS else "PT_BENCH_MAX_PATHS (explicit max_paths_in_flight; the library's rule cuts within it)"),
                       "library_clock_ms": round(lib_seconds * 1e3, 3) if lib_seconds and lib_seconds > 0 else None,
                       "spp_total": total_spp, "camera_samples": int(total_samples), "rays": int(total_rays), "rays_traced": int(traced_rays),
                       "rays_per_sample": round(total_rays / max(total_samples, 1), 4),
                       "E_H_S_per_sample": [round(E / total_samples, 4), round(H / total_samples, 4), round(S / total_samples, 4)],
                       "shadow_rays_traced_share": round(traced_shadow / max(S, 1), 4),
                       "framebuffer_sum": fb_sum,
                       "partition": "whole frame" if n == 1 else (f"128x128 tiles, spiral order, " + ("tile k -> rank k mod N" if os.environ.get("PT_BENCH_ROUND_ROBIN") == "1" else "cost-balanced ownership (LPT over per-tile ray counts)") + "; 1 collective"),
                       "partition_setup_ms": round(setup_ms, 1),
                       "sweep": sweep_label(spec_state, spec_info)["sweep"] + ((" of k_extend and k_connect (hiprtc at pt_create)" if "extend-only" not in os.environ.get("PATHTRACE_HIP_SPEC", "") else " of k_extend (hiprtc at pt_create), generic k_connect") if spec_state == 1 else ""),
                       "module": sweep_label(spec_state, spec_info),
                       "exchange": None if n == 1 else ("gather of owned tiles: %d bytes to rank 0" % exchange.bytes_moved() if exchange else "sum-reduce of whole frames"),
                       "knobs": knobs},
            "roofline": roofline,
            "plugin_path": plug,
            "scaling_proxy": proxy,
            "configs": configs,
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if r is not None:
        r.close()


if __name__ == "__main__":
    main()
