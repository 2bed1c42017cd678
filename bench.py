#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the per-pixel NEE path-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torch.distributed environment, this process starts the N ranks itself as child processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...`, one per GPU) BEFORE anything here
touches the GPU, relays rank 0's JSON line and exits with the children's return code; launched under
torch.distributed.run by someone else it runs as one rank (RANK / LOCAL_RANK / WORLD_SIZE from the environment).

Workload (BASELINE.json configs[1]): scenes/cornell_box.json, 1920x1080, max_bounces 10, light_samples 4, russian
roulette on, normal_offset 1e-4.  One *step* = one pass of the hot path over the whole frame at 16 samples per pixel
(33 M camera samples); the default K = 64 steps is exactly the 1024 spp of configs[1].  Steps are enqueued two per
wavefront launch (66 M paths per batch).  "ray" = one World::hit query,
extension + shadow, the reference's own unit (integrator.h:192,247).

N > 1 (strong scaling: the frame and its K*16 spp are fixed, the metric is "1080p@1024spp at 1/2/4/8 GPU"): the image is
partitioned by 128x128 tile in NaiveSpiral order, ownership balanced over measured per-tile ray counts
(pathtrace_amd/distributed.py; PT_BENCH_ROUND_ROBIN=1 = tile k -> rank k mod N).  Every rank renders ITS tiles for all
K steps with no communication -- 2 N steps per launch, so that a wavefront batch stays at ~66 M paths -- and ONE RCCL
sum-reduce of the framebuffer to rank 0 ends the timed region.  PT_BENCH_SCALING=weak keeps per-GPU work fixed instead
(every step renders 16*N spp).

The timed region holds only GPU work on device-resident data (scene tables + streams live in HBM; there are no host
buffers on the path) and runs with the library's per-launch event profiling OFF.  The same K steps are then repeated with
HIP events around every launch (on the stream each kernel is launched on) for the roofline block.  Rank 0 also reports:
  roofline      dominant kernel: algorithmic bytes / its HIP-event time against 8 TB/s (SURVEY.md 8d byte model), the
                measured HBM traffic and vector-instruction counts of the same build (rocprofv3 PMC passes, profiles/),
                and which roof binds
  configs       short sub-runs of BASELINE configs 3, 4 and config 5's 4K frame on this GPU
  cpu_baseline  the oracle (CPU restatement, stream mode, all host cores) on a bounded sample of the same workload
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
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# vector ALU roof: 256 CUs x 4 SIMDs, one wave64 FP32 mul/add/fma every 2 cycles at 2.4 GHz (MI355X_MICROARCH.md:
# v_fma_f32 2 cycles per wave64 = 157.3 TFLOP/s); other vector instruction classes issue at 4 or more cycles
VALU_PEAK_WAVE_INSTS = 1024 * 2.4e9 / 2
# SURVEY.md 8(d) byte model, per unit of each kernel
BYTES_PER_EXT_RAY_EXTEND = 32 + 16          # read ray, write hit
BYTES_PER_SHADOW_RAY_CONNECT = 48 + 24      # read shadow record, radiance RMW
PIPELINE_BYTES_PER_RAY_16x9 = 153.0         # whole pipeline, cornell_box 16:9


def shade_bytes(E, H, S):
    # read hit+ray+state 96 per E, write state 48 per H, continuation ray 32 per (E - samples), shadow record 48 per S
    return 96 * E + 48 * H + 48 * S


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


def file_derived_profile(dom):
    """HBM traffic and vector-instruction counts of the dominant kernel from the newest committed rocprofv3 PMC summaries
    (tools/profile_gpu.sh writes them; they are NOT measured by this run and are labelled with their file)."""
    import glob
    out = {"traffic": None, "traffic_source": None, "valu": None}
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))   # rNNx tags sort by round
    if tfiles:
        try:
            d = json.load(open(tfiles[-1]))
            out["traffic"] = d.get(dom, {}).get("hbm_bytes_per_launch")
            out["traffic_source"] = "profiles/" + os.path.basename(tfiles[-1])
        except Exception:
            pass
    mixes = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_instruction_mix.json")))
    if mixes:
        try:
            mk = json.load(open(mixes[-1])).get(dom)
            if mk:
                out["valu"] = {"valu_per_wave": mk["valu_per_wave"], "salu_over_valu": mk["salu_over_valu"],
                               "valu_insts_per_s_profiled": mk["valu_insts_per_s"],
                               "source": "profiles/" + os.path.basename(mixes[-1])}
        except Exception:
            pass
    return out


def sub_config(pt, name, scene_file, w, h, spp_step, steps, device):
    """A short run of another BASELINE configuration on this GPU: Mrays/s (unprofiled), rays per camera sample, dominant kernel."""
    scene = pt.Scene(os.path.join(ROOT, "scenes", scene_file), w, h)
    r = pt.Renderer(scene, device=device, seed=0, max_paths_in_flight=w * h * spp_step)
    r.render_async(0, spp_step)
    r.wait()
    r.clear()
    t0 = time.perf_counter()
    for i in range(steps):
        r.render_async(i * spp_step, (i + 1) * spp_step)
    r.wait()
    dt = time.perf_counter() - t0
    c = r.counters()
    r.set_profiling(True)
    for i in range(steps, steps + 2):
        r.render_async(i * spp_step, (i + 1) * spp_step)
    r.wait()
    kt = r.kernel_times()
    dom = max(("extend", "shade", "connect"), key=lambda k: kt[k]["ms"])
    r.close()
    scene.close()
    return {"config": name, "workload": f"scenes/{scene_file} {w}x{h}, {steps} steps of {spp_step} spp",
            "value": round(c["rays"] / dt / 1e6, 2), "unit": "Mrays/s",
            "rays_per_sample": round(c["rays"] / max(c["camera_samples"], 1), 4), "ms_per_step": round(dt / steps * 1e3, 4),
            "dominant_kernel": "k_" + dom,
            "kernel_ms_profiled_2_steps": {k: round(v["ms"], 3) for k, v in kt.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true")
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
    from pathtrace_amd.distributed import measure_tile_costs, reduce_framebuffer, tiles_for_rank
    setup_t0 = time.perf_counter()
    if n == 1:
        my_tiles = [(0, 0, WIDTH, HEIGHT)]
    else:
        # scheduling set-up, like the scene upload outside the timed region: every rank counts the rays of one sample
        # per pixel per tile (identical integers on every rank) and derives the same cost-balanced ownership map
        costs = None
        if os.environ.get("PT_BENCH_ROUND_ROBIN") != "1":
            planner = pt.Renderer(scene, device=local_rank, seed=0, max_paths_in_flight=TILE * TILE)
            costs = measure_tile_costs(planner, pt.spiral_tiles(WIDTH, HEIGHT, TILE, TILE))
            planner.close()
        my_tiles = tiles_for_rank(WIDTH, HEIGHT, TILE, TILE, rank, n, costs)
    setup_ms = (time.perf_counter() - setup_t0) * 1e3
    # One launch = `group` steps of this rank's pixels: a wavefront batch of ~W*H*32 = 66 M paths at every N (measured at
    # N = 1: 33 M-path batches 28.9, 66 M 29.8, 133 M 30.0 Grays/s -- the thin late bounces of a batch cost less per path).
    #   strong (default): total work fixed -- K steps of 16 spp over the frame; each rank renders its 1/N of the pixels
    #                     for all K steps, 2 N steps per launch
    #   weak:             per-GPU work fixed -- every step renders 16*N spp over the frame
    group = int(os.environ.get("PT_BENCH_GROUP", "2")) * n
    spp_launch = SPP_PER_STEP * group
    total_spp = SPP_PER_STEP * args.steps * (n if weak else 1)
    my_pixels = sum((x1 - x0) * (y1 - y0) for (x0, y0, x1, y1) in my_tiles)
    r = pt.Renderer(scene, device=local_rank, seed=0, max_paths_in_flight=my_pixels * spp_launch)
    # render straight into a torch tensor so that the final reduce needs no copy
    fb = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    r.set_device_framebuffer(fb.data_ptr(), fb.numel() * 4)

    def render_range(s0, s1):
        s = s0
        while s < s1:
            e = min(s + spp_launch, s1)
            r.render_tiles_async(my_tiles, s, e)
            s = e

    def sync():
        r.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_pass():
        r.clear()
        sync()
        t0 = time.perf_counter()
        render_range(0, total_spp)
        r.wait()
        if rehearsal and dist is not None:
            fb_host = fb.cpu()
            reduce_framebuffer(fb_host, dst=0)
            fb.copy_(fb_host)
        else:
            reduce_framebuffer(fb, dst=0)   # the one exchange of the path; no-op at N = 1
        sync()
        return time.perf_counter() - t0

    warm_spp = SPP_PER_STEP * args.warmup * (n if weak else 1)
    if warm_spp:
        render_range(0, warm_spp)
    sync()
    if dist is not None and not rehearsal and args.warmup > 0:
        reduce_framebuffer(fb, dst=0)   # untimed: RCCL sets its rings and kernels up on the first reduce of this shape
        sync()

    # ---- the timed region: exactly K steps, per-launch profiling off ----
    r.set_profiling(False)
    dt = timed_pass()
    ctr = r.counters()
    fb_sum = float(fb[..., :3].double().sum().item()) if rank == 0 else 0.0
    # ---- the same K steps again with HIP events around every launch (roofline block) ----
    r.set_profiling(True)
    dt_prof = timed_pass()
    kt = r.kernel_times()
    r.set_profiling(False)

    red_dev = "cpu" if rehearsal else f"cuda:{local_rank}"
    rays = torch.tensor([ctr["rays"], ctr["camera_samples"], ctr["extension_rays"], ctr["extension_hits"],
                         ctr["shadow_rays"]], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt, dt_prof], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_rays, total_samples, E, H, S = [float(x) for x in rays.tolist()]
    dt, dt_prof = [float(x) for x in tmax.tolist()]

    if rank == 0:
        # ---- roofline of the dominant kernel (rank 0's own launches, HIP events on the launching streams) ----
        model_bytes = {
            "extend": BYTES_PER_EXT_RAY_EXTEND * kt["extend"]["units"],
            "connect": BYTES_PER_SHADOW_RAY_CONNECT * kt["connect"]["units"],
            "shade": shade_bytes(ctr["extension_rays"], ctr["extension_hits"], ctr["shadow_rays"]),
        }
        dom = max(("extend", "shade", "connect"), key=lambda k: kt[k]["ms"])
        launches = max(kt[dom]["launches"], 1)
        avg_ms = kt[dom]["ms"] / launches
        achieved = (model_bytes[dom] / launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        fd = file_derived_profile(dom)
        # HBM roof from the counters: measured bytes per launch (PMC passes of the same build, profiles/) over this run's
        # own launch duration
        hbm_meas = None
        if fd["traffic"]:
            g = fd["traffic"] / (avg_ms * 1e-3) / 1e9
            hbm_meas = {"achieved": round(g, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(g / HBM_PEAK_GBS, 5),
                        "bytes_per_launch": fd["traffic"], "source": fd["traffic_source"],
                        "note": "traffic from the PMC passes of tools/profile_gpu.sh (not measured by this run) / this run's HIP-event launch time"}
        # vector ALU roof: wave64 vector instructions per second of the dominant kernel against one instruction per SIMD
        # every 2 cycles; instructions per wave come from the PMC pass of the same build, launch time from this run.
        # A wave processes 64 units (rays) per chunk pass; instructions per unit = valu_per_wave / (units per wave).
        valu = None
        if fd["valu"]:
            v = dict(fd["valu"])
            v["peak_wave_insts_per_s"] = VALU_PEAK_WAVE_INSTS
            v["achieved_wave_insts_per_s_profiled_pass"] = v.pop("valu_insts_per_s_profiled")
            v["frac"] = round(v["achieved_wave_insts_per_s_profiled_pass"] / VALU_PEAK_WAVE_INSTS, 4)
            v["note"] = ("instruction counts from the serialised PMC pass of tools/profile_gpu.sh (not measured by this run); "
                         "peak = 1024 SIMDs x 2.4 GHz / 2 cycles, the rate of FP32 mul/add/fma only")
            valu = v
        hbm_frac = achieved / HBM_PEAK_GBS
        bound = "hbm"
        if valu and valu["frac"] > max(hbm_frac, hbm_meas["frac"] if hbm_meas else 0.0):
            bound = "valu"
        roofline = {"bound": bound, "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(hbm_frac, 5), "traffic": fd["traffic"],
                    "avg_launch_ms": round(avg_ms, 5), "launches": launches,
                    "bytes_per_launch_model": round(model_bytes[dom] / launches, 1),
                    "kernel_ms": {k: round(v["ms"], 3) for k, v in kt.items()},
                    "kernel_launches": {k: v["launches"] for k, v in kt.items()},
                    "profiled_pass_ms_per_step": round(dt_prof / args.steps * 1e3, 4),
                    "hbm_measured": hbm_meas,
                    "valu": valu,
                    "pipeline_model_GBps": round(total_rays / dt * PIPELINE_BYTES_PER_RAY_16x9 / 1e9, 2),
                    "pipeline_model_frac": round(total_rays / dt * PIPELINE_BYTES_PER_RAY_16x9 / 1e9 / (HBM_PEAK_GBS * n), 5)}

        configs = None
        if n == 1 and not args.no_configs and not os.environ.get("PT_BENCH_SIZE"):
            configs = []
            for name, sf, w, h, spp, st in (("configs[2]", "cornell_box_small_lights.json", 1920, 1080, 16, 4),
                                            ("configs[3]", "cornell_box_with_volume.json", 1920, 1080, 16, 4),
                                            ("configs[4] frame, 1 GPU", "cornell_box.json", 3840, 2160, 4, 4)):
                try:
                    configs.append(sub_config(pt, name, sf, w, h, spp, st, local_rank))
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
            cpu = {"value": round(octr["rays"] / cdt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "sample": f"cornell_box {WIDTH}x{HEIGHT} x {spp_cpu} spp ({octr['rays']} rays), oracle stream mode, "
                             f"{cores} threads, {cdt:.2f} s wall",
                   "reference_order_1thread": {"value": round(mctr["rays"] / mdt / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                                               "sample": f"cornell_box 200x200x16 ({mctr['rays']} rays), oracle mt19937 mode, "
                                                         f"{mdt:.2f} s wall"}}
            if n == 1:
                # parity reported with the metric, at the bench's full frame size: the same 64 spp on the
                # GPU must be the oracle's framebuffer bit for bit, with equal path counters
                import numpy as np
                r.set_device_framebuffer(None, 0)
                r.clear()
                r.render_async(0, spp_cpu)
                gfb = r.framebuffer()
                gctr = r.counters()
                same = (gfb.view(np.uint32) == ofb.view(np.uint32)) | (gfb == ofb)
                parity = {"against": "oracle stream mode, same seed", "size": f"{WIDTH}x{HEIGHT}x{spp_cpu}",
                          "pixel_channels": int(same.size), "mismatched": int((~same).sum()), "tolerance_ulp": 0,
                          "counters_equal": all(gctr[g] == octr[o] for g, o in (
                              ("rays", "rays"), ("extension_rays", "ext_rays"), ("extension_hits", "ext_hits"),
                              ("shadow_rays", "shadow_rays"), ("term_miss", "term_miss"), ("term_rr", "term_rr"),
                              ("term_emitter", "term_emitter"), ("term_pdf", "term_pdf"),
                              ("term_bounce_limit", "term_bounce_limit")))}

        knobs = {k: v for k, v in os.environ.items() if k.startswith("PATHTRACE_HIP_") or k.startswith("PT_BENCH_")}
        out = {
            "metric": "Mrays/sec + achieved HBM GB/s %peak, cornell_box 1080p@1024spp, 1/2/4/8 GPU",
            "value": round(total_rays / dt / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"scenes/{os.path.basename(args.scene)} {WIDTH}x{HEIGHT}, iterative NEE path tracing, max_bounces 10, "
                                   f"light_samples 4, russian roulette; step = {SPP_PER_STEP} spp over the frame"
                                   + (f" x {n} (weak)" if weak else "") + f" (K=64 is 1024 spp), {spp_launch} spp per launch per rank",
                       "spp_total": total_spp, "camera_samples": int(total_samples), "rays": int(total_rays),
                       "rays_per_sample": round(total_rays / max(total_samples, 1), 4),
                       "E_H_S_per_sample": [round(E / total_samples, 4), round(H / total_samples, 4), round(S / total_samples, 4)],
                       "framebuffer_sum": fb_sum,
                       "partition": "whole frame" if n == 1 else (f"128x128 tiles, spiral order, " + ("tile k -> rank k mod N" if os.environ.get("PT_BENCH_ROUND_ROBIN") == "1" else "cost-balanced ownership (LPT over per-tile ray counts)") + "; 1 RCCL reduce"),
                       "partition_setup_ms": round(setup_ms, 1),
                       "knobs": knobs},
            "roofline": roofline,
            "configs": configs,
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
