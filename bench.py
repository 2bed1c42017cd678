#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the per-pixel NEE path-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): scenes/cornell_box.json, 1920x1080, max_bounces 10, light_samples 4, russian
roulette on, normal_offset 1e-4.  One *step* = one wavefront pass over the whole frame at SPP_PER_STEP*N = 16 N samples
per pixel (one batch of ~33 M camera samples per GPU); the default K = 64 steps at N = 1 is exactly the 1024 spp of
configs[1].  "ray" = one World::hit query, extension + shadow, the reference's own unit (integrator.h:192,247).

N > 1: one process per GPU; the image is partitioned by 128x128 tile in NaiveSpiral order, ownership
balanced over measured per-tile ray counts (pathtrace_amd/distributed.py; PT_BENCH_ROUND_ROBIN=1 = tile k -> rank
k mod N), every rank renders its tiles with no communication, and ONE RCCL sum-reduce of the framebuffer to rank 0
ends the timed region.  Per-GPU work is held fixed as N grows (samples per step scale with N): "weak".

The timed region holds only GPU work on device-resident data (scene tables + streams live in HBM; there are no host
buffers on the path).  Rank 0 also reports:
  roofline      the dominant kernel's algorithmic bytes / its HIP-event time, against 8 TB/s (SURVEY.md 8d byte model)
  cpu_baseline  the oracle (CPU restatement, stream mode, all host cores) on a bounded sample of the same workload
"""
import argparse
import json
import os
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
# SURVEY.md 8(d) byte model, per unit of each kernel
BYTES_PER_EXT_RAY_EXTEND = 32 + 16          # read ray, write hit
BYTES_PER_SHADOW_RAY_CONNECT = 48 + 24      # read shadow record, radiance RMW
PIPELINE_BYTES_PER_RAY_16x9 = 153.0         # whole pipeline, cornell_box 16:9


def shade_bytes(E, H, S):
    # read hit+ray+state 96 per E, write state 48 per H, continuation ray 32 per (E - samples), shadow record 48 per S
    return 96 * E + 48 * H + 48 * S


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scene", default=SCENE)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    if world != n:
        if world == 1 and n > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
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

    scene = pt.Scene(args.scene, WIDTH, HEIGHT)
    spp_step = SPP_PER_STEP * n
    from pathtrace_amd.distributed import measure_tile_costs, reduce_framebuffer, tiles_for_rank
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
    # path slots for exactly one step of this rank's pixels: one wavefront batch per step (at N = 1: W*H*16 = 33 M)
    my_pixels = sum((x1 - x0) * (y1 - y0) for (x0, y0, x1, y1) in my_tiles)
    r = pt.Renderer(scene, device=local_rank, seed=0, max_paths_in_flight=my_pixels * spp_step)
    # render straight into a torch tensor so that the final reduce needs no copy
    fb = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    r.set_device_framebuffer(fb.data_ptr(), fb.numel() * 4)

    def step(i):
        r.render_tiles_async(my_tiles, i * spp_step, (i + 1) * spp_step)

    def sync():
        r.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync()
    if dist is not None and not rehearsal and args.warmup > 0:
        reduce_framebuffer(fb, dst=0)   # untimed: RCCL sets its rings and kernels up on the first reduce of this shape
        sync()
    r.clear()
    r.set_profiling(True)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    r.wait()
    if rehearsal and dist is not None:
        fb_host = fb.cpu()
        reduce_framebuffer(fb_host, dst=0)
        fb.copy_(fb_host)
    else:
        reduce_framebuffer(fb, dst=0)   # the one exchange of the path; no-op at N = 1
    sync()
    dt = time.perf_counter() - t0

    ctr = r.counters()
    kt = r.kernel_times()
    red_dev = "cpu" if rehearsal else f"cuda:{local_rank}"
    rays = torch.tensor([ctr["rays"], ctr["camera_samples"], ctr["extension_rays"], ctr["extension_hits"],
                         ctr["shadow_rays"]], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_rays, total_samples, E, H, S = [float(x) for x in rays.tolist()]
    dt = float(tmax.item())

    if rank == 0:
        # ---- roofline of the dominant kernel (rank 0's own launches, HIP events on the render stream) ----
        model_bytes = {
            "extend": BYTES_PER_EXT_RAY_EXTEND * kt["extend"]["units"],
            "connect": BYTES_PER_SHADOW_RAY_CONNECT * kt["connect"]["units"],
            "shade": shade_bytes(ctr["extension_rays"], ctr["extension_hits"], ctr["shadow_rays"]),
        }
        dom = max(("extend", "shade", "connect"), key=lambda k: kt[k]["ms"])
        launches = max(kt[dom]["launches"], 1)
        avg_ms = kt[dom]["ms"] / launches
        achieved = (model_bytes[dom] / launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # the binding resource is the vector ALU, not HBM: share of the one-wide VALU issue rate this kernel sustained in
        # the serialised counter pass (tools/profile_gpu.sh pass 4 -> profiles/*_instruction_mix.json)
        valu = None
        try:
            import glob
            mixes = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_instruction_mix.json")))
            if mixes:
                mk = json.load(open(mixes[-1])).get(dom)
                if mk:
                    valu = {"valu_issue_fraction": mk["valu_issue_fraction"], "valu_per_wave": mk["valu_per_wave"],
                            "salu_over_valu": mk["salu_over_valu"], "effective_clock_GHz": mk["effective_clock_GHz"],
                            "source": "profiles/" + os.path.basename(mixes[-1])}
        except Exception:
            valu = None
        # the same kernel without a second batch sharing the chip: a short extra pass on a single-lane context (the timed
        # region above runs two batches in flight, which is faster overall but stretches every individual launch)
        solo = None
        try:
            os.environ["PATHTRACE_HIP_LANES"] = "1"
            r1 = pt.Renderer(scene, device=local_rank, seed=0, max_paths_in_flight=my_pixels * spp_step)
            r1.render_tiles_async(my_tiles, 0, spp_step)
            r1.wait()
            r1.set_profiling(True)
            for i in range(1, 4):
                r1.render_tiles_async(my_tiles, i * spp_step, (i + 1) * spp_step)
            r1.wait()
            k1 = r1.kernel_times()[dom]
            per_unit = {"extend": BYTES_PER_EXT_RAY_EXTEND, "connect": BYTES_PER_SHADOW_RAY_CONNECT}.get(dom)
            b1 = per_unit * k1["units"] if per_unit else shade_bytes(*[r1.counters()[x] * 3 / 4 for x in ("extension_rays", "extension_hits", "shadow_rays")])
            solo_gbs = b1 / (k1["ms"] * 1e-3) / 1e9
            solo = {"achieved": round(solo_gbs, 2), "frac": round(solo_gbs / HBM_PEAK_GBS, 5),
                    "avg_launch_ms": round(k1["ms"] / max(k1["launches"], 1), 5), "launches": k1["launches"]}
            r1.close()
        except Exception as e:   # the headline number does not depend on this pass
            solo = {"error": str(e)}
        finally:
            os.environ.pop("PATHTRACE_HIP_LANES", None)
        roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 5), "launches": launches,
                    "bytes_per_launch_model": round(model_bytes[dom] / launches, 1),
                    "kernel_ms": {k: round(v["ms"], 3) for k, v in kt.items()},
                    "single_batch_in_flight": solo,
                    "valu": valu,
                    "pipeline_model_GBps": round(total_rays / dt * PIPELINE_BYTES_PER_RAY_16x9 / 1e9, 2),
                    "pipeline_model_frac": round(total_rays / dt * PIPELINE_BYTES_PER_RAY_16x9 / 1e9 / (HBM_PEAK_GBS * n), 5)}

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
                # parity reported with the metric, at the bench's full frame size: the same 4 spp on the
                # GPU must be the oracle's framebuffer bit for bit, with equal path counters
                import numpy as np
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

        if rehearsal:
            img = fb.cpu().numpy()[..., :3]
            print(json.dumps({"rehearsal_fb_sum": float(img.astype("float64").sum()), "spp_total": spp_step * args.steps,
                              "nonzero_pixels": int((img.sum(axis=2) != 0).sum())}), file=sys.stderr, flush=True)
        out = {
            "metric": "Mrays/sec + achieved HBM GB/s %peak, cornell_box 1080p@1024spp, 1/2/4/8 GPU",
            "value": round(total_rays / dt / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"scenes/{os.path.basename(args.scene)} {WIDTH}x{HEIGHT}, iterative NEE path tracing, max_bounces 10, "
                                   "light_samples 4, russian roulette, %d spp per step per frame (K=64 at N=1 is 1024 spp)" % spp_step,
                       "spp_total": spp_step * args.steps, "camera_samples": int(total_samples), "rays": int(total_rays),
                       "rays_per_sample": round(total_rays / max(total_samples, 1), 4),
                       "E_H_S_per_sample": [round(E / total_samples, 4), round(H / total_samples, 4), round(S / total_samples, 4)],
                       "partition": "whole frame" if n == 1 else (f"128x128 tiles, spiral order, " + ("tile k -> rank k mod N" if os.environ.get("PT_BENCH_ROUND_ROBIN") == "1" else "cost-balanced ownership (LPT over per-tile ray counts)") + "; 1 RCCL reduce")},
            "roofline": roofline,
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
