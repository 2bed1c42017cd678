"""bench.py as the driver runs it (a plain command): N = 1, and N = 2 started by bench.py itself.
On a one-GPU box the two ranks share GPU 0 (PT_BENCH_REHEARSAL=1: gloo instead of RCCL, the reduce staged through host
memory); the image must be the N = 1 image bit for bit, which the framebuffer checksum and the ray counters show.
Plus one RCCL ("nccl") process group of world size 1 on the GPU: init, the framebuffer reduce of the path, destroy."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu


def run_bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    env.pop("PATHTRACE_HIP_SPEC", None)   # conftest switches the per-scene build off for the suite; bench.py runs as the driver runs it
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_and_matches_one_rank():
    common = ["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-configs"]
    small = {"PT_BENCH_SIZE": "640x360"}
    one = run_bench(["--gpus", "1"] + common, small)
    two = run_bench(["--gpus", "2"] + common, dict(small, PT_BENCH_REHEARSAL="1"))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    for k in ("spp_total", "camera_samples", "rays", "framebuffer_sum"):
        assert one["config"][k] == two["config"][k], (k, one["config"][k], two["config"][k])
    for d in (one, two):
        assert d["roofline"]["bound"] in ("hbm", "valu-issue", "latency") and d["roofline"]["bound_basis"] and d["value"] > 0 and d["steps"] == 4
        if d["roofline"]["bound"] == "latency":   # named from the same-build wait-state counters: a wave parked on s_waitcnt more than it issues and executes
            w = d["roofline"]["wave_life"]
            assert d["roofline"]["parked"] == w["parked"] > w["stalled_at_issue"] + w["executing"]
        # the launch plan is the library's: bench.py passes no size and makes one render call per pass
        lp = d["config"]["launch_plan"]
        assert lp["auto_sized"] == 1 and lp["batches"] >= 2 and "libpathtrace_hip.so" in lp["decided_by"]
        # the fractions the label is chosen from, side by side: the contract's model bytes, this implementation's record bytes
        assert 0 < d["roofline"]["hbm_frac_stream"] <= d["roofline"]["hbm_frac_model"] * 1.5 and d["roofline"]["frac"] == d["roofline"]["hbm_frac_model"]
        assert d["config"]["module"]["sweep"].startswith("per-scene build") and d["config"]["module"]["own_compiler"] is True, d["config"]["module"]
        # value = rays the device traced; the reference's count for the same image is reported beside it
        assert d["value"] <= d["value_reference_equivalent"] and d["config"]["rays_traced"] <= d["config"]["rays"]
        ser = d["roofline"]["serialised"]
        # the roofline block is reproducible from the line alone: one lane, so the kernels' event times do not overlap
        assert ser["lanes"] == 1 and ser["kernel_ms_sum"] <= ser["wall_ms"] * 1.02
        dom = d["roofline"]["kernel"][2:]
        assert abs(d["roofline"]["achieved"] - ser["kernels"][dom]["model_bytes"] / ser["kernels"][dom]["ms"] / 1e6) <= 0.01 * d["roofline"]["achieved"] + 0.1
    # N = 1 carries the one-GPU strong-scaling proxy: every rank of 2 / 4 / 8 renders the same total work
    px = one["scaling_proxy"]
    assert [e["n"] for e in px["by_n"]] == [2, 4, 8]
    for e in px["by_n"]:
        assert e["rays_all_ranks"] == one["config"]["rays"] and min(e["batches_per_rank"]) >= 2 and e["predicted_efficiency"] > 0
    assert two["scaling_proxy"] is None
    # the plugin surface (pth_main in a child process, config.json in) rendered the same frame and samples with the library's plan
    pm = one["plugin_path"]["pth_main"]
    assert pm["rc"] == 0 and pm["rays_traced"] == one["config"]["rays_traced"] and pm["launch_plan"]["batches"] == one["config"]["launch_plan"]["batches"]
    assert pm["value"] > 0 and two["plugin_path"] is None


def test_bench_four_rank_rehearsal_of_the_launch_path():
    # The launch path of the multi-GPU bench with as many ranks as one GPU box safely lets share its card (its process guard
    # allows six processes on the GPU, this test runner is one of them; eight ranks are the driver's to run on eight GPUs; the
    # eight-way ownership map and exchange are covered on the CPU, tests/test_distributed_gloo.py): bench.py starts torch.distributed.run itself, every rank runs the one-pass planner, derives
    # the same ownership map, waits for its own per-scene build, renders its tiles and the owned-tile gather assembles the frame
    # on rank 0 -- rays, samples and the framebuffer checksum equal N = 1.
    common = ["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-configs", "--no-scaling-proxy"]
    small = {"PT_BENCH_SIZE": "640x360"}
    one = run_bench(["--gpus", "1"] + common, small)
    four = run_bench(["--gpus", "4"] + common, dict(small, PT_BENCH_REHEARSAL="1"))
    assert four["n_gpus"] == 4 and four["scaling"] == "strong" and "gather of owned tiles" in four["config"]["exchange"]
    for k in ("spp_total", "camera_samples", "rays", "rays_traced", "framebuffer_sum"):
        assert one["config"][k] == four["config"][k], (k, one["config"][k], four["config"][k])
    # the exchange moved the other ranks' tiles (rank 0's own stay): less than one frame, not three frames
    moved = int(four["config"]["exchange"].split(":")[1].split()[0])
    assert 0 < moved < 640 * 360 * 16


def test_rccl_world_size_one_reduce_of_the_library_framebuffer():
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import pathtrace_amd as pt
from pathtrace_amd.distributed import reduce_framebuffer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sc = pt.Scene(os.path.join(%r, "scenes", "cornell_box.json"), 128, 72)
r = pt.Renderer(sc, device=0)
fb = torch.zeros((72, 128, 4), dtype=torch.float32, device="cuda:0")
r.set_device_framebuffer(fb.data_ptr(), fb.numel() * 4)
r.render_async(0, 4); r.wait()
before = fb.clone()
dist.reduce(fb, dst=0, op=dist.ReduceOp.SUM)          # RCCL kernel on the library-rendered tensor
dist.all_reduce(fb, op=dist.ReduceOp.SUM)              # world size 1: identity
torch.cuda.synchronize()
assert torch.equal(fb, before) and float(fb.sum()) > 0
r.set_device_framebuffer(None, 0)
ref = pt.Renderer(sc, device=0).render(4)
assert np.array_equal(before.cpu().numpy()[..., :3], ref)
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK")
""" % (ROOT, ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-1000:], p.stderr[-2000:])
