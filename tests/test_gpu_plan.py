"""The launch plan behind the C ABI (include/pathtrace_hip.h, ABI v6) on the GPU: a context created with
max_paths_in_flight = 0 -- what both plugin bindings pass -- is sized by the library from the render calls, cuts them by the
library's rule, grows when a later call wants more, and renders the same bits as any explicit size.  And the drop-in surface
reaches the rate of a harness that drives the library directly: renderer.h:553-603 gives its user full speed from config.json
alone, so must `pth_main` / `HipWavefront`."""
import json
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

import pathtrace_amd as pt
from conftest import ROOT, scene_path
from test_gpu_main import _workdir

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_auto_sized_context_plans_grows_and_renders_the_same_bits():
    w, h = 256, 144
    sc = pt.Scene(scene_path("cornell_box"), w, h)
    ref = pt.Renderer(sc, seed=3, max_paths_in_flight=w * h * 2)      # explicit size, as up to ABI v5
    want = ref.render(12)
    want_c = ref.counters()
    assert ref.plan()["auto_sized"] == 0 and ref.plan()["path_slots"] >= w * h * 2
    ref.close()
    r = pt.Renderer(sc, seed=3)                                         # max_paths_in_flight = 0
    p = r.plan()
    assert p["auto_sized"] == 1 and p["path_slots"] == 0 and p["stream_bytes"] == 0      # nothing allocated yet
    r.render_async(0, 4)                                                # the first call sizes the streams: 2 batches of 2 spp
    p = r.plan()
    assert (p["pixels"], p["samples"], p["spp_per_batch"], p["batches"]) == (w * h, 4, 2, 2)
    assert w * h * 2 <= p["path_slots"] < w * h * 2 + 8192 and p["grown"] == 0
    assert p["stream_bytes"] >= p["path_slots"] * 288 * p["lanes"] and p["hbm_free_bytes"] > 0
    r.render_async(4, 12)                                               # 8 more samples: the plan wants 4 per batch -> the streams grow
    p = r.plan()
    assert (p["samples"], p["spp_per_batch"], p["batches"]) == (8, 4, 2) and p["grown"] == 1 and p["path_slots"] >= w * h * 4
    got = r.framebuffer()
    assert np.array_equal(bits(got), bits(want)) and r.counters() == want_c
    secs = r.render_seconds()
    assert 0 < secs < 30
    assert r.wait_for(1000) is True
    r.clear()
    r.prime(5)                                                          # pt_prime: a warm-up that leaves nothing behind
    assert r.counters()["camera_samples"] == 0 and not r.framebuffer().any() and r.plan()["samples"] == 8
    r.reserve(w * h, 64)                                                # pt_reserve: sized up front, no growth inside the render
    g = r.plan()["grown"]
    r.render_async(0, 64)
    p = r.plan()
    assert p["grown"] == g and (p["spp_per_batch"], p["batches"]) == (32, 2)
    r.close()


def test_tiles_and_planner_on_an_auto_sized_context():
    w, h = 320, 180
    sc = pt.Scene(scene_path("cornell_box_small_lights"), w, h)
    tiles = pt.spiral_tiles(w, h, 64, 64)
    a = pt.Renderer(sc, seed=1, max_paths_in_flight=w * h * 4)
    costs_a = a.measure_tile_costs(tiles)
    a.render_tiles_async(tiles[::2], 0, 6)
    fa, ca = a.framebuffer(), a.counters()
    a.close()
    b = pt.Renderer(sc, seed=1)
    costs_b = b.measure_tile_costs(tiles)                                # sizes the streams for one sample per pixel ...
    b.render_tiles_async(tiles[::2], 0, 6)                               # ... and this call re-plans within / beyond them
    assert costs_a == costs_b
    assert np.array_equal(bits(b.framebuffer()), bits(fa)) and b.counters() == ca
    b.close()


def _rates(txt):
    m = re.search(r"traced (\d+) rays, at ([0-9.eE+-]+) rays per second", txt)
    t = re.search(r"time taken to compute ([0-9.eE+-]+)", txt)
    lp = re.search(r"launch plan: (\d+) batches of (\d+) spp", txt)
    assert m and t and lp, txt[-2000:]
    return int(m.group(1)), float(m.group(2)), float(t.group(1)), (int(lp.group(1)), int(lp.group(2)))


def test_the_drop_in_surface_runs_at_the_harness_rate(tmp_path, monkeypatch):
    """BASELINE config 2's frame at 256 spp through pth_main (a child process: config.json in, the reference's own statistics
    lines out) and, when the build container left it, through the reference-side plugin: the rate they print is within 3 % of
    what this process measures driving the library directly with the same single call, the plan is the library's (3 batches of 86
    spp), and the reference-side object's framebuffer is this process's, bit for bit."""
    monkeypatch.setenv("PATHTRACE_HIP_SPEC", "async")                   # the production default: the scene's own kernels
    w, h, spp = 1920, 1080, 256
    sc = pt.Scene(scene_path("cornell_box"), w, h)
    r = pt.Renderer(sc, seed=0)
    r.reserve(w * h, spp)
    r.spec_wait()
    r.prime(50)                                                         # the warm-up the plugin's constructor does
    best = None
    for _ in range(3):
        r.clear()
        t0 = time.perf_counter()
        r.render_async(0, spp)
        r.wait()
        dt = time.perf_counter() - t0
        assert abs(r.render_seconds() - dt) < 0.02 * dt + 2e-3          # the library's clock of the call is this wall time
        best = dt if best is None else min(best, dt)
    fb = r.framebuffer()
    c = r.counters()
    p = r.plan()
    r.close()
    want_plan = pt.plan_batches(w * h, spp, pt.PLAN_MAX_PATHS)
    assert (p["spp_per_batch"], p["batches"]) == want_plan == (86, 3)
    harness = c["rays_traced"] / best
    wd, cfg = _workdir(tmp_path, "cornell_box", film={"width": w, "height": h, "exposure": 0.0, "gamma": 2.2}, samples=spp)
    code = ("import sys; sys.path.insert(0, %r); import pathtrace_amd as pt; rc = pt.lib().pth_main(%r.encode()); sys.stdout.flush(); sys.exit(1 if rc else 0)"
            % (ROOT, str(wd)))
    rates = []
    for _ in range(2):
        cp = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        assert cp.returncode == 0, (cp.stdout[-1000:], cp.stderr[-2000:])
        traced, rate, secs, plan = _rates(cp.stdout)
        assert traced == c["rays_traced"] and plan == (3, 86)
        rates.append(rate)
    assert max(rates) >= 0.97 * harness, (rates, harness)
    driver = os.path.join(ROOT, "oracle", "_ref", "plugin_driver")
    if os.path.exists(driver):
        from oracle import scene_params as sp
        params = sp.load_scene_params(json.load(open(scene_path("cornell_box"))), base_dir=ROOT)
        ptxt, out = tmp_path / "scene.params", tmp_path / "fb.f32"
        ptxt.write_text(sp.to_text(params))
        prates = []
        for _ in range(2):
            q = subprocess.run([driver, str(ptxt), "render", str(w), str(h), str(spp), "10", "4", "1", "0.0001", "0", "128", "128", str(out)],
                               capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
            assert q.returncode == 0, (q.stdout[-1000:], q.stderr[-2000:])
            traced, rate, secs, plan = _rates(q.stdout)
            assert traced == c["rays_traced"] and plan == (3, 86)
            prates.append(rate)
        got = np.fromfile(out, np.float32).reshape(h, w, 3)
        assert np.array_equal(bits(got), bits(fb))
        assert max(prates) >= 0.97 * harness, (prates, harness)


def test_reference_side_plugin_on_two_contexts(tmp_path, monkeypatch):
    """The reference-side HipWavefront's multi-device form (config.threads > 1 = that many GPUs, capped by the devices present;
    PATHTRACE_HIP_DEVICES overrides), rehearsed with two contexts on device 0: every device driven by its own host thread, the
    owned tiles exchanged at the end -- the framebuffer the reference's Renderer ends up holding is the one-context one."""
    driver = os.path.join(ROOT, "oracle", "_ref", "plugin_driver")
    if not os.path.exists(driver):
        pytest.skip("oracle/_ref/plugin_driver was not built (needs the reference tree at build time)")
    from oracle import scene_params as sp
    w, h, spp = 400, 225, 6
    out = {}
    for scene in ("cornell_box", "cornell_box_with_volume"):
        params = sp.load_scene_params(json.load(open(scene_path(scene))), base_dir=ROOT)
        ptxt = tmp_path / f"{scene}.params"
        ptxt.write_text(sp.to_text(params))
        for tag, env, threads in (("one", {}, "1"), ("threads8", {}, "8"), ("two", {"PATHTRACE_HIP_DEVICES": "0,0"}, "1")):
            f = tmp_path / f"{scene}_{tag}.f32"
            q = subprocess.run([driver, str(ptxt), "render", str(w), str(h), str(spp), "10", "4", "1", "0.0001", "0", "64", "64", str(f), threads],
                               capture_output=True, text=True, cwd=str(tmp_path), timeout=300, env=dict(os.environ, **env))
            assert q.returncode == 0, (q.stdout[-1000:], q.stderr[-2000:])
            out[tag] = np.fromfile(f, np.float32).reshape(h, w, 3), q.stdout
        assert "per device" in out["one"][1]
        assert np.array_equal(bits(out["one"][0]), bits(out["two"][0]))
        assert np.array_equal(bits(out["one"][0]), bits(out["threads8"][0]))   # threads = 8 on a one-GPU box: capped to the one device
        sc = pt.Scene(scene_path(scene), w, h)
        r = pt.Renderer(sc, seed=0)
        assert np.array_equal(bits(r.render(spp)), bits(out["one"][0]))
        r.close()
