"""The per-scene build of the traversal sweep on the GPU (pt_spec.cpp): pt_create compiles k_extend / k_connect / k_trace once
more for the scene at hand (hiprtc) and launches them through the module API.  They must be the generic kernels bit for
bit: framebuffers and counters against the oracle, World::hit on boundary rays against the generic sweep, a render that
starts on the generic kernels and switches to the module when it is ready, and the silent fallback of a failed build."""
import numpy as np
import pytest

import pathtrace_amd as pt
from conftest import ALL_SCENES, SCENES, scene_path
from test_gpu_parity import assert_bit_identical, assert_counters, bits, oracle_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["extend", "extend+connect"])
def spec_sync(monkeypatch, request):
    # the module serves k_extend, k_connect and k_trace; PATHTRACE_HIP_SPEC=sync,extend-only leaves shadow rays on the generic k_connect
    monkeypatch.setenv("PATHTRACE_HIP_SPEC", "sync" if request.param == "extend+connect" else "sync,extend-only")


@pytest.mark.parametrize("scene", ALL_SCENES)
def test_specialised_sweep_is_bit_exact_vs_oracle(oracle, scene, spec_sync):
    w, h, spp = 96, 54, 6
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc, seed=5)
    if scene == "cornell_box_nested_fog":                  # a medium whose boundary is a medium runs the general sweep: no fast program
        assert r.spec_status() == -1                       # to specialise, the generic kernels render
    else:
        assert r.spec_status() == 1, pt.last_error()      # built inside pt_create, module loaded
    fb = r.render(spp)
    ctr = r.counters()
    r.close()
    ref, oc = oracle.Scene.from_json(scene_path(scene)).render_stream(oracle_cfg(oracle, w, h, spp), seed=5)
    assert_bit_identical(fb, ref, scene)
    assert_counters(ctr, oc, scene)


@pytest.mark.parametrize("scene,kw", [("cornell_box", dict(light_samples=3)), ("cornell_box_small_lights", dict(light_samples=2, max_bounces=4)),
                                      ("cornell_box_with_volume", dict(light_samples=1, russian_roulette=False, max_bounces=6)),
                                      ("light_test", dict(light_samples=5))])
def test_specialised_sweep_config_variants(oracle, scene, kw, spec_sync):
    w, h, spp = 80, 60, 4
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc, seed=1, **kw)
    assert r.spec_status() == 1, pt.last_error()
    fb = r.render(spp)
    ctr = r.counters()
    r.close()
    cfg = oracle_cfg(oracle, w, h, spp, **kw)
    ref, oc = oracle.Scene.from_json(scene_path(scene)).render_stream(cfg, seed=1)
    assert_bit_identical(fb, ref, scene)
    assert_counters(ctr, oc, scene)


@pytest.mark.parametrize("scene", SCENES + ["three_orbs"])
def test_specialised_world_hit_equals_the_generic_sweep_on_boundary_rays(scene, monkeypatch):
    from test_gpu_traversal import tame_rays
    sc = pt.Scene(scene_path(scene), 64, 64)
    o, d = tame_rays(np.random.default_rng(5), sc, n_random=20000)     # every wave takes the fast sweep: the code that was rebuilt
    assert len(o) > 20000
    monkeypatch.setenv("PATHTRACE_HIP_SPEC", "off")
    g = pt.Renderer(sc, max_paths_in_flight=4096)
    assert g.spec_status() == -1
    monkeypatch.setenv("PATHTRACE_HIP_SPEC", "sync")
    s = pt.Renderer(sc, max_paths_in_flight=4096)
    assert s.spec_status() == 1, pt.last_error()
    k0, k1, vd = 0x7654321, 0x0fedcba9, 24
    for nr in (1, 2, 4):
        nn = (len(o) // (256 * nr)) * 256 * nr
        on, dn = (o, d) if nr == 1 else (o[:nn:nr], d[:nn].reshape(-1, nr, 3))
        tg, ig = g.trace_rays(on, dn, k0, k1, vd)
        ts, is_ = s.trace_rays(on, dn, k0, k1, vd)
        assert np.array_equal(ig, is_) and np.array_equal(bits(tg), bits(ts)), (scene, nr)
        assert (ig >= 0).sum() > 2000
    g.close()
    s.close()


def test_a_render_switches_to_the_module_when_it_is_ready(monkeypatch):
    # async (the default outside this suite): pt_create returns at once, the first batches run the generic kernels, later
    # ones the module; the image is the one either of them alone renders
    scene, w, h, spp = "cornell_box_small_lights", 160, 90, 12
    sc = pt.Scene(scene_path(scene), w, h)
    monkeypatch.setenv("PATHTRACE_HIP_SPEC", "off")
    ref_r = pt.Renderer(sc, seed=8, max_paths_in_flight=w * h)
    ref = ref_r.render(spp)
    rc = ref_r.counters()
    ref_r.close()
    monkeypatch.setenv("PATHTRACE_HIP_SPEC", "async")
    r = pt.Renderer(sc, seed=8, max_paths_in_flight=w * h)      # one batch per sample
    seen = set()
    for s in range(spp):
        if s == spp // 2:
            assert r.spec_wait() == 1, pt.last_error()           # from here on the module
        seen.add(r.spec_status())
        r.render_async(s, s + 1)
    assert 1 in seen
    assert np.array_equal(bits(r.framebuffer()), bits(ref)) and r.counters() == rc
    r.close()


def test_a_failed_build_falls_back_to_the_generic_kernels(monkeypatch):
    scene, w, h, spp = "three_orbs", 96, 54, 4
    sc = pt.Scene(scene_path(scene), w, h)
    monkeypatch.setenv("PATHTRACE_HIP_SPEC", "off")
    g = pt.Renderer(sc, seed=2)
    ref = g.render(spp)
    g.close()
    monkeypatch.setenv("PATHTRACE_HIP_SPEC", "sync")
    monkeypatch.setenv("PATHTRACE_HIP_SPEC_BREAK", "1")
    r = pt.Renderer(sc, seed=2)                                   # pt_create succeeds, silently
    assert r.spec_status() == -1 and "fails on purpose" in pt.last_error()
    assert np.array_equal(bits(r.render(spp)), bits(ref))
    r.close()


@pytest.mark.parametrize("light_samples", [4, 3])
def test_walled_rooms_match_the_oracle_on_the_module(oracle, light_samples, monkeypatch):
    # tests/scene_gen.py room_scene: closed rooms whose walls the module's k_connect proves unreachable for a shadow ray instead of
    # testing them (world_hit_fast_rb SHADOW) -- with lights from 0.05 to 40 units under the ceiling (the marking flips between
    # them), a partition that must stay a tested leaf, one-sided lights, fog, the camera inside or outside.  Bit for bit the
    # oracle: framebuffer and counters; and the same through the generic kernels.
    import json
    from scene_gen import room_scene
    from test_gpu_fuzz import CTR

    for seed in range(10):
        js = room_scene(seed, clutter=30 if seed >= 8 else 0)   # the last two: 40 instances, the fast program keeps its tree of boxes
        sc = pt.Scene(text=json.dumps(js), width=56, height=40)
        osc = oracle.Scene(oracle.sp.load_scene_params(js))
        o, oc = osc.render_stream(oracle.make_config(56, 40, 3, light_samples=light_samples), seed=seed, threads=2)
        for spec in ("sync", "off"):
            monkeypatch.setenv("PATHTRACE_HIP_SPEC", spec)
            r = pt.Renderer(sc, seed=seed, light_samples=light_samples)
            assert r.spec_status() == (1 if spec == "sync" else -1), pt.last_error()
            g = r.render(3)
            gc = r.counters()
            r.close()
            assert ((bits(g) == bits(o)) | (g == o)).all(), (seed, spec)
            for a, b in CTR.items():
                assert gc[a] == oc[b], (seed, spec, a, gc[a], oc[b])
