"""Seeded sweep over ragged configurations: odd sizes down to one pixel, one sample, zero to a few bounces, light_samples
that do and do not divide into ray pairs, tiny path-slot budgets (many ragged batches), every scene.  The GPU framebuffer
and all path counters must equal the oracle's (stream mode) bit for bit."""
import os

import numpy as np
import pytest

import pathtrace_amd as pt
from conftest import ALL_SCENES, scene_path

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


CTR = {"rays": "rays", "extension_rays": "ext_rays", "extension_hits": "ext_hits", "shadow_rays": "shadow_rays",
       "term_miss": "term_miss", "term_rr": "term_rr", "term_emitter": "term_emitter", "term_pdf": "term_pdf",
       "term_bounce_limit": "term_bounce_limit"}


def test_ragged_configurations_match_the_oracle(oracle):
    rng = np.random.default_rng(20261003)
    cases = [(1, 1, 1, 1, 1, 0), (1, 1, 3, 0, 2, 64), (2, 1, 1, 10, 4, 0), (1, 7, 2, 2, 3, 5)]
    for _ in range(44):
        cases.append((int(rng.integers(1, 41)), int(rng.integers(1, 41)), int(rng.integers(1, 6)), int(rng.integers(0, 6)),
                      int(rng.integers(1, 7)), int(rng.choice([0, 0, 64, 257, 1000, 5000]))))
    for k, (w, h, spp, mb, ls, slots) in enumerate(cases):
        scene = ALL_SCENES[k % len(ALL_SCENES)]
        rr, od, seed = bool(k % 3), bool(k % 11 == 5), k * 7919
        sc = pt.Scene(scene_path(scene), w, h)
        r = pt.Renderer(sc, seed=seed, max_paths_in_flight=slots, max_bounces=mb, light_samples=ls, russian_roulette=rr, only_direct=od)
        g = r.render(spp)
        gc = r.counters()
        r.close()
        osc = oracle.Scene.from_json(scene_path(scene))
        cfg = oracle.make_config(w, h, spp, max_bounces=mb, light_samples=ls, russian_roulette=rr, only_direct=od)
        o, oc = osc.render_stream(cfg, seed=seed, threads=2)
        what = (scene, w, h, spp, mb, ls, slots, rr, od)
        assert ((bits(g) == bits(o)) | (g == o)).all(), what
        for a, b in CTR.items():
            assert gc[a] == oc[b], (what, a, gc[a], oc[b])
        assert gc["camera_samples"] == w * h * spp


def test_random_scenes_match_the_oracle(oracle):
    # generated scenes (tests/scene_gen.py; the oracle is pinned to the real reference on them): the whole device path --
    # flattening, sweep over BVHs of up to 45 nodes, rotated / scaled instances, sphere lights, fog, textures -- bit for bit
    import json
    from scene_gen import random_scene

    for seed in list(range(0, 24)) + [101, 202, 303, 404, 1001, 1002, 1003, 1004]:
        js = random_scene(seed, nested=seed > 1000)   # the last four: with a medium whose boundary is a medium (general sweep)
        sc = pt.Scene(text=json.dumps(js), width=48, height=36)
        r = pt.Renderer(sc, seed=seed)
        g = r.render(3)
        gc = r.counters()
        r.close()
        osc = oracle.Scene(oracle.sp.load_scene_params(js))
        o, oc = osc.render_stream(oracle.make_config(48, 36, 3), seed=seed, threads=2)
        assert ((bits(g) == bits(o)) | (g == o)).all(), seed
        for a, b in CTR.items():
            assert gc[a] == oc[b], (seed, a, gc[a], oc[b])


def test_large_instance_counts(oracle):
    # 200 instances (a BVH of ~255 nodes, 8 short-stack slots: the deepest the sweep holds) render bit-exactly; beyond
    # that pt_create refuses loudly instead of traversing wrongly
    import json
    from scene_gen import random_scene

    js = random_scene(7, n_inst=200, volume=False)
    sc = pt.Scene(text=json.dumps(js), width=24, height=18)
    r = pt.Renderer(sc, seed=1, light_samples=2)
    g = r.render(2)
    gc = r.counters()
    r.close()
    osc = oracle.Scene(oracle.sp.load_scene_params(js))
    o, oc = osc.render_stream(oracle.make_config(24, 18, 2, light_samples=2), seed=1, threads=2)
    assert ((bits(g) == bits(o)) | (g == o)).all()
    assert all(gc[a] == oc[b] for a, b in CTR.items())
    # 300 instances: a tree deeper than the 8 LDS short-stack slots.  The fast sweep folds leaf by leaf and needs no stack;
    # the general sweep (non-tame waves, NaN-t ties) keeps its partial results in a global scratch stack (pt_kernels.hip
    # stack_of); the per-lane walk (the default above 2048 instances, forced here) keeps its pending children there.  All
    # three must reproduce the oracle.
    js = random_scene(8, n_inst=300, volume=False)
    osc = oracle.Scene(oracle.sp.load_scene_params(js))
    o, oc = osc.render_stream(oracle.make_config(24, 18, 2, light_samples=2), seed=3, threads=2)
    for env in (None, "1", "walk"):   # fast sweep over the tree program / general sweep / per-lane walk
        if env == "1":
            os.environ["PATHTRACE_HIP_TRAVERSAL"] = "general"
        elif env == "walk":
            os.environ["PATHTRACE_HIP_TRAVERSAL"] = "walk"
        try:
            big = pt.Scene(text=json.dumps(js), width=24, height=18)
            rb = pt.Renderer(big, seed=3, light_samples=2)
            g = rb.render(2)
            gc = rb.counters()
            rb.close()
        finally:
            os.environ.pop("PATHTRACE_HIP_TRAVERSAL", None)
        assert ((bits(g) == bits(o)) | (g == o)).all(), env
        assert all(gc[a] == oc[b] for a, b in CTR.items()), env


def test_constant_medium_with_sphere_and_rect_boundaries(oracle):
    # volume.h:10 takes any hittable as the boundary: a sphere (two sphere::hit calls) and a rect (the second boundary hit
    # cannot exist, so the medium is never hit -- but its instance still sits in the BVH)
    import json
    from conftest import scene_path

    base = json.load(open(scene_path("cornell_box_with_volume")))
    for boundary in ({"id": "box", "type": "sphere", "radius": 90.0, "origin": [0.0, 10.0, 0.0]},
                     {"id": "box", "type": "rect", "size": [165, 165], "material": {"id": "white"}}):
        js = json.loads(json.dumps(base))
        js["primitives"] = [boundary if p.get("id") == "box" else p for p in js["primitives"]]
        sc = pt.Scene(text=json.dumps(js), width=64, height=48)
        r = pt.Renderer(sc, seed=5)
        g = r.render(4)
        gc = r.counters()
        r.close()
        osc = oracle.Scene(oracle.sp.load_scene_params(js))
        o, oc = osc.render_stream(oracle.make_config(64, 48, 4), seed=5, threads=2)
        assert ((bits(g) == bits(o)) | (g == o)).all(), boundary["type"]
        assert all(gc[a] == oc[b] for a, b in CTR.items()), boundary["type"]
