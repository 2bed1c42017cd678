"""Unit-level parity of World::hit (world.h:17-20 -> bvh.h:31-69 -> aabb.h:34-53 / primitive.h:186-312 / volume.h:29-93)
on the GPU against the oracle, on rays chosen to hit the corner cases the reference's comparison chains decide:
zero and negative-zero direction components (1/0 slabs, t = +-inf), rays lying IN a rect's plane (0/0 -> NaN t, which the
reference does not reject: SURVEY Q8), rays through exact edges and corners (xh == x0), origins on surfaces (t below
0.001), NaN / huge / tiny components, and plain random rays.  The device traversal states those chains as sign tests of
max/min "excess" values (pt_kernels.hip), so this is where a NaN or inf mismatch would show.

Bar: hit/miss and instance identical, t bit-identical (any NaN == any NaN).  Both ray instantiations are exercised:
1 ray per origin (k_extend's) and 4 rays sharing an origin (k_connect's).
"""
import numpy as np
import pytest

import pathtrace_amd as pt
from conftest import ALL_SCENES, scene_path

pytestmark = pytest.mark.gpu


def adversarial_rays(rng, bbox, n_random=60000):
    F = np.float32
    o, d = [], []
    # rays aimed at random points of the scene's own bounding box (so that every scene gets plenty of hits), from
    # inside and around it, plus axis-aligned +-0 directions from points inside it
    mn, mx = bbox[:3].astype(np.float64), bbox[3:].astype(np.float64)
    ext = mx - mn
    src = mn - 0.5 * ext + rng.random((20000, 3)) * 2.0 * ext
    dst = mn + rng.random((20000, 3)) * ext
    o.append(src)
    d.append((dst - src) * rng.choice([1e-3, 1.0, 37.0], (20000, 1)))
    ins = mn + rng.random((12, 3)) * ext
    zs0 = np.array([0.0, -0.0, 1.0, -1.0, 0.25])
    g0 = np.array(np.meshgrid(zs0, zs0, zs0)).reshape(3, -1).T
    for p in ins:
        o.append(np.repeat(p[None], len(g0), 0))
        d.append(g0)
    # random rays from inside and outside the box
    o.append(rng.uniform(-100, 655, (n_random, 3)))
    d.append(rng.normal(0, 1, (n_random, 3)) * rng.choice([1e-3, 1.0, 1e3], (n_random, 1)))
    # axis-aligned and plane-aligned directions with +0 / -0 components
    zs = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -277.5])
    grid = np.array(np.meshgrid(zs, zs, zs)).reshape(3, -1).T
    pts = np.array([[278, 278, 278], [277.5, 0.0, 277.5], [273, 554, 171], [0, 277.5, 277.5], [555, 277.5, 277.5],
                    [277.5, 555, 277.5], [277.5, 277.5, 555], [212.5, 82.5, 147.5], [347.5, 165, 377.5], [278, 278, -750],
                    [153, 554, 56], [393, 554, 286], [0, 0, 0], [555, 555, 555], [213, 554, 227]], np.float64)
    for p in pts:
        o.append(np.repeat(p[None], len(grid), 0))
        d.append(grid)
    # aim exactly at rect corners / edges / box corners from several origins
    targets = np.array([[0, 0, 0], [555, 0, 0], [0, 555, 0], [555, 555, 555], [153, 554, 56], [393, 554, 56], [153, 554, 286],
                        [393, 554, 286], [273, 554, 56], [555, 277.5, 0], [277.5, 0, 555], [130, 165, 65], [295, 165, 230],
                        [265, 330, 295], [430, 330, 460], [212.5, 165, 147.5]], np.float64)
    for src in pts[:10]:
        o.append(np.repeat(src[None], len(targets), 0))
        d.append(targets - src)
        o.append(np.repeat(src[None], len(targets), 0))
        d.append((targets - src) * 1e-4)
    # pathological values
    bad = np.array([[np.nan, 1, 1], [1, np.nan, 0], [np.inf, 1, 1], [1, -np.inf, 1], [1e38, 1e38, 1e38], [1e-38, 1e-38, 1e-38],
                    [1e-45, 0, 1], [0, 0, 0], [-0.0, -0.0, -0.0], [3e38, -3e38, 1]])
    for p in pts[:4]:
        o.append(np.repeat(p[None], len(bad), 0))
        d.append(bad)
    o.append(np.array([[np.nan, 278, 278], [278, np.inf, 278], [1e30, 278, 278], [278, 278, -1e30]]))
    d.append(np.array([[0, 0, 1], [0, -1, 0], [-1, 0, 0], [0, 0, 1]], np.float64))
    return np.concatenate(o).astype(F), np.concatenate(d).astype(F)


def same_t(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)) | (a == b)


@pytest.mark.parametrize("scene", ALL_SCENES)
def test_world_hit_matches_oracle_on_adversarial_rays(oracle, scene):
    rng = np.random.default_rng(7)
    sc = pt.Scene(scene_path(scene), 64, 64)
    o, d = adversarial_rays(rng, sc.nodes()[0][0])
    r = pt.Renderer(sc, max_paths_in_flight=4096)
    osc = oracle.Scene.from_json(scene_path(scene))
    k0, k1, vd = 0x1234567, 0x89abcdef, 40
    # one ray per origin
    t, ids = r.trace_rays(o, d, k0, k1, vd)
    hit, ot, inst = osc.world_hit_stream(o, d, k0, k1, vd)
    ginst = np.where(ids >= 0, ids >> 3, -1)
    assert np.array_equal(ginst, inst), f"{(ginst != inst).sum()} instance mismatches, first at {np.argmax(ginst != inst)}"
    ok = same_t(t, ot) | (hit == 0)
    assert ok.all(), f"{(~ok).sum()} t mismatches, first at {np.argmin(ok)}: {t[np.argmin(ok)]} vs {ot[np.argmin(ok)]}"
    if scene.startswith("cornell_box"):
        assert np.isnan(t[ids >= 0]).sum() > 0, "the ray set must contain in-plane rays that the reference reports as NaN-t hits"
    assert (ids < 0).sum() > 100 and (ids >= 0).sum() > 5000
    # several rays sharing an origin (any grouping of consecutive directions is a valid test): 4 = the four-wide
    # instantiation, 2 = the two-ray one k_connect launches by default
    for nr in (4, 2):
        nn = (len(o) // nr) * nr
        dn = d[:nn].reshape(-1, nr, 3)
        on = o[:nn:nr]
        tn, idn = r.trace_rays(on, dn, k0, k1, vd)
        for k in range(nr):
            hit, ot, inst = osc.world_hit_stream(on, dn[:, k], k0, k1, vd + 16 * k)
            ginst = np.where(idn[:, k] >= 0, idn[:, k] >> 3, -1)
            assert np.array_equal(ginst, inst), (nr, k, (ginst != inst).sum())
            assert (same_t(tn[:, k], ot) | (hit == 0)).all(), (nr, k)
    r.close()


@pytest.mark.parametrize("seed", [3, 11, 27, 39])
def test_world_hit_on_random_scenes(oracle, seed):
    # the same ray set against generated scenes (tests/scene_gen.py): rotated and non-uniformly scaled instances, spheres,
    # fog, 20-45 BVH nodes; plus rays aimed exactly at instance bounding-box corners
    import json
    from scene_gen import random_scene

    js = random_scene(seed)
    sc = pt.Scene(text=json.dumps(js), width=32, height=32)
    rng = np.random.default_rng(seed)
    o, d = adversarial_rays(rng, sc.nodes()[0][0], n_random=20000)
    _, _, bbox = sc.instance_tables()
    corners = np.concatenate([bbox[:, [0, 1, 2]], bbox[:, [3, 4, 5]], bbox[:, [0, 4, 2]], bbox[:, [3, 1, 5]]]).astype(np.float64)
    src = np.array([[278.0, 278.0, -700.0], [100.0, 500.0, 100.0]])
    for p in src:
        o = np.concatenate([o, np.repeat(p[None], len(corners), 0).astype(np.float32)])
        d = np.concatenate([d, (corners - p).astype(np.float32)])
    r = pt.Renderer(sc, max_paths_in_flight=4096)
    osc = oracle.Scene(oracle.sp.load_scene_params(js))
    k0, k1, vd = 0xabcdef1, 0x2468ace, 24
    t, ids = r.trace_rays(o, d, k0, k1, vd)
    hit, ot, inst = osc.world_hit_stream(o, d, k0, k1, vd)
    ginst = np.where(ids >= 0, ids >> 3, -1)
    assert np.array_equal(ginst, inst), f"{(ginst != inst).sum()} instance mismatches, first at {np.argmax(ginst != inst)}"
    assert (same_t(t, ot) | (hit == 0)).all()
    assert (ids >= 0).sum() > 2000
    nn = (len(o) // 2) * 2
    t2, id2 = r.trace_rays(o[:nn:2], d[:nn].reshape(-1, 2, 3), k0, k1, vd)
    for k in range(2):
        hit, ot, inst = osc.world_hit_stream(o[:nn:2], d[:nn].reshape(-1, 2, 3)[:, k], k0, k1, vd + 16 * k)
        assert np.array_equal(np.where(id2[:, k] >= 0, id2[:, k] >> 3, -1), inst), k
        assert (same_t(t2[:, k], ot) | (hit == 0)).all(), k
    r.close()


def _tame(a):
    """zero or 2^-20 <= |x| <= 2^20 (origins) -- the device's precondition for the fast sweep (pt_kernels.hip world_hit)"""
    a = np.abs(a)
    return (a == 0) | ((a >= 2.0 ** -20) & (a <= 2.0 ** 20))


def tame_rays(rng, sc, n_random=40000):
    """Rays that every lane of every wave sends down the FAST sweep (world_hit_fast: unscaled exact division, min/max
    slabs, axis-shaped transforms, fold instead of the COMBINE tree): finite, non-zero direction components within
    [2^-20, 2^20].  Aimed at random points, at the corners and edge midpoints of every rect / box face in world space
    (and a few ulps around them: exact-edge decisions, ties between adjacent leaves), along box edges, from origins on
    surfaces, inside and outside the scene."""
    F = np.float32
    fwd, inv, bbox = sc.instance_tables()
    d_ = sc.desc
    targets = []
    for i in range(d_.n_instances):
        pr = d_.primitives[d_.instances[i].primitive]
        loc = []
        if pr.type == pt.PRIM_RECT:
            x0, z0, x1, z1, y = list(pr.rect)
            for (cx, cz) in ((x0, z0), (x0, z1), (x1, z0), (x1, z1), ((x0 + x1) / 2, z0), (x0, (z0 + z1) / 2), ((x0 + x1) / 2, (z0 + z1) / 2)):
                c = (cx, y, cz)
                loc.append({0: (c[0], c[2], c[1]), 1: c, 2: (c[1], c[0], c[2])}[pr.plane])
        elif pr.type == pt.PRIM_BOX:
            p0, p1 = list(pr.p0), list(pr.p1)
            for a in (0, 1, 2):
                for b in (0, 1, 2):
                    for c in (0, 1, 2):
                        loc.append(tuple(((p0[k], (p0[k] + p1[k]) / 2, p1[k])[s]) for k, s in enumerate((a, b, c))))
        if loc:
            L = np.array(loc, np.float64)
            M = fwd[i].astype(np.float64).reshape(3, 4)
            targets.append(L @ M[:, :3].T + M[:, 3])
    targets = np.concatenate(targets) if targets else np.zeros((0, 3))
    mn, mx = bbox[:, :3].min(0).astype(np.float64), bbox[:, 3:].max(0).astype(np.float64)
    ext = np.maximum(mx - mn, 1.0)
    o, d = [], []
    # random -> random
    src = mn - 0.25 * ext + rng.random((n_random, 3)) * 1.5 * ext
    dst = mn + rng.random((n_random, 3)) * ext
    o.append(src); d.append((dst - src) * rng.choice([1e-2, 1.0, 37.0], (n_random, 1)))
    # several origins -> every target, exactly and a few float32 ulps around it
    srcs = mn + rng.random((24, 3)) * ext
    for s in srcs:
        for k in (0, 1, -1, 3, -4):
            tg = np.nextafter(targets.astype(F), (np.inf if k > 0 else -np.inf) * np.ones_like(targets, F)) if k else targets.astype(F)
            for _ in range(max(abs(k) - 1, 0)):
                tg = np.nextafter(tg, (np.inf if k > 0 else -np.inf) * np.ones_like(tg))
            o.append(np.repeat(s[None], len(targets), 0)); d.append(tg.astype(np.float64) - s.astype(F).astype(np.float64))
    # origins ON the surfaces (the targets themselves) towards other targets: shadow-ray-like, t of the own surface ~ 0
    if len(targets) > 1:
        idx = rng.integers(0, len(targets), (20000, 2))
        o.append(targets[idx[:, 0]]); d.append(targets[idx[:, 1]] - targets[idx[:, 0]])
    o = np.concatenate(o).astype(F)
    d = np.concatenate(d).astype(F)
    keep = _tame(o).all(1) & (d != 0).all(1) & _tame(d).all(1)
    o, d = o[keep], d[keep]
    n = (len(o) // 256) * 256   # whole workgroups of tame rays only
    return o[:n], d[:n]


@pytest.mark.parametrize("scene", ALL_SCENES)
def test_fast_sweep_matches_oracle_and_general_sweep(oracle, scene, monkeypatch):
    """Every ray here takes world_hit_fast on the device (all lanes tame).  It must equal the oracle's World::hit bit
    for bit, and so must the general sweep on the same rays (PATHTRACE_HIP_TRAVERSAL=general switches the fast one off) and the
    per-lane walk that large scenes use (PATHTRACE_HIP_TRAVERSAL=walk selects it for these small ones)."""
    rng = np.random.default_rng(21)
    sc = pt.Scene(scene_path(scene), 64, 64)
    o, d = tame_rays(rng, sc)
    assert len(o) > 30000
    osc = oracle.Scene.from_json(scene_path(scene))
    k0, k1, vd = 0x7654321, 0x0fedcba9, 24
    hit, ot, inst = osc.world_hit_stream(o, d, k0, k1, vd)
    results = []
    for env in (None, "1", "walk"):   # fast sweep (flat or tree program) / general sweep / per-lane walk
        monkeypatch.delenv("PATHTRACE_HIP_TRAVERSAL", raising=False)
        if env == "1":
            monkeypatch.setenv("PATHTRACE_HIP_TRAVERSAL", "general")
        elif env == "walk":
            monkeypatch.setenv("PATHTRACE_HIP_TRAVERSAL", "walk")
        r = pt.Renderer(sc, max_paths_in_flight=4096)
        t, ids = r.trace_rays(o, d, k0, k1, vd)
        ginst = np.where(ids >= 0, ids >> 3, -1)
        assert np.array_equal(ginst, inst), (env, int((ginst != inst).sum()), int(np.argmax(ginst != inst)))
        ok = same_t(t, ot) | (hit == 0)
        assert ok.all(), (env, int((~ok).sum()), int(np.argmin(ok)))
        for nr in (2, 4):
            nn = (len(o) // (256 * nr)) * 256 * nr
            dn = d[:nn].reshape(-1, nr, 3)
            on = o[:nn:nr]
            tn, idn = r.trace_rays(on, dn, k0, k1, vd)
            results.append((env, nr, tn.copy(), idn.copy()))
            for k in range(nr):
                h2, ot2, inst2 = osc.world_hit_stream(on, dn[:, k], k0, k1, vd + 16 * k)
                g2 = np.where(idn[:, k] >= 0, idn[:, k] >> 3, -1)
                assert np.array_equal(g2, inst2), (env, nr, k)
                assert (same_t(tn[:, k], ot2) | (h2 == 0)).all(), (env, nr, k)
        results.append((env, 1, t.copy(), ids.copy()))
        r.close()
    monkeypatch.delenv("PATHTRACE_HIP_TRAVERSAL", raising=False)
    # fast == general == walk including the face of the hit (ids, not only instances)
    by = {(e, nr): (t_, i_) for e, nr, t_, i_ in results}
    for nr in (1, 2, 4):
        tf, idf = by[(None, nr)]
        for other in ("1", "walk"):
            tg, idg = by[(other, nr)]
            assert np.array_equal(idf, idg), (other, nr)
            assert (same_t(tf, tg) | (idf < 0)).all(), (other, nr)
    assert (ids >= 0).sum() > 5000
