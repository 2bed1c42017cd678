"""GPU parity tests proper (run with -m gpu on an MI355X).  Everything goes through the C ABI
(libpathtrace_hip.so); the oracle (CPU restatement, stream mode: same counter RNG, same portable math) is
only the checker.

Bar (floating point path, tolerance stated here): the per-pixel linear framebuffer SUM must be BIT-IDENTICAL
to the oracle in stream mode -- tolerance 0 ulp -- together with every path counter.  This is stricter than
the fallback tolerance of SURVEY.md 8(d) (1e-3 relative on 99.5 % of pixel-channels); if a future change makes
the arithmetic diverge, that fallback is what `assert_close_enough` checks and the test says so loudly.
"""
import numpy as np
import pytest

import pathtrace_amd as pt
from conftest import ALL_SCENES, EXTRA_SCENES, SCENES, TEXTURE_SCENES, scene_path

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def oracle_cfg(oracle, w, h, spp, **kw):
    return oracle.make_config(w, h, spp, **kw)


def gpu_render(scene, w, h, spp, seed=0, max_paths=0, rect=None, **kw):
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc, seed=seed, max_paths_in_flight=max_paths,
                    max_bounces=kw.get("max_bounces", 10), light_samples=kw.get("light_samples", 4),
                    russian_roulette=kw.get("russian_roulette", True), only_direct=kw.get("only_direct", False))
    fb = r.render(spp, rect)
    ctr = r.counters()
    r.close()
    return fb, ctr


def assert_bit_identical(gpu, ref, what):
    same = bits(gpu) == bits(ref)
    # +0.0 / -0.0 are the same radiance
    same |= (gpu == ref)
    if not same.all():
        bad = np.argwhere(~same)
        j, i, c = bad[0]
        rel = np.abs(gpu - ref) / np.maximum(np.abs(ref), 1e-30)
        raise AssertionError(f"{what}: {len(bad)} of {same.size} pixel-channels differ; first at (i={i}, j={j}, c={c}): "
                             f"gpu={gpu[j, i, c]!r} oracle={ref[j, i, c]!r}; max rel err {rel.max():.3g}")


CTR_MAP = {"rays": "rays", "extension_rays": "ext_rays", "extension_hits": "ext_hits", "shadow_rays": "shadow_rays",
           "term_miss": "term_miss", "term_rr": "term_rr", "term_emitter": "term_emitter", "term_pdf": "term_pdf",
           "term_bounce_limit": "term_bounce_limit"}


def assert_counters(gc, oc, what):
    for g, o in CTR_MAP.items():
        assert gc[g] == oc[o], f"{what}: counter {g}: gpu {gc[g]} oracle {oc[o]}"


@pytest.mark.parametrize("scene", ALL_SCENES)
def test_bit_exact_vs_oracle_stream_64x64x4(oracle, scene):
    w, h, spp = 64, 64, 4
    ref, oc = oracle.Scene.from_json(scene_path(scene)).render_stream(oracle_cfg(oracle, w, h, spp), seed=0)
    gpu, gc = gpu_render(scene, w, h, spp)
    assert_bit_identical(gpu, ref, scene)
    assert_counters(gc, oc, scene)
    assert gc["camera_samples"] == w * h * spp


@pytest.mark.parametrize("scene", SCENES)
def test_bit_exact_config1_size_200x200x16(oracle, scene):
    # BASELINE config 1 size on all three scenes (the oracle needs ~1 s with 8+ threads)
    w, h, spp = 200, 200, 16
    ref, oc = oracle.Scene.from_json(scene_path(scene)).render_stream(oracle_cfg(oracle, w, h, spp), seed=0)
    gpu, gc = gpu_render(scene, w, h, spp)
    assert_bit_identical(gpu, ref, scene)
    assert_counters(gc, oc, scene)


@pytest.mark.parametrize("scene", EXTRA_SCENES + TEXTURE_SCENES)
def test_bit_exact_extra_scenes_16x9(oracle, scene):
    # beyond BASELINE: sphere lights with cone sampling + metal, dielectric spheres, a room-filling volume;
    # (8f-4): checker / perlin / image textures, textured emitter, textured and image World::background
    w, h, spp = 160, 90, 8
    ref, oc = oracle.Scene.from_json(scene_path(scene)).render_stream(oracle_cfg(oracle, w, h, spp), seed=5)
    gpu, gc = gpu_render(scene, w, h, spp, seed=5)
    assert_bit_identical(gpu, ref, scene)
    assert_counters(gc, oc, scene)


@pytest.mark.parametrize("kw", [dict(light_samples=1), dict(russian_roulette=False), dict(only_direct=True),
                                dict(max_bounces=3, light_samples=2), dict(max_bounces=1), dict(light_samples=7)])
@pytest.mark.parametrize("scene", ["cornell_box", "cornell_box_with_volume"])
def test_bit_exact_config_variants(oracle, scene, kw):
    w, h, spp = 96, 54, 6   # 16:9, ~43 % of the camera rays miss the box
    ref, oc = oracle.Scene.from_json(scene_path(scene)).render_stream(oracle_cfg(oracle, w, h, spp, **kw), seed=3)
    gpu, gc = gpu_render(scene, w, h, spp, seed=3, **kw)
    assert_bit_identical(gpu, ref, f"{scene} {kw}")
    assert_counters(gc, oc, f"{scene} {kw}")


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_box_small_lights", "cornell_box_with_volume", "light_test"])
def test_hits_without_a_shadow_record_change_nothing(oracle, scene, monkeypatch):
    # k_shade stages a hit's light samples in LDS and writes no shadow record when none of them can contribute (every
    # coefficient +-0 or NaN: surfaces facing away from the light, hits on the light); their shadow rays are counted, not
    # traced.  With PATHTRACE_HIP_SHADE=nostage every hit gets its record: same bits, same counters, both equal to the oracle.
    w, h, spp = 128, 72, 8
    ref, oc = oracle.Scene.from_json(scene_path(scene)).render_stream(oracle_cfg(oracle, w, h, spp), seed=9)
    staged, c0 = gpu_render(scene, w, h, spp, seed=9)
    monkeypatch.setenv("PATHTRACE_HIP_SHADE", "nostage")
    plain, c1 = gpu_render(scene, w, h, spp, seed=9)
    traced = ("rays_traced", "shadow_rays_traced")
    assert np.array_equal(bits(staged), bits(plain)) and {k: v for k, v in c0.items() if k not in traced} == {k: v for k, v in c1.items() if k not in traced}
    # the reference's count is the same either way; what was traced differs: every ray without staging, fewer with it
    assert c1["rays_traced"] == c1["rays"] and c1["shadow_rays_traced"] == c1["shadow_rays"]
    assert c0["rays"] - c0["rays_traced"] == c0["shadow_rays"] - c0["shadow_rays_traced"] > 0 and c0["shadow_rays_traced"] % 4 == 0
    assert_bit_identical(staged, ref, scene)
    assert_counters(c0, oc, scene)


@pytest.mark.parametrize("scene", ["cornell_box", "three_orbs"])
def test_chunk_sort_by_shading_class_changes_nothing(scene, monkeypatch):
    # k_shade's counting sort of a chunk by shading class is on by default only for textured scenes and scenes of more
    # than two lights; forced on and off, the image and the counters are the same (per-path arithmetic is lane-free)
    w, h, spp = 128, 72, 8
    monkeypatch.setenv("PATHTRACE_HIP_SHADE", "sort")
    a, ca = gpu_render(scene, w, h, spp, seed=4)
    monkeypatch.setenv("PATHTRACE_HIP_SHADE", "nosort")
    b, cb = gpu_render(scene, w, h, spp, seed=4)
    assert np.array_equal(bits(a), bits(b)) and ca == cb


def test_batching_tiling_and_sample_ranges_do_not_change_the_image(oracle):
    # size-independent property: the image is a pure function of (pixel, sample, seed); how the work is cut into
    # batches (max_paths_in_flight), tiles (NaiveSpiral) or sample ranges must not change a single bit.
    scene, w, h, spp = "cornell_box_small_lights", 160, 90, 8
    whole, c0 = gpu_render(scene, w, h, spp)
    small, c1 = gpu_render(scene, w, h, spp, max_paths=5000)   # many ragged batches, rows split into bands
    assert np.array_equal(bits(whole), bits(small)) and c0 == c1
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc, max_paths_in_flight=20000)
    for rect in pt.spiral_tiles(w, h, 64, 64):   # 3x2 tiles, clamped at the right/top edge
        r.render_async(0, 3, rect)
        r.render_async(3, spp, rect)
    tiled = r.framebuffer()
    assert np.array_equal(bits(whole), bits(tiled)) and r.counters() == c0
    # all tiles of two "ranks" as two multi-rect batches (tile k -> rank k mod 2), bands forced by a small slot budget
    r.clear()
    tiles = pt.spiral_tiles(w, h, 64, 64)
    for rank in range(2):
        r.render_tiles_async(tiles[rank::2], 0, spp)
    assert np.array_equal(bits(whole), bits(r.framebuffer())) and r.counters() == c0
    r2 = pt.Renderer(sc, max_paths_in_flight=3000)
    r2.render_tiles_async(tiles[1::2] + tiles[0::2], 0, spp)
    assert np.array_equal(bits(whole), bits(r2.framebuffer())) and r2.counters() == c0
    r2.close()
    # a single pixel / single sample batch (ragged minimum)
    r.clear()
    r.render_async(5, 6, (17, 23, 18, 24))
    one = r.framebuffer()
    ref, _ = oracle.Scene.from_json(scene_path(scene)).sample_stream(oracle_cfg(oracle, w, h, spp), 17, 23, 5)
    assert np.array_equal(bits(one[23, 17]), bits(ref))
    assert one.sum() == one[23, 17].sum()
    r.close()


@pytest.mark.parametrize("max_paths", [8192, 8192 + 77, 3 * 4096 - 1])
def test_multi_rect_batches_at_the_allocation_edge(max_paths, monkeypatch):
    # the index range behind round 2's abort.  Segments of 256 slots and a slot budget that the batches
    # fill to the last record (8192 = both tiles x 1 sample = 32 full segments), one past a segment boundary, and one
    # short of it: the last (merged) segment of every bounce ends at, or hangs over, the end of the streams.  Same bits
    # as the whole frame in one batch with the default segments.
    scene, w, h, spp = "cornell_box", 128, 64, 5
    whole, c0 = gpu_render(scene, w, h, spp)
    monkeypatch.setenv("PATHTRACE_HIP_PLAN", "seg=256")
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc, max_paths_in_flight=max_paths)
    tiles = pt.spiral_tiles(w, h, 64, 64)
    r.render_tiles_async(tiles[::-1], 0, spp)
    assert np.array_equal(bits(whole), bits(r.framebuffer())) and r.counters() == c0
    r.clear()
    for t in tiles:
        r.render_tiles_async([t], 0, 2)
        r.render_tiles_async([t], 2, spp)
    assert np.array_equal(bits(whole), bits(r.framebuffer())) and r.counters() == c0
    r.close()


def test_cost_balanced_tile_ownership(oracle):
    # the N > 1 scheduler on one GPU: per-tile costs measured through the C ABI equal the oracle's ray counts, the
    # ownership map derived from them is a partition, and the "ranks" rendered back to back give the one-GPU image
    from pathtrace_amd.distributed import measure_tile_costs, tiles_for_rank

    scene, w, h, spp, tile, world = "cornell_box", 320, 180, 4, 64, 3
    whole, c0 = gpu_render(scene, w, h, spp)
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc)
    tiles = pt.spiral_tiles(w, h, tile, tile)
    costs = measure_tile_costs(r, tiles)
    assert r.counters()["rays"] == 0 and not r.framebuffer().any()      # measuring leaves no trace
    # a tile's cost = the reference's ray count for it (one sample per pixel) + one unit per pixel: the tile rendered alone
    # says the same, and the extension rays among them are the oracle's
    osc = oracle.Scene.from_json(scene_path(scene))
    cfg = oracle_cfg(oracle, w, h, 1)
    scratch = np.zeros((h, w, 3), np.float32)
    for k in (0, 1, len(tiles) // 2, len(tiles) - 1):
        r.clear()
        r.render_tiles_async([tiles[k]], 0, 1)
        ck = r.counters()
        _, oc = osc.render_stream(cfg, seed=0, rect=tiles[k], threads=2, fb=scratch)
        assert ck["rays"] == oc["rays"] and ck["extension_rays"] == oc["ext_rays"]
        assert costs[k] == ck["rays"] + (tiles[k][2] - tiles[k][0]) * (tiles[k][3] - tiles[k][1])
    r.clear()
    # the same counts when the slot budget cuts the tiles into bands and the pass into several batch groups, and for 2 spp
    r_small = pt.Renderer(sc, max_paths_in_flight=5000)
    assert measure_tile_costs(r_small, tiles) == costs
    two = r_small.measure_tile_costs(tiles[:5], 2)
    r_small.close()
    r.render_tiles_async([tiles[1]], 0, 2)
    assert two[1] == r.counters()["rays"]
    r.clear()
    per_rank = [tiles_for_rank(w, h, tile, tile, k, world, costs) for k in range(world)]
    assert sorted(t for tr in per_rank for t in tr) == sorted(tiles)
    loads = [sum(costs[tiles.index(t)] for t in tr) for tr in per_rank]
    rr = [sum(costs[k::world]) for k in range(world)]
    assert max(loads) <= max(rr)
    for tr in per_rank:
        r.render_tiles_async(tr, 0, spp)
    assert np.array_equal(bits(whole), bits(r.framebuffer())) and r.counters() == c0
    r.close()


@pytest.mark.parametrize("spec", ["off", "sync"])
@pytest.mark.parametrize("scene,max_bounces", [("cornell_box", 80), ("three_orbs", 100), ("cornell_box_with_volume", 70)])
def test_more_bounces_than_live_count_words(oracle, scene, max_bounces, spec, monkeypatch):
    # The live-count bound of the queues (pt_kernels.hip chunk_limit) has words for 64 bounces; later ones trust the
    # segment counters again.  With russian roulette every batch is dead long before: the counters of every bounce must
    # then read zero (k_extend zeroes them whatever the bound says), else bounce 64 walks the stale records of the bounce
    # where the paths died a second time.  The reference accepts any max_bounces (config.h:116).
    monkeypatch.setenv("PATHTRACE_HIP_SPEC", spec)
    w, h, spp = 64, 36, 3
    kw = dict(max_bounces=max_bounces)
    ref, oc = oracle.Scene.from_json(scene_path(scene)).render_stream(oracle_cfg(oracle, w, h, spp, **kw), seed=2)
    gpu, gc = gpu_render(scene, w, h, spp, seed=2, **kw)
    assert_bit_identical(gpu, ref, f"{scene} max_bounces={max_bounces}")
    assert_counters(gc, oc, f"{scene} max_bounces={max_bounces}")
    # small slot budgets: many batches, each dead at its own bounce
    gpu2, gc2 = gpu_render(scene, w, h, spp, seed=2, max_paths=700, **kw)
    assert np.array_equal(bits(gpu), bits(gpu2)) and gc == gc2


@pytest.mark.parametrize("spec", ["off", "sync"])
def test_production_queue_geometry_in_one_batch(oracle, spec, monkeypatch):
    # One batch that takes every production-default turn of the queue walk at once: 512 x 512 = 2^18 pixels per sample (a
    # multiple of 2^16: segments of 4096 + 256 slots, pt_context.cpp render_group), 12 samples in flight = 723 segments (more
    # than the 512 below which segments stop merging: bounce 0 merges to 362, the later bounces keep them), the permuted segment
    # walk and the live-count bound -- on the library's kernels and on the scene's per-scene module.
    monkeypatch.setenv("PATHTRACE_HIP_SPEC", spec)
    w, h, spp = 512, 512, 12
    scene = "cornell_box"
    ref, oc = oracle.Scene.from_json(scene_path(scene)).render_stream(oracle_cfg(oracle, w, h, spp), seed=5)
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc, seed=5, max_paths_in_flight=w * h * spp)
    if spec == "sync":
        assert r.spec_wait() == 1, r.spec_info()
    gpu = r.render(spp)
    gc = r.counters()
    r.close()
    assert_bit_identical(gpu, ref, f"{scene} 512x512x12 in one batch, PATHTRACE_HIP_SPEC={spec}")
    assert_counters(gc, oc, scene)
    assert gc["camera_samples"] == w * h * spp


@pytest.mark.parametrize("scene,max_bounces", [("three_orbs", 50), ("light_test", 12), ("cornell_box", 50)])
def test_tile_costs_equal_the_ray_count_when_batches_die_early(scene, max_bounces):
    # pt_measure_tile_costs' contract: a tile's cost = the reference's ray count for it + one per pixel.  Open scenes and
    # small batches die many bounces before max_bounces: the tally of the remaining bounces must add nothing.
    from pathtrace_amd.distributed import measure_tile_costs

    w, h, tile = 160, 90, 32
    sc = pt.Scene(scene_path(scene), w, h)
    tiles = pt.spiral_tiles(w, h, tile, tile)
    for max_paths in (0, 3000):
        r = pt.Renderer(sc, max_bounces=max_bounces, max_paths_in_flight=max_paths)
        costs = measure_tile_costs(r, tiles)
        for k in (0, len(tiles) // 3, len(tiles) - 1):
            r.clear()
            r.render_tiles_async([tiles[k]], 0, 1)
            assert costs[k] == r.counters()["rays"] + (tiles[k][2] - tiles[k][0]) * (tiles[k][3] - tiles[k][1]), (scene, max_paths, k)
        r.clear()
        r.render_tiles_async(tiles, 0, 1)
        assert sum(costs) == r.counters()["rays"] + w * h
        r.close()


def test_progressive_snapshot_of_the_live_framebuffer(tmp_path):
    # SURVEY 8f-3 (renderer.h:605-620): the preview reads the framebuffer while the render is still running.  A snapshot
    # must not wait for the queued work, its sample count is monotone, every pixel is a prefix sum of that pixel's
    # samples (radiance >= 0, so prefix sums are bounded by the final sum) holding AT LEAST the counted batches, and once
    # idle it is the framebuffer.
    scene, w, h, spp = "cornell_box", 480, 270, 64
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc, max_paths_in_flight=w * h * 2)          # 32 batches of 2 spp
    prefix = {}
    for k in (2, 4, 8, 16, 32, 64):
        r.clear()
        r.render_async(0, k)
        prefix[k] = r.framebuffer()
    final = prefix[spp]
    r.clear()
    r.render_async(0, spp)
    seen, busy_snaps = [], 0
    while True:
        done, _, _ = r.poll()
        fb, n = r.snapshot()
        seen.append(n)
        assert n % (w * h * 2) == 0 and n <= w * h * spp
        assert (fb <= final).all() and (fb >= 0).all()
        k = n // (w * h)
        if k in prefix:
            assert (fb >= prefix[k]).all()
        if not done:
            busy_snaps += 1
        if done:
            break
    assert seen == sorted(seen)
    fb, n = r.snapshot()
    assert n == w * h * spp and np.array_equal(bits(fb), bits(final)) and np.array_equal(bits(fb), bits(r.framebuffer()))
    assert busy_snaps >= 1, "the render finished before the first snapshot: enlarge the job"
    # the preview file the plugin writes from such a snapshot: divisor 1 + done / (W*H)
    out = str(tmp_path / "preview.ppm")
    pt.write_ppm(out, prefix[8], 1 + 8)
    assert open(out, "rb").read().startswith(b"P6\n480 270\n255\n")
    r.clear()
    assert r.snapshot()[1] == 0 and not r.snapshot()[0].any()
    r.close()


def test_seed_changes_the_stream_but_not_the_estimate(oracle):
    scene, w, h, spp = "cornell_box", 64, 64, 64
    a, _ = gpu_render(scene, w, h, spp, seed=1)
    b, _ = gpu_render(scene, w, h, spp, seed=2)
    assert not np.array_equal(a, b)
    # same estimator: image means agree within Monte-Carlo error (cornell_box radiance is well behaved)
    assert np.allclose(a.mean(axis=(0, 1)), b.mean(axis=(0, 1)), rtol=0.03)


@pytest.mark.parametrize("scene", SCENES + TEXTURE_SCENES)
def test_statistical_agreement_with_reference_fixture(scene):
    # L3 of the parity ladder: same estimator, different random streams.  Like for like: the GPU renders
    # BASELINE config 1 (200x200x16) with 8 seeds; the REAL reference's fixture (one draw from the reference's
    # own stream) must look like a ninth seed.  The estimator is heavy tailed, so per-pixel
    # 16-spp means are clamped at 1.0 on both sides before comparing (clamp stated here; tolerances below).
    import os
    from conftest import GOLD
    w = h = 200
    spp, nseed, clamp = 16, 8, 1.0
    gold = np.minimum(np.load(os.path.join(GOLD, f"fb_{scene}_200x200x16.npy")) / spp, clamp)
    sc = pt.Scene(scene_path(scene), w, h)
    imgs = []
    for seed in range(nseed):
        r = pt.Renderer(sc, seed=seed)
        imgs.append(np.minimum(r.render(spp) / spp, clamp))
        r.close()
    imgs = np.array(imgs)
    # (1) per-channel image mean: |z| < 4 against the seed-to-seed spread
    m = imgs.mean(axis=(1, 2))
    z = (gold.mean(axis=(0, 1)) - m.mean(0)) / (m.std(0, ddof=1) * np.sqrt(1 + 1 / nseed))
    assert (np.abs(z) < 4).all(), z
    # (2) 8x8 block means (625 blocks x 3 channels): >= 95 % within 4 sigma, >= 97.5 % within 6 sigma
    bl = imgs.reshape(nseed, 25, 8, 25, 8, 3).mean(axis=(2, 4))
    rb = gold.reshape(25, 8, 25, 8, 3).mean(axis=(1, 3))
    zb = np.abs(rb - bl.mean(0)) / (bl.std(0, ddof=1) * np.sqrt(1 + 1 / nseed) + 1e-4)
    assert (zb < 4).mean() >= 0.95 and (zb < 6).mean() >= 0.975, ((zb < 4).mean(), (zb < 6).mean())
    if scene == "cornell_box":
        # the directly visible light is 2*Le = 1.2 in the reference (emission added twice), here clamped to 1
        raw = np.load(os.path.join(GOLD, f"fb_{scene}_200x200x16.npy")) / spp
        lit_r = np.isclose(raw, 1.2, atol=1e-3).all(axis=2)
        r = pt.Renderer(sc, seed=0)
        lit_g = np.isclose(r.render(spp) / spp, 1.2, atol=1e-3).all(axis=2)
        r.close()
        assert lit_r.sum() > 1000 and (lit_g & lit_r).sum() >= 0.98 * lit_r.sum()


def test_async_protocol_and_errors():
    sc = pt.Scene(scene_path("cornell_box"), 256, 256)
    r = pt.Renderer(sc)
    r.render_async(0, 32)
    done, samples, rays = r.poll()          # non-blocking, may or may not be finished
    r.wait()
    done, samples, rays = r.poll()
    assert done and samples == 256 * 256 * 32 and rays == r.counters()["rays"]
    with pytest.raises(pt.PathtraceError):
        r.render_async(0, 1, (0, 0, 300, 10))   # outside the film
    with pytest.raises(pt.PathtraceError):
        r.render_async(4, 4)                    # empty sample range
    r.clear()
    assert r.framebuffer().sum() == 0 and r.counters()["rays"] == 0
    r.close()
    # unsupported scene features are refused loudly at load, not silently approximated
    import json
    s = json.load(open(scene_path("cornell_box")))
    s["textures"] = [{"id": "n", "type": "png", "data": {"path": "assets/does_not_exist.png"}}]   # the reference's own case
    with pytest.raises(pt.PathtraceError):
        pt.Scene(text=json.dumps(s), width=64, height=64)
    s["textures"] = [{"id": "c", "type": "checker", "data": {"scale": 1.0, "odd": {"texture": "nope"}, "even": {"color": [1, 1, 1]}}}]
    with pytest.raises(pt.PathtraceError):
        pt.Scene(text=json.dumps(s), width=64, height=64)
    # image textures where the reference itself reads indeterminate or out-of-bounds data: on an emitter (NaN u, v from the
    # in-plane NEE ray of a path that lands on the light) and on a sphere (sphere::hit never sets u, v)
    import os
    from conftest import ROOT
    s = json.load(open(scene_path("image_room")))
    for t in s["textures"]:
        if t["type"] == "png":
            t["data"]["path"] = os.path.join(ROOT, t["data"]["path"])
    s["materials"][3] = {"id": "lamp", "type": "diffuse_light", "data": {"texture": "poster", "power": 7}}
    with pytest.raises(pt.PathtraceError, match="image-textured diffuse_light"):
        pt.Renderer(pt.Scene(text=json.dumps(s), width=64, height=64))
    s = json.load(open(scene_path("image_room")))
    for t in s["textures"]:
        if t["type"] == "png":
            t["data"]["path"] = os.path.join(ROOT, t["data"]["path"])
    s["instances"].append({"type": "direct", "primitive": {"type": "sphere", "material": {"id": "poster"}, "radius": 50},
                           "transform": {"translate": [200, 50, 200]}})
    with pytest.raises(pt.PathtraceError, match="u, v of a rect"):
        pt.Renderer(pt.Scene(text=json.dumps(s), width=64, height=64))
    s = json.load(open(scene_path("cornell_box_with_volume")))
    # a medium whose boundary is another medium renders since round 5 (scenes/cornell_box_nested_fog.json, the parity tests);
    # THREE deep is refused loudly
    s["primitives"].append({"id": "fog2", "type": "volume", "primitive": "fog", "density": 0.01, "color": [0.5, 0.5, 0.5]})
    s["instances"].append({"type": "ref", "primitive": {"id": "fog2"}})
    pt.Renderer(pt.Scene(text=json.dumps(s), width=64, height=64)).close()
    s["primitives"].append({"id": "fog3", "type": "volume", "primitive": "fog2", "density": 0.01, "color": [0.5, 0.5, 0.5]})
    s["instances"].append({"type": "ref", "primitive": {"id": "fog3"}})
    with pytest.raises(pt.PathtraceError, match="three deep"):
        pt.Renderer(pt.Scene(text=json.dumps(s), width=64, height=64))
