"""Host front end of the product (C++ behind the C ABI) -- no GPU needed.

* the shared library loads and exports every symbol include/pathtrace_hip.h declares;
* the flattened scene (matrices, bboxes, BVH, lights, camera) is bit-identical to the reference fixtures
  (tests/golden/tables_*.txt, dumped by oracle/_ref) and to the oracle's own tables;
* config.json parsing follows config.h:19-27,98-131 (defaults, required keys, the gamma/exposure swap);
* NaiveSpiral tile order (queue.h:68-127).
"""
import json
import os
import re

import numpy as np
import pytest

import pathtrace_amd as pt
from conftest import ALL_SCENES, GOLD, ROOT, SCENES, scene_path
from test_oracle_golden import _parse_tables, bits


def test_library_loads_and_exports_every_declared_symbol():
    L = pt.lib()
    header = open(os.path.join(ROOT, "include", "pathtrace_hip.h")).read()
    declared = set(re.findall(r"\b(pth?_[a-z_]+)\s*\(", header))
    declared -= {"pt_ctx", "pth_scene"}
    assert declared == set(pt.EXPORTS), declared ^ set(pt.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.pt_abi_version() == 6


@pytest.mark.parametrize("scene", ALL_SCENES)
def test_flattened_scene_matches_reference_tables(oracle, scene):
    gold = _parse_tables(os.path.join(GOLD, f"tables_{scene}.txt"))
    sc = pt.Scene(scene_path(scene), 1920, 1080)
    fwd, inv, bbox = sc.instance_tables()
    assert np.array_equal(bits(fwd), bits(gold["fwd"]))
    assert np.array_equal(bits(inv), bits(gold["inv"]))
    assert np.array_equal(bits(bbox), bits(gold["bbox"]))
    assert sc.lights() == gold["lights"]
    assert np.array_equal(bits(sc.camera()[:21]), bits(gold["cam"][:21]))
    # BVH: same preorder nodes / children as the oracle (itself pinned to the reference dump)
    osc = oracle.Scene.from_json(scene_path(scene))
    on, pn = osc.nodes(), sc.nodes()
    assert len(on) == len(pn)
    for (b1, l1, r1), (b2, l2, r2) in zip(on, pn):
        assert (l1, r1) == (l2, r2)
        assert np.array_equal(bits(b1), bits(b2))
    # materials / primitives agree with the oracle-side parser
    d = sc.desc
    P = osc.params
    assert d.n_materials == len(P.materials) and d.n_primitives == len(P.prims)
    for i, m in enumerate(P.materials):
        dm = d.materials[i]
        assert dm.type == m.type and tuple(np.float32(x) for x in dm.color) == tuple(m.color)
        assert (np.float32(dm.power), bool(dm.two_sided)) == (m.power, m.two_sided)
        assert dm.texture == m.texture and np.float32(dm.alpha) == m.alpha
    # textures: the same table in the same order, the Perlin tables of the static initialisers
    ts = sc.textures()
    assert len(ts) == len(P.textures) and d.background_texture == P.background_texture
    for dt, t in zip(ts, P.textures):
        assert (dt.type, dt.even, dt.odd, dt.width, dt.height) == (t.type, t.even, t.odd, t.width, t.height)
        assert tuple(np.float32(x) for x in dt.color) == tuple(t.color) and np.float32(dt.alpha) == t.alpha and np.float32(dt.scale) == t.scale
    gold = np.load(os.path.join(GOLD, "perlin_tables.npy"))
    rv, pm = sc.perlin_tables()
    assert np.array_equal(rv.ravel().view(np.uint32), gold[:768].view(np.uint32)) and np.array_equal(pm.ravel(), gold[768:].astype(np.int32))
    for i, p in enumerate(P.prims):
        dp = d.primitives[i]
        assert (dp.type, dp.material) == (p.type, p.mat)
        if p.type == pt.PRIM_RECT:
            assert tuple(np.float32(x) for x in dp.rect) == tuple(p.rect) and (dp.plane, bool(dp.flipped)) == (p.plane, p.flipped)
        elif p.type == pt.PRIM_BOX:
            assert tuple(np.float32(x) for x in dp.p0) == tuple(p.p0) and tuple(np.float32(x) for x in dp.p1) == tuple(p.p1)
        elif p.type == pt.PRIM_SPHERE:
            assert tuple(np.float32(x) for x in dp.center) == tuple(p.center) and np.float32(dp.radius) == p.radius
        elif p.type == pt.PRIM_VOLUME:
            assert (dp.boundary, np.float32(dp.density), dp.phase_material) == (p.boundary, p.density, p.phase_mat)


CONFIG = {
    "film": {"width": 600, "height": 400, "exposure": 0.5, "gamma": 2.2},
    "ppm_output_path": "output/render.ppm", "traced_paths_output_path": "output/out.txt",
    "traced_paths_2d_output_path": "output/out_2d.txt", "scene": "scenes/cornell_box.json",
    "render_type": "tiled", "integrator_type": "iterative nee path tracing", "should_trace_paths": True,
    "avg_number_of_paths": 100, "block_width": 128, "block_height": 128, "normal_offset": 0.0001,
    "max_bounces": 10, "samples": 20, "light_samples": 4, "russian_roulette": True, "threads": 10,
}


def test_config_parse_and_quirks():
    hc = pt.load_config(text=json.dumps(CONFIG))
    assert (hc.width, hc.height, hc.samples, hc.max_bounces, hc.light_samples, hc.threads) == (600, 400, 20, 10, 4, 10)
    # config.h:24-25: "gamma" is stored in exposure and "exposure" in gamma
    assert hc.exposure == np.float32(2.2) and hc.gamma == np.float32(0.5)
    assert hc.render_type == 2 and hc.integrator_type == 4
    assert hc.scene_path == b"scenes/cornell_box.json" and hc.png_output_path == b"out.png"
    assert hc.trace_probability == np.float32(100.0 / (20 * 600 * 400))
    # defaults (config.h:98-131)
    minimal = {"film": {}, "traced_paths_output_path": "a", "traced_paths_2d_output_path": "b"}
    hc = pt.load_config(text=json.dumps(minimal))
    assert (hc.width, hc.height, hc.block_width, hc.block_height) == (400, 300, 64, 64)
    assert (hc.render_type, hc.integrator_type, hc.max_bounces, hc.samples, hc.light_samples, hc.threads) == (1, 0, 10, 20, 1, 1)
    assert hc.russian_roulette == 1 and hc.normal_offset == np.float32(0.0001) and hc.trace_probability == 0.0
    assert hc.scene_path == b"scenes/scene.json" and hc.ppm_output_path == b"out.ppm"
    # unknown strings fall back to enum value 0 (std::map::operator[]); our own render type parses
    hc = pt.load_config(text=json.dumps({**minimal, "render_type": "bogus", "integrator_type": "bogus"}))
    assert (hc.render_type, hc.integrator_type) == (0, 0)
    assert pt.load_config(text=json.dumps({**minimal, "render_type": "hip_wavefront"})).render_type == 3
    # required keys: .get<std::string>() on a missing key throws in the reference
    with pytest.raises(pt.PathtraceError):
        pt.load_config(text=json.dumps({"film": {}}))
    with pytest.raises(pt.PathtraceError):
        pt.load_config(text="{ not json")


def test_spiral_tile_order(oracle):
    # 200x200/128 -> (0,0),(1,0),(1,1),(0,1); every tile exactly once at 1080p / 4K
    assert pt.spiral_tiles(200, 200, 128, 128) == [(0, 0, 128, 128), (128, 0, 200, 128), (128, 128, 200, 200), (0, 128, 128, 200)]
    for (w, h, n) in [(1920, 1080, 135), (3840, 2160, 510), (96, 54, 6), (600, 600, 25)]:
        tiles = pt.spiral_tiles(w, h, 128 if w > 100 else 32, 128 if w > 100 else 32)
        assert len(tiles) == n and len(set(tiles)) == n
        cover = np.zeros((h, w), np.int32)
        for x0, y0, x1, y1 in tiles:
            cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all()


def test_scene_errors_are_loud():
    with pytest.raises(pt.PathtraceError):
        pt.Scene("/nonexistent/scene.json", 10, 10)
    with pytest.raises(pt.PathtraceError):
        pt.Scene(text='{"camera": {"look_from": [0,0,0], "look_at": [0,0,1]}, "instances": []}', width=10, height=10)
    bad = json.load(open(scene_path("cornell_box")))
    bad["instances"][0]["primitive"]["id"] = "nope"
    with pytest.raises(pt.PathtraceError):
        pt.Scene(text=json.dumps(bad), width=10, height=10)


@pytest.mark.parametrize("scene", ALL_SCENES)
def test_film_output_matches_reference_ppm(tmp_path, scene):
    # F6: the reference's own P6 file (calculate_luminance helpers.h:146-168 -> tonemap_uncharted tonemap.h:4-24 -> to_srgb
    # helpers.h:78-93 -> output_to_file renderer.h:24-55, with film.exposure holding the "gamma" value 2.2 per the
    # config.h:24-25 swap) for the reference's own framebuffer.  Byte-identical, header included.
    fb = np.load(os.path.join(GOLD, f"fb_{scene}_64x64x4.npy"))
    gold = open(os.path.join(GOLD, f"ppm_{scene}_64x64x4.ppm"), "rb").read()
    out = tmp_path / "o.ppm"
    pt.write_ppm(str(out), fb, samples=4, exposure_field=2.2)
    got = out.read_bytes()
    assert len(got) == len(gold)
    assert got == gold, f"{sum(a != b for a, b in zip(got, gold))} of {len(gold)} bytes differ"


def test_film_output_matches_reference_ppm_on_random_scenes(tmp_path, manifest):
    # the same for twelve generated scenes (bright sphere lights, fog, open sky: very different white points)
    n = 0
    for e in manifest["random_scenes"]:
        ppm = os.path.join(GOLD, f"ppm_random_{e['seed']:02d}.ppm")
        if not os.path.exists(ppm):
            continue
        fb = np.load(os.path.join(GOLD, e["file"]))
        out = tmp_path / "r.ppm"
        pt.write_ppm(str(out), fb, samples=e["samples"], exposure_field=2.2)
        assert out.read_bytes() == open(ppm, "rb").read(), e["seed"]
        n += 1
    assert n >= 12


def test_ppm_writer(tmp_path):
    # film output (renderer.h:24-55): header, size, bottom row written last, white point = max luminance
    fb = np.zeros((4, 6, 3), np.float32)
    fb[0, 0] = 8.0    # bottom-left pixel, the brightest
    fb[3, 5] = 0.5
    p = tmp_path / "o.ppm"
    pt.write_ppm(str(p), fb, samples=2)
    raw = p.read_bytes()
    assert raw.startswith(b"P6\n6 4\n255\n") and len(raw) == len(b"P6\n6 4\n255\n") + 6 * 4 * 3
    px = np.frombuffer(raw[len(b"P6\n6 4\n255\n"):], np.uint8).reshape(4, 6, 3)
    assert tuple(px[3, 0]) == (255, 255, 255)   # file row 3 = framebuffer row 0
    assert 0 < px[0, 5, 0] < 255 and px[1, 1].sum() == 0


def _one_texture_scene(path):
    return {"camera": {"look_from": [0, 0, -5], "look_at": [0, 0, 0]}, "world": {"color": [0, 0, 0]},
            "textures": [{"id": "t", "type": "png", "data": {"path": path}}],
            "materials": [{"id": "m", "type": "lambertian", "data": {"texture": "t"}}], "primitives": [],
            "instances": [{"type": "direct", "primitive": {"type": "rect", "material": {"id": "m"}, "size": [1, 1]}}]}


@pytest.mark.parametrize("name", ["poster", "sky", "lamp", "palette", "grey"])
def test_png_reader_matches_the_test_side_reader(name):
    # what lodepng::decode hands to from_4byte_vector (scene_parser.h:39-55): RGBA8, row 0 first.  The assets cover all
    # five colour types, all five row filters, stored / fixed / dynamic deflate blocks, tRNS and split IDAT chunks
    # (tools/author_scenes.py); the checker is an independent reader on Python's zlib (oracle/scene_params.py).
    from oracle import scene_params as sp

    path = os.path.join(ROOT, "assets", name + ".png")
    w, h, rgba = sp.decode_png(path)
    sc = pt.Scene(text=json.dumps(_one_texture_scene(path)), width=8, height=8)
    t = sc.textures()[0]
    assert (t.type, t.width, t.height, t.texel_offset) == (3, w, h, 0)
    assert sc.texel_bytes() == rgba and len(rgba) == 4 * w * h


def test_png_errors_are_loud(tmp_path):
    for body in (b"", b"not a png at all", open(os.path.join(ROOT, "assets", "sky.png"), "rb").read()[:90]):
        p = tmp_path / "bad.png"
        p.write_bytes(body)
        with pytest.raises(pt.PathtraceError):
            pt.Scene(text=json.dumps(_one_texture_scene(str(p))), width=8, height=8)
    # relative paths resolve against the parent of the scene file's directory (the reference: its working directory)
    sc = pt.Scene(scene_path("image_room"), 8, 8)
    assert [t.type for t in sc.textures()] == [3, 3, 0, 1]


def test_random_scenes_flatten_like_the_oracle(oracle):
    # the C++ front end against the oracle-side parser + constructors (themselves bit-identical to the reference on these
    # scenes, test_oracle_golden.py): matrices, inverses, bounding boxes, BVH order and boxes, lights, materials, textures
    from scene_gen import random_scene

    for seed in range(24, 64):
        js = random_scene(seed)
        osc = oracle.Scene(oracle.sp.load_scene_params(js))
        sc = pt.Scene(text=json.dumps(js), width=40, height=30)
        for a, b in zip(sc.instance_tables(), osc.instance_tables()):
            assert np.array_equal(bits(a), bits(b)), seed
        on, pn = osc.nodes(), sc.nodes()
        assert len(on) == len(pn), seed
        for (b1, l1, r1), (b2, l2, r2) in zip(on, pn):
            assert (l1, r1) == (l2, r2) and np.array_equal(bits(b1), bits(b2)), seed
        assert sc.lights() == osc.lights(), seed
        P, d = osc.params, sc.desc
        assert d.n_materials == len(P.materials) and d.n_primitives == len(P.prims) and d.n_textures == len(P.textures)
        for i, m in enumerate(P.materials):
            dm = d.materials[i]
            assert (dm.type, dm.texture) == (m.type, m.texture) and tuple(np.float32(x) for x in dm.color) == tuple(m.color), seed
        assert d.background_texture == P.background_texture and np.array_equal(bits(sc.camera()[:21]), bits(osc.camera(40, 30)[:21]))


def test_the_library_reads_exactly_the_documented_environment_variables():
    """at most twelve PATHTRACE_HIP_* variables, every one listed in the table of include/pathtrace_hip.h."""
    import glob
    read = set()
    for f in glob.glob(os.path.join(ROOT, "pathtrace_amd", "csrc", "*", "*")):
        if f.endswith((".cpp", ".hip", ".h")):
            read |= set(re.findall(r'(?:getenv|get|env_token)\("(PATHTRACE_HIP_[A-Z_]+)"', open(f).read()))
    header = open(os.path.join(ROOT, "include", "pathtrace_hip.h")).read()
    table = header[header.index("environment variables the library reads"):header.index("Everything else that used to be a run-time knob")]
    documented = set(re.findall(r"^ \*   (PATHTRACE_HIP_[A-Z_]+)", table, re.M))
    assert read == documented, read ^ documented
    assert len(read) <= 12
