"""The reference-side binding compiles and flattens the reference's own objects into the scene the product's front end
builds.

tools/integration/hip_wavefront.h (`HipWavefront : Renderer` + the World -> pt_scene_desc walk) is compiled against the
REAL reference headers under /root/reference and linked with libpathtrace_hip.so by `make -C oracle plugin` (output
oracle/_ref/plugin_driver: build container only, never shipped, not run on the GPU box).  For every scene the flat scene
it derives from the reference's World / bvh_node / instance / material / texture objects must equal, field by field and
bit by bit, the one pth_scene_from_file parses from the scene JSON.  Skipped where the reference tree is absent."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import pathtrace_amd as pt
from conftest import ALL_SCENES, ROOT, scene_path

REF = "/root/reference"
DRIVER = os.path.join(ROOT, "oracle", "_ref", "plugin_driver")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "renderer.h")), reason="reference tree not present")


def f32(x):
    return "%08x" % np.float32(x).view(np.uint32)


def canonical(desc):
    """The text tools/integration/plugin_driver.cpp `flatten` writes, from a pt_scene_desc seen through ctypes."""
    d = desc
    out = ["counts %d %d %d %d %d %d" % (d.n_materials, d.n_primitives, d.n_instances, d.n_nodes, d.n_lights, d.n_textures)]
    for i in range(d.n_textures):
        t = d.textures[i]
        out.append("texture %d %s %s %s %s %d %d %s %d %d %d" % (t.type, f32(t.color[0]), f32(t.color[1]), f32(t.color[2]), f32(t.alpha),
                                                               t.even, t.odd, f32(t.scale), t.width, t.height, t.texel_offset))
    h = 1469598103934665603
    for i in range(d.texel_bytes):
        h = ((h ^ d.texels[i]) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    out.append("texels %d %016x" % (d.texel_bytes, h))
    for i in range(d.n_materials):
        m = d.materials[i]
        out.append("material %d %s %s %s %s %s %d %s %s %d" % (m.type, f32(m.color[0]), f32(m.color[1]), f32(m.color[2]), f32(m.alpha),
                                                             f32(m.power), m.two_sided, f32(m.fuzz), f32(m.ior), m.texture))
    for i in range(d.n_primitives):
        p = d.primitives[i]
        out.append(" ".join(["prim %d %d" % (p.type, p.material)] + [f32(p.rect[k]) for k in range(5)] + ["%d %d" % (p.plane, p.flipped)] +
                            [f32(p.p0[k]) for k in range(3)] + [f32(p.p1[k]) for k in range(3)] + [f32(p.center[k]) for k in range(3)] +
                            ["%s %d %s %d" % (f32(p.radius), p.boundary, f32(p.density), p.phase_material)]))
    for i in range(d.n_instances):
        n = d.instances[i]
        out.append(" ".join(["instance %d" % n.primitive] + [f32(n.fwd[k]) for k in range(12)] + [f32(n.inv[k]) for k in range(12)] +
                            [f32(n.bbox[k]) for k in range(6)]))
    for i in range(d.n_nodes):
        n = d.nodes[i]
        out.append(" ".join(["node"] + [f32(n.bbox[k]) for k in range(6)] + ["%d %d" % (n.left, n.right)]))
    out.append(" ".join(["lights"] + [str(d.lights[i]) for i in range(d.n_lights)]))
    c = d.camera
    cam = [x for name in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w") for x in getattr(c, name)] + [c.lens_radius]
    out.append(" ".join(["camera"] + [f32(x) for x in cam]))
    out.append(" ".join(["background"] + [f32(d.background[k]) for k in range(3)]))
    out.append("background_texture %d" % d.background_texture)
    if d.n_textures:
        out.append(" ".join(["perlin_ranvec"] + [f32(d.perlin_ranvec[k]) for k in range(768)]))
        out.append(" ".join(["perlin_perm"] + [str(d.perlin_perm[k]) for k in range(768)]))
    return "\n".join(out) + "\n"


@pytest.fixture(scope="module")
def driver():
    pt.lib()   # the product library the plugin links must exist
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "plugin"], capture_output=True, text=True)
    assert p.returncode == 0 and os.path.exists(DRIVER), p.stdout + p.stderr
    return DRIVER


def flatten_with_plugin(driver, js, w, h, tmp_path, base_dir=None):
    from oracle import scene_params as sp
    params = sp.load_scene_params(js, base_dir=base_dir) if base_dir else sp.load_scene_params(js)
    ptxt, out = tmp_path / "scene.params", tmp_path / "flat.txt"
    ptxt.write_text(sp.to_text(params))
    p = subprocess.run([driver, str(ptxt), "flatten", str(w), str(h), str(out)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return out.read_text()


@pytest.mark.parametrize("scene", ALL_SCENES)
def test_plugin_flattens_the_reference_world_into_the_front_ends_scene(driver, scene, tmp_path):
    w, h = 200, 120
    js = json.load(open(scene_path(scene)))
    got = flatten_with_plugin(driver, js, w, h, tmp_path, base_dir=ROOT)
    sc = pt.Scene(scene_path(scene), w, h)   # keep the scene alive while its arrays are read
    want = canonical(sc.desc)
    if got != want:
        gl, wl = got.splitlines(), want.splitlines()
        diff = [(i, a, b) for i, (a, b) in enumerate(zip(gl, wl)) if a != b][:3]
        raise AssertionError(f"{scene}: {len(gl)} vs {len(wl)} lines; first differences: {diff}")


@pytest.mark.parametrize("seed", [2, 13])
def test_plugin_on_generated_scenes(driver, seed, tmp_path):
    from scene_gen import random_scene
    js = random_scene(seed)
    got = flatten_with_plugin(driver, js, 64, 48, tmp_path)
    sc = pt.Scene(text=json.dumps(js), width=64, height=48)
    want = canonical(sc.desc)
    assert got == want
