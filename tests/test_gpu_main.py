"""The plugin surface end to end on the GPU: pth_main = the reference's main() (main.cpp:108-168) with render_type
"hip_wavefront": config.json + scene JSON in, P6 PPM out, preview PPMs while rendering (renderer.h:605-620)."""
import json
import os
import shutil
import threading
import time

import numpy as np
import pytest

import pathtrace_amd as pt
from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu


def _workdir(tmp_path, scene, **cfg):
    wd = tmp_path / "run"
    (wd / "scenes").mkdir(parents=True)
    (wd / "output").mkdir()
    shutil.copy(scene_path(scene), wd / "scenes" / f"{scene}.json")
    config = {
        "film": {"width": 64, "height": 64, "exposure": 0.0, "gamma": 2.2},
        "ppm_output_path": "output/render.ppm", "traced_paths_output_path": "output/out.txt",
        "traced_paths_2d_output_path": "output/out_2d.txt", "scene": f"scenes/{scene}.json",
        "render_type": "hip_wavefront", "integrator_type": "iterative nee path tracing", "should_trace_paths": False,
        "block_width": 128, "block_height": 128, "normal_offset": 0.0001, "max_bounces": 10, "samples": 4,
        "light_samples": 4, "russian_roulette": True, "threads": 1,
    }
    config.update(cfg)
    (wd / "config.json").write_text(json.dumps(config))
    return wd, config


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_box_with_volume", "three_orbs"])
def test_main_writes_the_ppm_of_the_rendered_framebuffer(tmp_path, scene):
    wd, cfg = _workdir(tmp_path, scene)
    rc = pt.lib().pth_main(os.fsencode(str(wd)))
    assert rc == 0, pt.last_error()
    got = (wd / "output" / "render.ppm").read_bytes()
    w, h, spp = cfg["film"]["width"], cfg["film"]["height"], cfg["samples"]
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc)
    fb = r.render(spp)
    r.close()
    ref = tmp_path / "ref.ppm"
    pt.write_ppm(str(ref), fb, spp, cfg["film"]["gamma"])     # config.h:24-25: "gamma" lands in film.exposure
    assert got == ref.read_bytes()


def test_main_fails_loudly(tmp_path):
    wd, _ = _workdir(tmp_path, "cornell_box", integrator_type="recursive path tracing")
    assert pt.lib().pth_main(os.fsencode(str(wd))) != 0 and "iterative nee" in pt.last_error()
    wd2 = tmp_path / "empty"
    wd2.mkdir()
    assert pt.lib().pth_main(os.fsencode(str(wd2))) != 0


def test_main_rewrites_a_preview_while_rendering(tmp_path):
    # a render of a second or two: sync_progress must have replaced the PPM by complete preview images (same header,
    # same size) before the final one, and the final file is the fully rendered image
    w, h, spp = 1920, 1080, 2048
    wd, cfg = _workdir(tmp_path, "cornell_box", film={"width": w, "height": h, "exposure": 0.0, "gamma": 2.2}, samples=spp)
    ppm = wd / "output" / "render.ppm"
    size = len(b"P6\n1920 1080\n255\n") + w * h * 3
    out = {}
    th = threading.Thread(target=lambda: out.setdefault("rc", pt.lib().pth_main(os.fsencode(str(wd)))))
    th.start()
    previews = []
    while th.is_alive():
        if ppm.exists() and ppm.stat().st_size == size:
            data = ppm.read_bytes()
            if len(data) == size and (not previews or previews[-1] != data):
                previews.append(data)
        time.sleep(0.02)
    th.join()
    assert out["rc"] == 0, pt.last_error()
    final = ppm.read_bytes()
    assert len(final) == size
    assert any(p != final for p in previews), "no preview PPM was seen before the final image"
    sc = pt.Scene(scene_path("cornell_box"), w, h)
    r = pt.Renderer(sc)
    fb = r.render(spp)
    r.close()
    ref = tmp_path / "ref.ppm"
    pt.write_ppm(str(ref), fb, spp, 2.2)
    assert final == ref.read_bytes()


@pytest.mark.parametrize("scene", ["cornell_box", "cornell_box_with_volume", "textured_room"])
def test_reference_side_plugin_renders_through_the_renderer_protocol(scene, tmp_path):
    """tools/integration/hip_wavefront.h -- `HipWavefront : Renderer`, compiled against the REAL reference headers in the build
    container (make -C oracle plugin -> oracle/_ref/plugin_driver, a binary that travels with the snapshot; the reference's
    sources do not) -- driven the way main.cpp:156-167 drives a renderer: start_render, sync_progress until is_done,
    finalize.  The framebuffer the reference-side object ends up holding (vec3 **framebuffer, renderer.h:141) must be the
    library's own render of the same scene and config, bit for bit."""
    import json
    import subprocess
    from oracle import scene_params as sp
    driver = os.path.join(ROOT, "oracle", "_ref", "plugin_driver")
    if not os.path.exists(driver):
        pytest.skip("oracle/_ref/plugin_driver was not built (needs the reference tree at build time)")
    w, h, spp = 160, 90, 6
    params = sp.load_scene_params(json.load(open(scene_path(scene))), base_dir=ROOT)
    ptxt, out = tmp_path / "scene.params", tmp_path / "fb.f32"
    ptxt.write_text(sp.to_text(params))
    # <cfg> = W H spp max_bounces light_samples rr normal_offset only_direct block_w block_h
    p = subprocess.run([driver, str(ptxt), "render", str(w), str(h), str(spp), "10", "4", "1", "0.0001", "0", "64", "64", str(out)],
                       capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-1500:])
    got = np.fromfile(out, np.float32).reshape(h, w, 3)
    sc = pt.Scene(scene_path(scene), w, h)
    r = pt.Renderer(sc, seed=0)
    want = r.render(spp)
    r.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
