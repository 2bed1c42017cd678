#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref/ref_driver, i.e. the
reference's own headers compiled in place with its own flags, g++ -O3, threads=1).

Runs only in the build container (needs /root/reference to build oracle/_ref).  The outputs
are data (inputs + expected outputs); no reference source is stored.

  F1  fb_<scene>_<cfg>.npy       linear float framebuffer SUM (row 0 = bottom), + ray count
  F2  rng_after_static_init.npy  first 4096 random_double() values after static init
  F3  tables_<scene>.txt         fwd/inv matrices, bboxes, BVH topology, lights, camera (hex floats)
  F4  samples_<scene>.npy        first 4096 camera samples: u v ray(7) col(3) rays
  F5  hits_<scene>.npy           world->hit of the first 4096 camera rays: hit t p n inst
  F6  ppm_<scene>_64x64x4.ppm    the reference's P6 film output (tonemap + sRGB) for framebuffer fb_<scene>_64x64x4
  F7  perlin_tables.npy          perlin::ranvec (768 floats) + perm_x/y/z (768 values) as the static initialisers left them
  F9  fb_random_<seed>.npy       framebuffers of 24 seeded random scenes (tests/scene_gen.py) at 40x30x3
  F8  texeval_<scene>.npz        2048 rows (u v px py pz) + every texture's value rgb / alpha there (texture scenes)
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import pt_oracle as po  # noqa: E402
from oracle import scene_params as sp  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = ["cornell_box", "cornell_box_small_lights", "cornell_box_with_volume"]
# scenes beyond the BASELINE configs: sphere lights + metal, dielectric, a room-filling volume
EXTRA_SCENES = ["light_test", "three_orbs", "cornell_box_with_volume2", "cornell_box_nested_fog"]   # the last: a medium whose boundary is a medium
# checker / perlin textures, textured emitter, textured World::background (scenes/ + tools/author_scenes.py)
TEXTURE_SCENES = ["cornell_box_image_light", "textured_room", "image_room"]
EXTRA_SCENES = EXTRA_SCENES + TEXTURE_SCENES

# name -> (width, height, samples, kwargs)
CONFIGS = {
    "200x200x16": (200, 200, 16, {}),                       # BASELINE config 1
    "64x64x4": (64, 64, 4, {}),                             # quick set
    "96x54x8_t32": (96, 54, 8, dict(block_w=32, block_h=32)),  # 16:9, 3x2 tiles, clamped edge tiles
    "48x48x8_ls1": (48, 48, 8, dict(light_samples=1)),
    "48x48x8_norr": (48, 48, 8, dict(russian_roulette=False)),
    "48x48x8_direct": (48, 48, 8, dict(only_direct=True)),
    "48x48x8_mb3": (48, 48, 8, dict(max_bounces=3, light_samples=2)),
}
FULL_ONLY = {"200x200x16"}  # rendered for every scene; the variants only for these:
VARIANT_SCENES = {"cornell_box", "cornell_box_with_volume"}


def main():
    po.build()
    assert po.ref_available(), "oracle/_ref/ref_driver missing (needs /root/reference)"
    os.makedirs(GOLD, exist_ok=True)
    manifest = {"generator": "tools/make_golden.py", "reference_build": "g++ -pthread --std=c++14 -O3, threads=1",
                "framebuffers": [], "samples": [], "hits": [], "tables": [], "texeval": []}
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rng.f64")
        po.ref_run(sp.load_scene_params(os.path.join(ROOT, "scenes", SCENES[0] + ".json")), "rng", ["4096", out], d)
        np.save(os.path.join(GOLD, "rng_after_static_init.npy"), np.fromfile(out, np.float64))
        out = os.path.join(d, "perlin.f32")
        po.ref_run(sp.load_scene_params(os.path.join(ROOT, "scenes", SCENES[0] + ".json")), "perlin", [out], d)
        np.save(os.path.join(GOLD, "perlin_tables.npy"), np.fromfile(out, np.float32))
        for scene in SCENES + EXTRA_SCENES:
            P = sp.load_scene_params(os.path.join(ROOT, "scenes", scene + ".json"))
            if scene in TEXTURE_SCENES:
                rng = np.random.default_rng(20260101)
                n = 2048
                pts = np.zeros((n, 5), np.float32)
                pts[:, :2] = rng.uniform(-1.5, 2.5, (n, 2))
                pts[: n // 2, 2:] = rng.uniform(-50, 600, (n // 2, 3))      # room-scale hit points
                pts[n // 2:, 2:] = rng.normal(size=(n // 2, 3))            # unit directions (background lookups)
                pts[n // 2:, 2:] /= np.linalg.norm(pts[n // 2:, 2:], axis=1, keepdims=True)
                pin, pout = os.path.join(d, "pts.f32"), os.path.join(d, "tex.f32")
                pts.tofile(pin)
                po.ref_run(P, "texeval", [pin, str(n), pout], d)
                vals = np.fromfile(pout, np.float32).reshape(len(P.textures), n, 4)
                np.savez(os.path.join(GOLD, f"texeval_{scene}.npz"), points=pts, values=vals)
                manifest["texeval"].append({"scene": scene, "file": f"texeval_{scene}.npz", "n": n, "textures": len(P.textures)})
            t = os.path.join(d, "tables.txt")
            po.ref_run(P, "tables", [t, "1920", "1080"], d)
            with open(t) as f, open(os.path.join(GOLD, f"tables_{scene}.txt"), "w") as g:
                g.write(f.read())
            manifest["tables"].append({"scene": scene, "file": f"tables_{scene}.txt", "camera_aspect": [1920, 1080]})
            for cname, (w, h, spp, kw) in CONFIGS.items():
                if scene in EXTRA_SCENES and cname not in ("64x64x4", "96x54x8_t32") and not (scene in TEXTURE_SCENES and cname == "200x200x16"):
                    continue
                if cname not in FULL_ONLY and cname != "64x64x4" and scene not in VARIANT_SCENES and scene not in EXTRA_SCENES:
                    continue
                cfg = po.make_config(w, h, spp, **kw)
                ppm = os.path.join(d, "out.ppm") if cname == "64x64x4" else None
                fb, rays = po.ref_render(P, cfg, d, ppm)
                if ppm:   # F6: the reference's own film output for this framebuffer (renderer.h:24-55)
                    with open(ppm, "rb") as f, open(os.path.join(GOLD, f"ppm_{scene}_{cname}.ppm"), "wb") as g:
                        g.write(f.read())
                fn = f"fb_{scene}_{cname}.npy"
                np.save(os.path.join(GOLD, fn), fb)
                manifest["framebuffers"].append({"scene": scene, "config": cname, "file": fn, "width": w, "height": h,
                                                 "samples": spp, "kwargs": kw, "rays": rays})
                print(scene, cname, rays, flush=True)
            cfg = po.make_config(200, 200, 16)
            n = 4096
            np.save(os.path.join(GOLD, f"samples_{scene}.npy"), po.ref_samples(P, cfg, n, d))
            np.save(os.path.join(GOLD, f"hits_{scene}.npy"), po.ref_samples(P, cfg, n, d, mode="hits"))
            manifest["samples"].append({"scene": scene, "file": f"samples_{scene}.npy", "n": n, "config": "200x200x16"})
            manifest["hits"].append({"scene": scene, "file": f"hits_{scene}.npy", "n": n, "config": "200x200x16"})
        # F9: seeded random scenes (tests/scene_gen.py): arbitrary rotations / scales, spheres, several lights, volumes,
        # textures, up to 40 instances -- the reference's framebuffer and ray count for each
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from scene_gen import random_scene
        manifest["random_scenes"] = []
        for seed in range(24):
            P = sp.load_scene_params(random_scene(seed))
            cfg = po.make_config(40, 30, 3, block_w=16, block_h=16)
            ppm = os.path.join(d, "rnd.ppm") if seed < 12 else None
            fb, rays = po.ref_render(P, cfg, d, ppm)
            if ppm:
                with open(ppm, "rb") as f, open(os.path.join(GOLD, f"ppm_random_{seed:02d}.ppm"), "wb") as g:
                    g.write(f.read())
            np.save(os.path.join(GOLD, f"fb_random_{seed:02d}.npy"), fb)
            manifest["random_scenes"].append({"seed": seed, "file": f"fb_random_{seed:02d}.npy", "width": 40, "height": 30,
                                              "samples": 3, "kwargs": {"block_w": 16, "block_h": 16}, "rays": rays,
                                              "instances": len(P.instances)})
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("done")


if __name__ == "__main__":
    main()
