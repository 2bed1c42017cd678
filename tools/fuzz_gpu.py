#!/usr/bin/env python3
"""One-off large fuzz of the device path against the oracle (not part of the test suite: minutes on the GPU box).

    python tools/fuzz_gpu.py [first_seed] [n_seeds] > gpurun_out/fuzz.json

For every seed: a generated scene (tests/scene_gen.py; varying instance counts so that both the flat program and the tree
program of the fast sweep run, plus the general sweep with PATHTRACE_HIP_TRAVERSAL=general on every 5th seed; light_samples
4, 1, 2, 7, 3 by seed so that k_shade's staged and unstaged instantiations both run; the chunk sort forced on for every 3rd
seed and the staging forced off for every 7th, a medium inside a medium on every 9th, a walled room of rects on every 4th), rendered at 96x64x4 on the GPU and by the oracle in stream mode; framebuffer
bits and all nine path counters must agree.  With PATHTRACE_HIP_SPEC=sync in the environment every scene is rendered by its own
build of the traversal kernels (counted as "per_scene_build")."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import pathtrace_amd as pt
from oracle import pt_oracle as oracle
from scene_gen import random_scene, room_scene

CTR = {"rays": "rays", "extension_rays": "ext_rays", "extension_hits": "ext_hits", "shadow_rays": "shadow_rays",
       "term_miss": "term_miss", "term_rr": "term_rr", "term_emitter": "term_emitter", "term_pdf": "term_pdf",
       "term_bounce_limit": "term_bounce_limit"}


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    # FUZZ_SIZE=WxHxSPP (default 96x64x4) and FUZZ_ROOMS=1 (walled rooms only): the long run behind the shadow sweeps' wall proof,
    # whose fall-backs -- a grazing ray in a wall's own plane, a hit beyond the sample -- need many rays to occur at all
    fw, fh, fspp = (int(x) for x in os.environ.get("FUZZ_SIZE", "96x64x4").split("x"))
    rooms_only = os.environ.get("FUZZ_ROOMS", "") == "1"
    oracle.build()
    bad, rays, modes = [], 0, {"flat": 0, "tree": 0, "general": 0}
    for seed in range(first, first + n):
        n_inst = [None, 6, 12, 30, 60, 120][seed % 6]
        nested = seed % 9 == 0   # a medium whose boundary is a medium (round 5): the general sweep carries the whole scene
        js = random_scene(seed, nested=nested) if n_inst is None else random_scene(seed, n_inst=n_inst, volume=(seed % 4 != 0), nested=nested)
        room = rooms_only or seed % 4 == 3     # a closed room of rects (round 5): the walls the module's shadow sweep proves unreachable
        if room:
            js = room_scene(seed)
            modes["walled_room"] = modes.get("walled_room", 0) + 1
        general = seed % 5 == 0
        if general:
            os.environ["PATHTRACE_HIP_TRAVERSAL"] = "general"
        else:
            os.environ.pop("PATHTRACE_HIP_TRAVERSAL", None)
        ls = [4, 1, 2, 7, 3][seed % 5 if not general else (seed // 5) % 5]
        shade = [tok for tok, on in (("sort", seed % 3 == 0), ("nostage", seed % 7 == 0)) if on]
        if shade:
            os.environ["PATHTRACE_HIP_SHADE"] = ",".join(shade)
        else:
            os.environ.pop("PATHTRACE_HIP_SHADE", None)
        try:
            sc = pt.Scene(text=json.dumps(js), width=fw, height=fh)
            r = pt.Renderer(sc, seed=seed, light_samples=ls)
        except pt.PathtraceError as e:
            modes.setdefault("refused", 0)
            modes["refused"] += 1
            continue
        if r.spec_status() == 1:   # PATHTRACE_HIP_SPEC=sync: the scene's own build of k_extend / k_connect renders it
            modes["per_scene_build"] = modes.get("per_scene_build", 0) + 1
        g = r.render(fspp)
        gc = r.counters()
        r.close()
        osc = oracle.Scene(oracle.sp.load_scene_params(js))
        o, oc = osc.render_stream(oracle.make_config(fw, fh, fspp, light_samples=ls), seed=seed, threads=16)
        same = (g.view(np.uint32) == o.view(np.uint32)) | (g == o)
        ok = bool(same.all()) and all(gc[a] == oc[b] for a, b in CTR.items())
        rays += gc["rays"]
        modes["staged" if (ls <= 4 and seed % 7 != 0) else "unstaged"] = modes.get("staged" if (ls <= 4 and seed % 7 != 0) else "unstaged", 0) + 1
        modes["general" if (general or nested) else ("flat" if sc.desc.n_instances <= 24 else "tree")] += 1
        if nested:
            modes["nested_medium"] = modes.get("nested_medium", 0) + 1
        if not ok:
            bad.append({"seed": seed, "mismatched": int((~same).sum())})
        print(f"seed {seed} inst {sc.desc.n_instances} {'ok' if ok else 'MISMATCH'}", file=sys.stderr, flush=True)
    print(json.dumps({"first_seed": first, "seeds": n, "size": [fw, fh, fspp], "rays": rays, "sweeps": modes, "mismatching_scenes": bad}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
