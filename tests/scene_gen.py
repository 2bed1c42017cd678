"""Seeded random scenes in the reference's scene-JSON schema (scene_parser.h:104-595), for parity sweeps beyond the
hand-authored scenes: arbitrary rotations and (non-uniform) scales, rect / box / sphere primitives in all alignments, refs
and direct instances, several lights (rect and sphere), metal / dielectric, an optional constant_medium with a box
boundary, optional checker / perlin textures on surfaces, emitter and background; `nested=True` adds a medium whose
boundary is a medium (box or sphere innermost; the default leaves the seeds the fixtures were made with as they are).
Test infrastructure only."""
import numpy as np


def random_scene(seed: int, n_inst=None, volume=None, textures=None, nested=False):
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(np.float32(rng.uniform(a, b)))
    col = lambda lo=0.05, hi=0.95: [u(lo, hi), u(lo, hi), u(lo, hi)]
    n_inst = int(rng.integers(3, 40)) if n_inst is None else n_inst
    volume = bool(rng.integers(0, 4) == 0) if volume is None else volume
    textures = bool(rng.integers(0, 3) == 0) if textures is None else textures
    tex = []
    mats = [{"id": f"lam{i}", "type": "lambertian", "data": {"color": col()}} for i in range(int(rng.integers(2, 6)))]
    mats.append({"id": "metal", "type": "metal", "data": {"color": col(0.5, 1.0), "roughness": u(0, 1.5)}})
    mats.append({"id": "glass", "type": "dielectric", "data": {"ior": u(1.1, 1.9)}})
    if textures:
        tex = [{"id": "c0", "type": "constant", "data": {"color": col(), "alpha": u(0.2, 1.0)}},
               {"id": "c1", "type": "constant", "data": {"color": col()}},
               {"id": "chk", "type": "checker", "data": {"scale": u(0.01, 0.2), "odd": {"texture": "c0"}, "even": {"color": col()}}},
               {"id": "noise", "type": "perlin", "data": {"scale": u(0.005, 0.3)}},
               {"id": "chk2", "type": "checker", "data": {"scale": u(0.5, 6.0), "odd": {"texture": "chk"}, "even": {"texture": "noise"}}}]
        mats += [{"id": "tchk", "type": "lambertian", "data": {"texture": "chk"}},
                 {"id": "tnoise", "type": "lambertian", "data": {"texture": "noise"}},
                 {"id": "tc0", "type": "lambertian", "data": {"texture": "c0"}}]
    n_light_mats = int(rng.integers(1, 4))
    for i in range(n_light_mats):
        d = {"color": col(0.5, 1.0), "power": u(2.0, 30.0)}
        if rng.integers(0, 3) == 0:
            d["two_sided"] = False
        if textures and i == 0 and rng.integers(0, 2):
            d = {"texture": "chk", "power": u(2.0, 30.0)}
        mats.append({"id": f"light{i}", "type": "diffuse_light", "data": d})
    surf = [m["id"] for m in mats if m["type"] != "diffuse_light"]
    lights = [m["id"] for m in mats if m["type"] == "diffuse_light"]

    def prim(mat, kind=None):
        kind = kind or rng.choice(["rect", "rect", "box", "sphere"])
        if kind == "rect":
            p = {"type": "rect", "material": {"id": mat}, "size": [u(20, 400), u(20, 400)]}
            a = rng.choice(["xz", "xy", "yz", None])
            if a:
                p["align"] = str(a)
            if rng.integers(0, 2):
                p["flip"] = True
            return p
        if kind == "box":
            return {"type": "box", "material": {"id": mat}, "size": [u(20, 250), u(20, 250), u(20, 250)]}
        return {"type": "sphere", "material": {"id": mat}, "radius": u(10, 120)}

    prims = [dict(prim(str(rng.choice(surf))), id=f"p{i}") for i in range(int(rng.integers(1, 5)))]
    if volume:
        prims.append({"id": "vbox", "type": "box", "size": [u(100, 400), u(100, 400), u(100, 400)]})
        prims.append({"id": "fog", "type": "volume", "primitive": "vbox", "density": u(0.0005, 0.02), "color": col(0.3, 1.0)})

    if nested:   # appended after everything the default draws: earlier seeds keep their scenes
        inner = "nbox" if rng.integers(0, 2) else "nball"
        prims.append({"id": "nbox", "type": "box", "size": [u(100, 300), u(100, 300), u(100, 300)]} if inner == "nbox"
                     else {"id": "nball", "type": "sphere", "radius": u(60, 160), "material": {"id": surf[0]}})
        prims.append({"id": "fog_in", "type": "volume", "primitive": inner, "density": u(0.005, 0.05), "color": col(0.3, 1.0)})
        prims.append({"id": "fog_out", "type": "volume", "primitive": "fog_in", "density": u(0.003, 0.03), "color": col(0.3, 1.0)})

    def xf():
        t = {"translate": [u(0, 555), u(0, 555), u(0, 555)]}
        r = rng.integers(0, 4)
        if r == 1:
            t["rotate"] = [float(rng.choice([0.0, 0.5, 1.0, 1.5])), float(rng.choice([0.0, 0.5, 1.0])), 0.0]
        elif r >= 2:
            t["rotate"] = [u(-1, 1), u(-1, 1), u(-1, 1)]
        s = rng.integers(0, 4)
        if s == 1:
            t["scale"] = u(0.3, 2.0)
        elif s == 2:
            t["scale"] = [u(0.3, 2.0), u(0.3, 2.0), u(0.3, 2.0)]
        return t

    inst = []
    for i in range(n_inst):
        if rng.integers(0, 3) == 0:
            e = {"type": "ref", "primitive": {"id": str(rng.choice([p["id"] for p in prims if p["id"] not in ("vbox", "nbox", "nball", "fog_in")]))},
                 "transform": xf()}
        else:
            e = {"type": "direct", "primitive": prim(str(rng.choice(surf))), "transform": xf()}
        if rng.integers(0, 12) == 0:
            e = {"skip": True, **e}
        inst.append(e)
    for i in range(int(rng.integers(1, 4))):
        kind = "sphere" if rng.integers(0, 3) == 0 else "rect"
        inst.append({"type": "direct", "primitive": prim(str(rng.choice(lights)), kind), "transform": xf()})
    if nested:
        inst.append({"type": "ref", "primitive": {"id": "fog_out"}, "transform": xf()})
    order = rng.permutation(len(inst))
    inst = [inst[k] for k in order]
    world = {"color": col(0.0, 0.4)}
    if textures and rng.integers(0, 2):
        world = {"texture": "chk2"}
    cam = {"look_from": [u(-300, 850), u(50, 500), u(-900, -300)], "look_at": [u(150, 400), u(150, 400), u(150, 400)],
           "fov": u(25, 70), "aperture": float(rng.choice([0.0, 0.0, u(1, 20)])), "dist_to_focus": u(5, 800)}
    return {"camera": cam, "world": world, "assets": [], "textures": tex, "materials": mats, "primitives": prims, "instances": inst}
