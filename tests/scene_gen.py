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


def room_scene(seed: int, clutter: int = 0):
    """A closed room of rects (translated, and rotated by half and three-half turns like the reference's Cornell box builds its
    ceiling and back wall) with one or two rect lights hung close under the ceiling, a block or two, sometimes a fog block and a
    free-standing partition: the scenes whose walls the shadow sweep of the per-scene build proves unreachable instead of testing
    (pt_context.cpp rect_walls_scene) -- with the lights at the distances where that choice flips.  Test infrastructure only."""
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(np.float32(rng.uniform(a, b)))
    col = lambda lo=0.1, hi=0.9: [u(lo, hi), u(lo, hi), u(lo, hi)]
    W, H, D = u(300, 900), u(250, 700), u(300, 900)
    mats = [{"id": "white", "type": "lambertian", "data": {"color": col(0.6, 0.8)}},
            {"id": "red", "type": "lambertian", "data": {"color": [u(0.5, 0.8), 0.05, 0.05]}},
            {"id": "green", "type": "lambertian", "data": {"color": [0.1, u(0.4, 0.6), 0.1]}},
            {"id": "metal", "type": "metal", "data": {"color": col(0.6, 1.0), "roughness": u(0, 1)}},
            {"id": "light", "type": "diffuse_light", "data": {"color": col(0.5, 1.0), "power": u(5, 30),
                                                              **({"two_sided": False} if rng.integers(0, 2) else {})}}]
    prims = [{"id": "floor", "type": "rect", "material": {"id": "white"}, "size": [W, D]},
             {"id": "back", "type": "rect", "material": {"id": "white"}, "size": [W, H]}]
    inst = [{"type": "ref", "primitive": {"id": "floor"}, "transform": {"translate": [W / 2, 0.0, D / 2]}},
            {"type": "ref", "primitive": {"id": "floor"}, "transform": {"rotate": [1.0, 0.0, 0.0], "translate": [W / 2, H, D / 2]}},
            {"type": "ref", "primitive": {"id": "back"}, "transform": {"rotate": [1.5, 0.0, 0.0], "translate": [W / 2, H / 2, D]}},
            {"type": "direct", "primitive": {"type": "rect", "material": {"id": "green"}, "size": [H, D], "align": "yz", "flip": True},
             "transform": {"translate": [W, H / 2, D / 2]}},
            {"type": "direct", "primitive": {"type": "rect", "material": {"id": "red"}, "size": [H, D], "align": "yz"},
             "transform": {"translate": [0.0, H / 2, D / 2]}}]
    if rng.integers(0, 2):   # a front wall too: the camera then sits inside
        inst.append({"type": "direct", "primitive": {"type": "rect", "material": {"id": "white"}, "size": [W, H], "align": "xy"},
                     "transform": {"translate": [W / 2, H / 2, 0.0]}})
    for k in range(int(rng.integers(1, 3))):
        # distance under the ceiling: from far inside the margin of rect_walls_scene (2^-10 of the room) to well outside it
        gap = float(np.float32(rng.choice([0.05, 0.3, 0.6, 1.0, 3.0, 40.0])))
        lw, ld = u(40, W / 3), u(40, D / 3)
        inst.append({"type": "direct", "primitive": {"type": "rect", "material": {"id": "light"}, "size": [lw, ld]},
                     "transform": {"translate": [u(lw, W - lw), H - gap, u(ld, D - ld)]}})
    if rng.integers(0, 6) == 0:   # a sphere light is sampled by direction, its hit is not at t = 1: no wall may be marked then
        inst.append({"type": "direct", "primitive": {"type": "sphere", "material": {"id": "light"}, "radius": u(10, 30)},
                     "transform": {"translate": [u(W / 4, 3 * W / 4), u(H / 2, 3 * H / 4), u(D / 4, 3 * D / 4)]}})
    for k in range(int(rng.integers(1, 3))):
        s = [u(40, W / 4), u(40, H / 2), u(40, D / 4)]
        hd = 0.5 * float(np.hypot(s[0], s[2])) + 2.0   # the block turns about y inside the room, whatever the angle
        inst.append({"type": "direct", "primitive": {"type": "box", "material": {"id": str(rng.choice(["white", "metal"]))}, "size": s},
                     "transform": {"translate": [u(hd, W - hd), s[1] / 2, u(hd, D - hd)], "rotate": [0.0, u(-0.3, 0.3), 0.0]}})
    for k in range(clutter):   # many small blocks on the floor: more than 24 instances, the fast program keeps its tree (drawn only when asked for)
        s = [u(8, 30), u(8, 60), u(8, 30)]
        inst.append({"type": "direct", "primitive": {"type": "box", "material": {"id": str(rng.choice(["white", "metal", "red"]))}, "size": s},
                     "transform": {"translate": [u(40, W - 40), s[1] / 2, u(40, D - 40)]}})
    if rng.integers(0, 3) == 0:
        prims.append({"id": "vbox", "type": "box", "size": [u(60, W / 3), u(60, H / 2), u(60, D / 3)]})
        prims.append({"id": "fog", "type": "volume", "primitive": "vbox", "density": u(0.002, 0.02), "color": col(0.5, 1.0)})
        inst.append({"type": "ref", "primitive": {"id": "fog"}, "transform": {"translate": [u(W / 4, 3 * W / 4), H / 3, u(D / 4, 3 * D / 4)]}})
    if rng.integers(0, 2):   # a partition inside the room: shadow rays do cross its plane, it must stay a tested leaf
        inst.append({"type": "direct", "primitive": {"type": "rect", "material": {"id": "white"}, "size": [u(50, H / 2), u(50, D / 2)], "align": "yz"},
                     "transform": {"translate": [u(W / 4, 3 * W / 4), H / 4, D / 2]}})
    order = rng.permutation(len(inst))
    inst = [inst[k] for k in order]
    inside = bool(rng.integers(0, 2))
    cam = {"look_from": [u(W / 4, 3 * W / 4), u(H / 4, 3 * H / 4), u(5, D / 4) if inside else -u(300, 900)],
           "look_at": [u(W / 4, 3 * W / 4), u(H / 4, 3 * H / 4), D], "fov": u(30, 70), "aperture": 0.0, "dist_to_focus": 10.0}
    return {"camera": cam, "world": {"color": [0.0, 0.0, 0.0]}, "assets": [], "textures": [], "materials": mats, "primitives": prims, "instances": inst}
