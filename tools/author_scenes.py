#!/usr/bin/env python3
"""Re-author the three BASELINE Cornell-box scenes in the reference's scene-JSON schema.

The schema is the one SURVEY.md A.1 documents (reference scene_parser.h:241-595,
main.cpp:86-104); the numeric content is the one SURVEY.md A.6 lists.  Nothing is read
from /root/reference: the scenes are 8-9 instances and are restated here as data.
Run:  python tools/author_scenes.py   (writes scenes/*.json)
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "scenes")


def lambertian(mid, rgb):
    return {"id": mid, "type": "lambertian", "data": {"color": list(rgb)}}


def light_mat(rgb):
    return {"id": "light", "type": "diffuse_light", "data": {"color": list(rgb)}}


def ref(pid, **transform):
    return {"type": "ref", "primitive": {"id": pid}, "transform": transform}


def direct(prim, skip=False, **transform):
    d = {"type": "direct", "primitive": prim, "transform": transform}
    if skip:
        d = {"skip": True, **d}
    return d


def walls(short_box):
    """The five walls + the two boxes; `short_box` is the list of instance(s) standing in
    the short-box slot (file order matters: it is the BVH input order)."""
    return [
        ref("white_wall", translate=[277.5, 0.0, 277.5]),
        ref("white_wall", rotate=[1.0, 0.0, 0.0], translate=[277.5, 555, 277.5]),
        ref("white_wall", rotate=[1.5, 0, 0], translate=[277.5, 277.5, 555]),
        direct({"type": "rect", "material": {"id": "green"}, "size": [555, 555],
                "align": "yz", "flip": True}, translate=[555, 277.5, 277.5]),
        direct({"type": "rect", "material": {"id": "red"}, "size": [555, 555],
                "align": "yz"}, translate=[0, 277.5, 277.5]),
        *short_box,
        direct({"type": "box", "material": {"id": "white"}, "size": [165, 330, 165]},
               translate=[347.5, 165, 377.5], rotate=[0.0, 0.05, 0.0]),
    ]


def rect_light(w, h, at):
    return direct({"type": "rect", "material": {"id": "light"}, "size": [w, h]}, translate=list(at))


SKIPPED_SPHERE = direct({"type": "sphere", "material": {"id": "light"}}, skip=True,
                        scale=[100.0, 20.0, 100.0], translate=[273, 200, 171])

SHORT_BOX_XF = dict(translate=[212.5, 82.5, 147.5], rotate=[0.0, -0.1, 0.0])


def camera(z):
    return {"look_from": [278.0, 278.0, z], "look_at": [278.0, 278.0, 0.0], "fov": 40.0,
            "aperture": 0.0, "dist_to_focus": 10.0}


BASE_PRIMS = [
    {"id": "white_wall", "type": "rect", "material": {"id": "white"}, "size": [555, 555]},
    {"id": "box", "type": "box", "material": {"id": "white"}, "size": [165, 165, 165]},
]

scenes = {}

scenes["cornell_box"] = {
    "camera": camera(-750.0),
    "world": {"color": [0.0, 0.0, 0.0]},
    "assets": [], "textures": [],
    "materials": [lambertian("green", (0.12, 0.85, 0.05)), lambertian("red", (0.95, 0.05, 0.05)),
                  lambertian("white", (0.73, 0.73, 0.73)), light_mat((0.6, 0.6, 0.6))],
    "primitives": BASE_PRIMS,
    "instances": walls([ref("box", **SHORT_BOX_XF)]) + [rect_light(240, 230, (273, 554.0, 171)), SKIPPED_SPHERE],
}

scenes["cornell_box_small_lights"] = {
    "camera": camera(-750.0),
    "world": {"color": [0.0, 0.0, 0.0]},
    "assets": [], "textures": [],
    "materials": [lambertian("green", (0.12, 0.45, 0.15)), lambertian("red", (0.65, 0.05, 0.05)),
                  lambertian("white", (0.73, 0.73, 0.73)), light_mat((15.0, 15.0, 15.0))],
    "primitives": BASE_PRIMS,
    "instances": walls([ref("box", **SHORT_BOX_XF)]) + [rect_light(30, 30, (243, 554.0, 171)),
                                                          rect_light(30, 30, (303, 554.0, 171)), SKIPPED_SPHERE],
}

scenes["cornell_box_with_volume"] = {
    "camera": camera(-700.0),
    "world": {"color": [0.1, 0.1, 0.1]},
    "assets": [], "textures": [],
    "materials": [lambertian("green", (0.12, 0.45, 0.15)), lambertian("red", (0.65, 0.05, 0.05)),
                  lambertian("white", (0.73, 0.73, 0.73)),
                  # the reference parser has no "isotropic" case, so this entry is ignored there
                  {"id": "isotropic", "type": "isotropic", "data": {"color": [0.4, 0.4, 0.4], "density": 0.004}},
                  light_mat((1.0, 1.0, 1.0))],
    "primitives": [
        BASE_PRIMS[0],
        {"id": "box", "type": "box", "size": [165, 165, 165]},
        {"id": "fog", "type": "volume", "primitive": "box", "density": 0.004, "color": [0.9, 0.9, 0.9]},
    ],
    "instances": walls([{"skip": True, **ref("box", **SHORT_BOX_XF)}, ref("fog", **SHORT_BOX_XF)])
    + [rect_light(240, 230, (273, 554.0, 171)), SKIPPED_SPHERE],
}

# ---- scenes beyond the BASELINE configs (sphere lights, metal, dielectric, a room-filling volume) ----
def cam(frm, at, fov, focus):
    return {"look_from": list(frm), "look_at": list(at), "fov": fov, "aperture": 0.0, "dist_to_focus": focus}


def direct_notransform(prim):
    return {"type": "direct", "primitive": prim}


scenes["light_test"] = {
    "camera": cam((0.0, 80.0, -80.0), (0.0, 0.0, 0.0), 40.0, 40.0),
    "world": {"color": [0.01, 0.01, 0.01]},
    "assets": [], "textures": [],
    "materials": [lambertian("white", (0.7, 0.7, 0.7)),
                  {"id": "metal1", "type": "metal", "data": {"color": [0.9, 0.9, 0.9], "roughness": 0.1}},
                  {"id": "metal2", "type": "metal", "data": {"color": [0.9, 0.9, 0.9], "roughness": 0.2}},
                  {"id": "metal3", "type": "metal", "data": {"color": [0.9, 0.9, 0.9], "roughness": 0.3}},
                  light_mat((25.0, 25.0, 25.0))],
    "primitives": [{"id": "wall", "type": "rect", "material": {"id": "white"}, "size": [80, 80]}],
    "instances": [
        {"type": "ref", "primitive": {"id": "wall"}},
        ref("wall", rotate=[-0.5, 0.0, 0.0], translate=[0.0, 40.0, 40.0]),
        direct({"type": "sphere", "material": {"id": "light"}, "radius": 2}, translate=[-15, 20.0, 4.0]),
        direct({"type": "sphere", "material": {"id": "light"}, "radius": 1.5}, translate=[-5, 20.0, 4.0]),
        direct({"type": "sphere", "material": {"id": "light"}, "radius": 1}, translate=[5, 20.0, 4.0]),
        direct({"type": "sphere", "material": {"id": "light"}, "radius": 0.5}, translate=[15, 20.0, 4.0]),
        direct({"type": "rect", "material": {"id": "metal3"}, "size": [40, 4.8]}, rotate=[-0.0922222, 0.0, 0.0], translate=[0, 3, 0]),
        direct({"type": "rect", "material": {"id": "metal2"}, "size": [40, 4]}, rotate=[-0.155, 0.0, 0.0], translate=[0, 4, 6]),
        direct({"type": "rect", "material": {"id": "metal1"}, "size": [40, 4]}, rotate=[-0.2177777, 0.0, 0.0], translate=[0, 10, 8]),
    ],
}

scenes["three_orbs"] = {
    "camera": cam((0.0, 20.0, -40.0), (0.0, 0.0, 0.0), 40.0, 40.0),
    "world": {"color": [0.0, 0.0, 0.0]},
    "assets": [], "textures": [],
    "materials": [lambertian("green", (0.12, 0.45, 0.15)), lambertian("white", (0.7, 0.7, 0.7)),
                  {"id": "metal", "type": "metal", "data": {"color": [0.9, 0.9, 0.9]}},
                  {"id": "glass", "type": "dielectric", "data": {"color": [0.9, 0.9, 0.9]}},
                  light_mat((25.0, 25.0, 25.0))],
    "primitives": [],
    "instances": [
        direct_notransform({"type": "rect", "material": {"id": "white"}, "size": [80, 80]}),
        direct({"type": "rect", "material": {"id": "light"}, "size": [5, 5]}, translate=[0, 20.0, 10]),
        direct({"type": "sphere", "material": {"id": "green"}, "radius": 4}, translate=[-10, 4, 0]),
        direct({"type": "sphere", "material": {"id": "glass"}, "radius": 4}, translate=[0, 4, 0]),
        direct({"type": "sphere", "material": {"id": "metal"}, "radius": 4}, translate=[10, 4, 0]),
    ],
}

scenes["cornell_box_with_volume2"] = {
    "camera": camera(-700.0),
    "world": {"color": [0.0, 0.0, 0.0]},
    "assets": [], "textures": [],
    "materials": [lambertian("green", (0.12, 0.45, 0.15)), lambertian("red", (0.65, 0.05, 0.05)),
                  lambertian("white", (0.73, 0.73, 0.73)),
                  {"id": "isotropic", "type": "isotropic", "data": {"color": [0.93, 0.93, 0.93], "density": 0.0004}},
                  light_mat((1.0, 1.0, 1.0))],
    "primitives": [
        BASE_PRIMS[0],
        {"id": "box", "type": "box", "size": [555, 555, 555]},
        {"id": "fog", "type": "volume", "primitive": "box", "density": 0.001, "color": [0.9, 0.9, 0.9]},
    ],
    "instances": walls([{"skip": True, **ref("box", **SHORT_BOX_XF)}, ref("fog", translate=[277.5, 277.5, 277.5])])
    + [rect_light(240, 230, (273, 554.0, 171)), SKIPPED_SPHERE],
}

# A medium whose boundary is a medium (volume.h:10: constant_medium takes ANY hittable as its boundary; the scene format's
# "primitive" id of a volume may name an earlier volume, scene_parser.h:214-232): the outer medium's two boundary queries
# are two scattering events of the inner one, each with a free-flight draw of its own, then the outer's own draw between
# them.  One with a box as the innermost boundary (the short block's place), one with a sphere floating beside it.
scenes["cornell_box_nested_fog"] = {
    "camera": camera(-700.0),
    "world": {"color": [0.05, 0.05, 0.08]},
    "assets": [], "textures": [],
    "materials": [lambertian("green", (0.12, 0.45, 0.15)), lambertian("red", (0.65, 0.05, 0.05)),
                  lambertian("white", (0.73, 0.73, 0.73)), light_mat((1.0, 1.0, 1.0))],
    "primitives": [
        BASE_PRIMS[0],
        {"id": "box", "type": "box", "size": [165, 165, 165]},
        {"id": "fog_in", "type": "volume", "primitive": "box", "density": 0.02, "color": [0.9, 0.9, 0.9]},
        {"id": "fog_out", "type": "volume", "primitive": "fog_in", "density": 0.01, "color": [0.7, 0.85, 1.0]},
        {"id": "ball", "type": "sphere", "radius": 80, "material": {"id": "white"}},
        {"id": "mist_in", "type": "volume", "primitive": "ball", "density": 0.03, "color": [0.9, 0.9, 0.9]},
        {"id": "mist_out", "type": "volume", "primitive": "mist_in", "density": 0.02, "color": [1.0, 0.8, 0.6]},
    ],
    "instances": walls([{"skip": True, **ref("box", **SHORT_BOX_XF)}, ref("fog_out", **SHORT_BOX_XF)])
    + [ref("mist_out", translate=[150.0, 380.0, 330.0]), rect_light(240, 230, (273, 554.0, 171)), SKIPPED_SPHERE],
}

# The reference's seventh scene, cornell_box_image_light.json.  Its "png" texture points at
# assets/light_texture.png, which the reference repository does not contain, so the reference cannot load the file as
# shipped; here that one texture entry carries the parser's own "skip" flag (scene_parser.h:265).  Nothing else refers to
# it (the keys spelled "texture2" are ignored by the parser), so the scene is otherwise the reference's: a checker
# lambertian on a 10 km sphere under the floor, a metal sphere, constant textures referenced by id, an unused perlin one.
def const_tex(tid, rgb, alpha=1.0):
    return {"id": tid, "type": "constant", "data": {"color": list(rgb), "alpha": alpha}}


scenes["cornell_box_image_light"] = {
    "camera": cam((-100.0, 400.0, -750.0), (278.0, 278.0, 0.0), 70.0, 100.0),
    "world": {"color": [0.0, 0.0, 0.0], "texture2": "checker_texture"},
    "assets": [],
    "textures": [
        {"skip": True, "id": "light_texture", "type": "png", "data": {"path": "assets/light_texture.png"}},
        const_tex("green", (0.12, 0.85, 0.05)), const_tex("red", (1.0, 0.05, 0.05)),
        const_tex("transparent_red", (1.0, 0.05, 0.05), 0.5), const_tex("blue", (0.05, 0.05, 1.0)),
        {"id": "checker_texture", "type": "checker", "data": {"scale": 0.05, "odd": {"texture": "red"}, "even": {"texture": "blue"}}},
        {"id": "noise_texture", "type": "perlin", "data": {"scale": 0.1}},
    ],
    "materials": [
        {"id": "green", "type": "lambertian", "data": {"texture": "green"}},
        {"id": "red", "type": "lambertian", "data": {"texture": "red"}},
        {"id": "white", "type": "lambertian", "data": {"texture2": "checker_texture", "color": [0.73, 0.73, 0.73]}},
        {"id": "checker_light", "type": "diffuse_light", "data": {"texture": "checker_texture", "power": 5}},
        {"id": "noise", "type": "lambertian", "data": {"texture": "noise_texture"}},
        {"id": "glass", "type": "dielectric", "data": {"ior": 1.5}},
        {"id": "metal", "type": "metal", "data": {"roughness": 0.5}},
        {"id": "checker", "type": "lambertian", "data": {"texture": "checker_texture"}},
        {"id": "light", "type": "diffuse_light", "data": {"color": [1, 1, 1], "texture2": "light_texture", "power": 1, "two_sided": True}},
    ],
    "primitives": [
        {"id": "white_wall", "type": "rect", "material": {"id": "white"}, "size": [555, 555]},
        {"id": "box", "type": "box", "material": {"id": "white"}, "size": [165, 165, 165]},
        {"id": "sphere", "type": "sphere", "material": {"id": "metal"}, "radius": 67.5},
        {"id": "big_sphere", "type": "sphere", "material": {"id": "checker"}, "radius": 10000.0},
    ],
    "instances": [
        ref("white_wall", translate=[277.5, 0.0, 277.5], rotate=[0.0, 0.0, 0.0]),
        ref("white_wall", rotate=[1.0, 0.0, 0.0], translate=[277.5, 555, 277.5]),
        ref("white_wall", rotate=[1.5, 0, 0], translate=[277.5, 277.5, 555]),
        direct({"type": "rect", "material": {"id": "green"}, "size": [555, 555], "align": "yz", "flip": True},
               translate=[555, 277.5, 277.5]),
        direct({"type": "rect", "material": {"id": "red"}, "size": [555, 555], "align": "yz"},
               translate=[0, 277.5, 277.5], rotate=[0.0, 0.0, 0.0]),
        ref("sphere", translate=[212.5, 82.5, 147.5], rotate=[0.0, -0.1, 0.0]),
        ref("big_sphere", translate=[277.5, -10000.0, 277.5]),
        direct({"type": "box", "material": {"id": "white"}, "size": [165, 330, 165]},
               translate=[347.5, 165, 377.5], rotate=[0.0, 0.05, 0.0]),
        direct({"type": "rect", "material": {"id": "light"}, "size": [240, 230]}, translate=[273, 520.0, 171], rotate=[1.0, 0, 0]),
    ],
}

# Not a reference scene: a coverage scene of our own for the texture row, in the reference's schema and
# rendered by the real reference for the fixtures like the others -- perlin-noise floor, checker wall whose children are
# textures by id, a checker-textured emitter with a half-transparent child (alpha scales the emission, material.h:218), a
# second plain light, and a checker sky as World::background seen through the missing right wall and ceiling
# (integrator.h:325-336).
scenes["textured_room"] = {
    "camera": camera(-750.0),
    "world": {"texture": "sky"},
    "assets": [],
    "textures": [
        const_tex("white", (0.73, 0.73, 0.73)), const_tex("green", (0.12, 0.45, 0.15)),
        const_tex("warm", (1.0, 0.55, 0.2)), const_tex("cool_half", (0.3, 0.5, 1.0), 0.5),
        {"id": "tiles", "type": "checker", "data": {"scale": 0.04, "odd": {"texture": "white"}, "even": {"texture": "green"}}},
        {"id": "lamp", "type": "checker", "data": {"scale": 0.09, "odd": {"texture": "warm"}, "even": {"texture": "cool_half"}}},
        {"id": "marble", "type": "perlin", "data": {"scale": 0.03}},
        {"id": "sky", "type": "checker", "data": {"scale": 4.0, "odd": {"color": [0.35, 0.45, 0.7]}, "even": {"color": [0.08, 0.08, 0.12]}}},
        {"id": "nested", "type": "checker", "data": {"scale": 0.011, "odd": {"texture": "tiles"}, "even": {"texture": "marble"}}},
    ],
    "materials": [
        lambertian("red", (0.65, 0.05, 0.05)),
        {"id": "tiles", "type": "lambertian", "data": {"texture": "tiles"}},
        {"id": "marble", "type": "lambertian", "data": {"texture": "marble"}},
        {"id": "nested", "type": "lambertian", "data": {"texture": "nested"}},
        {"id": "lamp", "type": "diffuse_light", "data": {"texture": "lamp", "power": 6}},
        {"id": "light", "type": "diffuse_light", "data": {"color": [9.0, 9.0, 9.0], "two_sided": False}},
    ],
    "primitives": [
        {"id": "floor", "type": "rect", "material": {"id": "marble"}, "size": [555, 555]},
        {"id": "back", "type": "rect", "material": {"id": "tiles"}, "size": [555, 555]},
    ],
    "instances": [
        ref("floor", translate=[277.5, 0.0, 277.5]),
        ref("back", rotate=[1.5, 0, 0], translate=[277.5, 277.5, 555]),
        direct({"type": "rect", "material": {"id": "red"}, "size": [555, 555], "align": "yz"}, translate=[0, 277.5, 277.5]),
        direct({"type": "box", "material": {"id": "nested"}, "size": [165, 330, 165]},
               translate=[347.5, 165, 377.5], rotate=[0.0, 0.05, 0.0]),
        direct({"type": "sphere", "material": {"id": "tiles"}, "radius": 70}, translate=[190, 70, 190]),
        direct({"type": "rect", "material": {"id": "lamp"}, "size": [260, 220]}, translate=[250, 500.0, 250], rotate=[1.0, 0, 0]),
        direct({"type": "rect", "material": {"id": "light"}, "size": [60, 60], "flip": True}, translate=[420, 450.0, 120]),
    ],
}

# ---- image textures (image.h): small PNG assets written here, pixel content from closed formulas ----------------------
def write_png(path, w, h, ctype, pixel, palette=None, trns=None, level=6, strategy=0):
    """8-bit non-interlaced PNG; rows cycle through the five filter types so a reader has to undo all of them."""
    import struct
    import zlib
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    rows = [[pixel(x, y) for x in range(w)] for y in range(h)]
    raw = bytearray()
    prev = [0] * (w * ch)
    for y in range(h):
        cur = [c for px in rows[y] for c in px]
        ft = y % 5
        raw.append(ft)
        for i, v in enumerate(cur):
            a = cur[i - ch] if i >= ch else 0
            b = prev[i]
            c = prev[i - ch] if i >= ch else 0
            if ft == 0:
                pr = 0
            elif ft == 1:
                pr = a
            elif ft == 2:
                pr = b
            elif ft == 3:
                pr = (a + b) >> 1
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            raw.append((v - pr) & 255)
        prev = cur

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xffffffff)
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
    data = co.compress(bytes(raw)) + co.flush()
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(c for rgb in palette for c in rgb))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    out += chunk(b"IDAT", data[:len(data) // 2]) + chunk(b"IDAT", data[len(data) // 2:]) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(out)


def write_assets():
    import zlib
    adir = os.path.join(HERE, "..", "assets")
    os.makedirs(adir, exist_ok=True)
    # poster: RGBA, colour ramps with a transparent-ish diagonal band
    write_png(os.path.join(adir, "poster.png"), 16, 12, 6,
              lambda x, y: (40 + 13 * x, 250 - 17 * y, (x * y * 7) % 256, 255 if (x + y) % 5 else 96))
    # sky: RGB environment map, brighter towards the top rows
    write_png(os.path.join(adir, "sky.png"), 32, 16, 2,
              lambda x, y: (30 + 6 * y + (x % 4) * 9, 60 + 9 * y, 235 - 5 * y - (x % 3) * 20))
    # grey + alpha, fixed-Huffman deflate blocks (reader test; an image-textured emitter crashes the reference)
    write_png(os.path.join(adir, "lamp.png"), 8, 8, 4, lambda x, y: (255 - 20 * ((x + y) % 4), 255 if (x ^ y) & 1 else 128),
              strategy=zlib.Z_FIXED)
    # decoder-only cases: palette + tRNS (stored deflate blocks), grey with a transparent key
    write_png(os.path.join(adir, "palette.png"), 7, 5, 3, lambda x, y: ((x + 2 * y) % 6,),
              palette=[(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (9, 99, 199), (0, 0, 0)], trns=[255, 128, 0], level=0)
    write_png(os.path.join(adir, "grey.png"), 9, 4, 0, lambda x, y: ((x * 31 + y * 17) % 256,), trns=[0, 48])


# Our own coverage scene for image textures: an image-textured lambertian floor and box (rect u, v incl. the reference's
# v = (zh - x0) / (z1 - z0) slip, primitive.h:207), an image inside a checker, and an image as World::background looked
# up by direction (integrator.h:327-332 with TAU = 2 * M_PI unparenthesised).  The light is NOT image-textured: the
# reference crashes on that as soon as a path lands on the light (its NEE ray lies in the light's plane, rect::hit
# accepts the NaN t, and image_texture indexes with NaN u, v -- image.h:46); the product refuses such a scene.
scenes["image_room"] = {
    "camera": camera(-750.0),
    "world": {"texture": "sky"},
    "assets": [],
    "textures": [
        {"id": "poster", "type": "png", "data": {"path": "assets/poster.png"}},
        {"id": "sky", "type": "png", "data": {"path": "assets/sky.png"}},
        const_tex("white", (0.73, 0.73, 0.73)),
        {"id": "mix", "type": "checker", "data": {"scale": 0.03, "odd": {"texture": "poster"}, "even": {"texture": "white"}}},
    ],
    "materials": [
        lambertian("red", (0.65, 0.05, 0.05)),
        {"id": "poster", "type": "lambertian", "data": {"texture": "poster"}},
        {"id": "mix", "type": "lambertian", "data": {"texture": "mix"}},
        {"id": "lamp", "type": "diffuse_light", "data": {"color": [1.0, 0.9, 0.8], "power": 7}},
    ],
    "primitives": [
        {"id": "floor", "type": "rect", "material": {"id": "poster"}, "size": [555, 555]},
        {"id": "back", "type": "rect", "material": {"id": "mix"}, "size": [555, 555]},
    ],
    "instances": [
        ref("floor", translate=[277.5, 0.0, 277.5]),
        ref("back", rotate=[1.5, 0, 0], translate=[277.5, 277.5, 555]),
        direct({"type": "rect", "material": {"id": "red"}, "size": [555, 555], "align": "yz"}, translate=[0, 277.5, 277.5]),
        direct({"type": "box", "material": {"id": "poster"}, "size": [165, 330, 165]},
               translate=[347.5, 165, 377.5], rotate=[0.0, 0.05, 0.0]),
        direct({"type": "rect", "material": {"id": "lamp"}, "size": [260, 220]}, translate=[250, 500.0, 250], rotate=[1.0, 0, 0]),
    ],
}

if __name__ == "__main__":
    write_assets()
    os.makedirs(OUT, exist_ok=True)
    for name, sc in scenes.items():
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(sc, f, indent=1)
            f.write("\n")
        print("wrote", name)
