"""TEST INFRASTRUCTURE (oracle side) -- scene-JSON -> constructor parameters.

Restates, in Python, what the reference's scene loader does with a scene JSON *up to the
point where it calls the geometry constructors* (reference scene_parser.h:104-239 for
primitives, :241-595 for build_scene, main.cpp:86-104 for the camera).  The output is a
plain list of "constructor calls" (materials, primitives, instances in BVH input order,
light flags, camera, background) that

  * `oracle/pt_oracle.c` consumes through ctypes (it then restates transform3 / instance /
    bvh_node / camera constructors itself), and
  * `oracle/ref_driver.cpp` (the real reference headers, built into oracle/_ref/) consumes
    as a text file, so that both see bit-identical float32 inputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product has its own C++ parser (pathtrace_amd/csrc/host/scene_loader.cpp); the two are
compared against each other in tests/test_scene_flatten.py.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

F = np.float32

# material type codes (shared with pt_oracle.h and include/pathtrace_hip.h)
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = 0, 1, 2, 3, 4
# primitive type codes
PRIM_RECT, PRIM_BOX, PRIM_SPHERE, PRIM_VOLUME = 0, 1, 2, 3
# rect plane codes: reference primitive.h:11-16 (enum plane_enum { XY, XZ, YZ })
PLANE_XY, PLANE_XZ, PLANE_YZ = 0, 1, 2

# texture type codes (reference scene.h:63-69 texture_type)
TEX_CONSTANT, TEX_CHECKER, TEX_PERLIN, TEX_IMAGE = 0, 1, 2, 3

MAUVE = (F(0.8), F(0.2), F(0.8))  # scene_parser.h:16


def f32(x) -> np.float32:
    return F(float(x))


def vec3(a):
    return (f32(a[0]), f32(a[1]), f32(a[2]))


@dataclass
class Texture:
    """texture.h: constant_texture :14-31, checker_texture :33-75, noise_texture :185-196; image.h:7-50 image_texture."""
    type: int
    color: tuple = (F(0), F(0), F(0))
    alpha: np.float32 = F(1.0)
    even: int = -1          # checker: texture indices
    odd: int = -1
    scale: np.float32 = F(1.0)
    width: int = 0          # image: decoded RGBA8, row 0 first (lodepng order)
    height: int = 0
    rgba: Optional[bytes] = None


@dataclass
class Material:
    type: int
    color: tuple = (F(0), F(0), F(0))
    alpha: np.float32 = F(1.0)
    power: np.float32 = F(1.0)
    two_sided: bool = True
    fuzz: np.float32 = F(0.0)
    ior: np.float32 = F(1.45)
    json_type: str = "lambertian"  # the *type string* the parser keeps (light test, scene_parser.h:541)
    texture: int = -1       # index into SceneParams.textures when albedo / emit is not a constant texture


@dataclass
class Prim:
    type: int
    mat: int = -1
    # rect (reference primitive.h:120-138): x0,z0,x1,z1,y in the XZ-canonical frame
    rect: tuple = (F(0),) * 5
    plane: int = PLANE_XZ
    flipped: bool = False
    # box (primitive.h:229-242)
    p0: tuple = (F(0),) * 3
    p1: tuple = (F(0),) * 3
    # sphere
    center: tuple = (F(0),) * 3
    radius: np.float32 = F(1.0)
    # volume (volume.h:10-17): boundary prim index, density, phase material index
    boundary: int = -1
    density: np.float32 = F(0)
    phase_mat: int = -1


@dataclass
class Instance:
    prim: int
    scale: tuple = (F(1), F(1), F(1))
    rotate: tuple = (F(0), F(0), F(0))
    translate: tuple = (F(0), F(0), F(0))
    is_light: bool = False


@dataclass
class Camera:
    look_from: tuple
    look_at: tuple
    fov: np.float32
    aperture: np.float32
    dist_to_focus: np.float32


@dataclass
class SceneParams:
    materials: List[Material] = field(default_factory=list)
    prims: List[Prim] = field(default_factory=list)
    instances: List[Instance] = field(default_factory=list)
    camera: Optional[Camera] = None
    background: tuple = MAUVE
    textures: List[Texture] = field(default_factory=list)
    background_texture: int = -1   # World::background when it is not a constant texture


class _Builder:
    def __init__(self):
        self.sp = SceneParams()
        self.textures: Dict[str, int] = {}        # id -> index into sp.textures
        self.base_dir = "."
        self.materials: Dict[str, int] = {}       # id -> material index
        self.prims: Dict[str, int] = {}           # id -> prim index (-1 = null hittable)
        self.last_id = 0
        self._error_mat: Optional[int] = None
        self._error_tex: Optional[int] = None

    # scene_parser.h:20-24 / 92-96: one shared mauve lambertian
    def error_material(self) -> int:
        if self._error_mat is None:
            self.sp.materials.append(Material(MAT_LAMBERTIAN, MAUVE, json_type="lambertian"))
            self._error_mat = len(self.sp.materials) - 1
        return self._error_mat

    def new_id(self) -> str:  # scene_parser.h:26-32
        s = str(self.last_id)
        self.last_id += 1
        return s

    def add_material(self, mid: str, idx: int):
        # std::map::emplace keeps the first entry on duplicate ids
        if mid not in self.materials:
            self.materials[mid] = idx

    def add_texture(self, t: Texture) -> int:
        self.sp.textures.append(t)
        return len(self.sp.textures) - 1

    def error_texture(self) -> int:   # scene_parser.h:98-102: one shared mauve constant texture
        if self._error_tex is None:
            self._error_tex = self.add_texture(Texture(TEX_CONSTANT, MAUVE))
        return self._error_tex

    def parse_textures(self, scene):
        """scene_parser.h:263-330; std::map::emplace keeps the first entry on duplicate ids."""
        for el in scene.get("textures", []) or []:
            if el.get("skip", False):
                continue
            tid = el["id"]
            if "data" not in el:
                self.textures.setdefault(tid, self.error_texture())
                continue
            data = el["data"]
            ttype = el["type"] if el["type"] in ("constant", "checker", "perlin", "png") else "constant"   # scene.h:71-79
            if ttype == "constant":
                idx = self.add_texture(Texture(TEX_CONSTANT, vec3(data["color"]), f32(data.get("alpha", 1.0))))
            elif ttype == "checker":
                def child(d):
                    if "texture" in d:
                        return self.textures[d["texture"]]
                    return self.add_texture(Texture(TEX_CONSTANT, vec3(d["color"])))
                odd = child(data["odd"])       # scene_parser.h:292-309: odd first, then even
                even = child(data["even"])
                idx = self.add_texture(Texture(TEX_CHECKER, even=even, odd=odd, scale=f32(data["scale"])))
            elif ttype == "perlin":
                idx = self.add_texture(Texture(TEX_PERLIN, scale=f32(data.get("scale", 1.0))))
            else:
                w, h, rgba = decode_png(os.path.join(self.base_dir, data["path"]))
                idx = self.add_texture(Texture(TEX_IMAGE, width=w, height=h, rgba=rgba))
            self.textures.setdefault(tid, idx)

    def texture_ref(self, tid: str):
        """(inline colour, alpha, texture index): constant textures are folded into the material like before."""
        idx = self.textures[tid]
        t = self.sp.textures[idx]
        if t.type == TEX_CONSTANT:
            return t.color, t.alpha, -1
        return (F(0), F(0), F(0)), F(1.0), idx

    def parse_materials(self, scene):
        for el in scene.get("materials", []) or []:
            if el.get("skip", False):
                continue
            mid = el["id"]
            if "data" not in el:
                self.add_material(mid, self.error_material())
                continue
            data = el["data"]
            mtype = el["type"]
            known = {"lambertian", "metal", "dielectric", "isotropic", "diffuse_light"}
            if mtype not in known:
                mtype = "lambertian"  # std::map::operator[] default-constructs enum value 0
            if mtype == "lambertian":
                if "color" in data:
                    m = Material(MAT_LAMBERTIAN, vec3(data["color"]), json_type="lambertian")
                elif "texture" in data:
                    col, a, ti = self.texture_ref(data["texture"])
                    m = Material(MAT_LAMBERTIAN, col, alpha=a, json_type="lambertian", texture=ti)
                else:
                    self.add_material(mid, self.error_material())
                    continue
            elif mtype == "metal":
                col = vec3(data["color"]) if "color" in data else (F(1), F(1), F(1))
                fz = f32(data.get("roughness", 0.0))
                m = Material(MAT_METAL, col, fuzz=fz if fz < 1 else F(1), json_type="metal")
            elif mtype == "dielectric":
                # attenuation of dielectric::scatter is (1,1,1) (material.h:118-124); "color" is announced as unsupported
                m = Material(MAT_DIELECTRIC, (F(1), F(1), F(1)), ior=f32(data["ior"]) if "ior" in data else F(1.450),
                             json_type="dielectric")
            elif mtype == "diffuse_light":
                power = f32(data["power"]) if "power" in data else F(1.0)
                two_sided = bool(data.get("two_sided", True))
                if "texture" in data:
                    col, a, ti = self.texture_ref(data["texture"])
                else:
                    col, a, ti = (vec3(data["color"]) if "color" in data else (F(1), F(1), F(1))), F(1.0), -1
                m = Material(MAT_DIFFUSE_LIGHT, col, alpha=a, power=power, two_sided=two_sided,
                             json_type="diffuse_light", texture=ti)
            else:  # "isotropic": the reference's switch has no case for it (scene_parser.h:444)
                continue
            self.sp.materials.append(m)
            self.add_material(mid, len(self.sp.materials) - 1)

    def parse_prim(self, el) -> int:
        """scene_parser.h:104-239.  Returns prim index, or -1 for a null hittable."""
        if "material" in el and isinstance(el["material"], dict) and "id" in el["material"]:
            mat = self.materials[el["material"]["id"]]  # assert(materials.count(id) > 0)
        else:
            mat = self.error_material()
        ptype = el["type"]
        if ptype == "sphere":
            p = Prim(PRIM_SPHERE, mat, radius=f32(el.get("radius", 1.0)),
                     center=vec3(el["origin"]) if "origin" in el else (F(0), F(0), F(0)))
        elif ptype == "rect":
            plane = {"xy": PLANE_XY, "xz": PLANE_XZ, "yz": PLANE_YZ}.get(el.get("align", "xz"), PLANE_XY)
            flipped = bool(el.get("flip", False))
            if all(k in el for k in ("a0", "b0", "a1", "b1")):
                r = (f32(el["a0"]), f32(el["b0"]), f32(el["a1"]), f32(el["b1"]), f32(el["c"]))
            else:
                a, b = (f32(el["size"][0]), f32(el["size"][1])) if "size" in el else (F(1), F(1))
                # rect(x, z, mat, ...) : rect(-x / 2.0, -z / 2.0, x / 2.0, z / 2.0, 0.0, ...)  primitive.h:126-130
                r = (F(-float(a) / 2.0), F(-float(b) / 2.0), F(float(a) / 2.0), F(float(b) / 2.0), F(0.0))
            p = Prim(PRIM_RECT, mat, rect=r, plane=plane, flipped=flipped)
        elif ptype == "box":
            if "p0" in el and "p1" in el:
                p0, p1 = vec3(el["p0"]), vec3(el["p1"])
            else:
                s = vec3(el["size"]) if "size" in el else (F(1), F(1), F(1))
                # box(w,h,d,mat) : box(vec3(-w/2,-h/2,-d/2), vec3(w/2,h/2,d/2), mat)  primitive.h:230
                p0 = tuple(F(-x / F(2)) for x in s)
                p1 = tuple(F(x / F(2)) for x in s)
            p = Prim(PRIM_BOX, mat, p0=p0, p1=p1)
        elif ptype == "volume":
            b = self.prims.get(el["primitive"], -1)
            if b < 0:
                raise ValueError("volume refers to an unknown boundary primitive")
            color = vec3(el["color"]) if "color" in el else MAUVE
            self.sp.materials.append(Material(MAT_ISOTROPIC, color, json_type="isotropic(phase)"))
            phase = len(self.sp.materials) - 1
            # the wrapped material (light test only) is the boundary primitive's (scene_parser.h:231)
            p = Prim(PRIM_VOLUME, self.sp.prims[b].mat, boundary=b, density=f32(el["density"]), phase_mat=phase)
        else:
            raise NotImplementedError(f"primitive type {ptype!r} is outside the hot-path scope")
        self.sp.prims.append(p)
        return len(self.sp.prims) - 1

    def build(self, scene) -> SceneParams:
        for el in scene.get("assets", []) or []:
            if el.get("skip", False):
                continue
            assert el["type"] == "object"
        self.parse_textures(scene)
        self.parse_materials(scene)
        for el in scene.get("primitives", []) or []:
            pid = el["id"] if "id" in el else self.new_id()
            idx = self.parse_prim(el)
            self.prims.setdefault(pid, idx)
        inst_prim: List[Optional[str]] = []
        for el in scene.get("instances", []) or []:
            if el["type"] == "ref":
                inst_prim.append(el["primitive"]["id"])
                continue
            pid = self.new_id()
            idx = self.parse_prim(el["primitive"])  # built even when the instance is skipped (:464-480)
            self.prims.setdefault(pid, idx)
            inst_prim.append(pid)
        for el, pid in zip(scene.get("instances", []) or [], inst_prim):
            if el.get("skip", False):
                continue
            inst = Instance(prim=self.prims[pid])
            if "transform" in el:
                t = el["transform"]
                if "scale" in t and isinstance(t["scale"], list):
                    inst.scale = vec3(t["scale"])
                else:
                    s = f32(t.get("scale", 1.0))
                    inst.scale = (s, s, s)
                if "rotate" in t:
                    inst.rotate = vec3(t["rotate"])
                if "translate" in t:
                    inst.translate = vec3(t["translate"])
            prim = self.sp.prims[inst.prim]
            inst.is_light = self.sp.materials[prim.mat].json_type == "diffuse_light"
            self.sp.instances.append(inst)
        w = scene.get("world")
        if w is not None:
            if "texture" in w:
                col, _, ti = self.texture_ref(w["texture"])
                self.sp.background, self.sp.background_texture = col, ti
            elif "color" in w:
                self.sp.background = vec3(w["color"])
            else:
                self.sp.background = MAUVE
        cam = scene["camera"]
        self.sp.camera = Camera(vec3(cam["look_from"]), vec3(cam["look_at"]), f32(cam.get("fov", 30.0)),
                                f32(cam.get("aperture", 0.0)), f32(cam.get("dist_to_focus", 10.0)))
        return self.sp


def load_scene_params(path_or_dict, base_dir: Optional[str] = None) -> SceneParams:
    """base_dir: directory that relative PNG paths are resolved against (the reference resolves them against its
    working directory); defaults to the parent of the scene file's directory, i.e. the repo root for scenes/*.json."""
    b = _Builder()
    if isinstance(path_or_dict, (str, bytes)):
        with open(path_or_dict) as f:
            scene = json.load(f)
        b.base_dir = base_dir or os.path.dirname(os.path.dirname(os.path.abspath(path_or_dict)))
    else:
        scene = path_or_dict
        b.base_dir = base_dir or "."
    return b.build(scene)


def decode_png(path: str):
    """What lodepng::decode(image, w, h, path) hands to from_4byte_vector (scene_parser.h:39-55): RGBA8, row 0 first.
    lodepng 's own code is not vendored in the reference; this is a plain PNG reader (zlib from the Python stdlib) for
    non-interlaced 8-bit grey / grey+alpha / RGB / RGBA / palette images -- the test-side twin of the product's reader."""
    import struct
    import zlib
    raw = open(path, "rb").read()
    if raw[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG: " + path)
    pos, idat, plte, trns, hdr = 8, b"", None, None, None
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = body
        elif typ == b"tRNS":
            trns = body
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 3, 4, 6):
        raise NotImplementedError("PNG: only non-interlaced 8-bit images are supported")
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    data = zlib.decompress(idat)
    stride = w * ch
    rows = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    p = 0
    for y in range(h):
        ft = data[p]
        line = np.frombuffer(data, np.uint8, stride, p + 1).astype(np.int32)
        p += 1 + stride
        cur = np.zeros(stride, np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            for i in range(stride):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if ft == 1:
                    pr = a
                elif ft == 3:
                    pr = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pr) & 255
        rows[y] = cur
        prev = cur
    px = rows.reshape(h, w, ch)
    out = np.zeros((h, w, 4), np.uint8)
    out[..., 3] = 255
    if ctype == 0:
        out[..., 0] = out[..., 1] = out[..., 2] = px[..., 0]
        if trns is not None and len(trns) >= 2:
            out[..., 3] = np.where(px[..., 0] == trns[1], 0, 255)
    elif ctype == 4:
        out[..., 0] = out[..., 1] = out[..., 2] = px[..., 0]
        out[..., 3] = px[..., 1]
    elif ctype == 2:
        out[..., :3] = px
        if trns is not None and len(trns) >= 6:
            key = np.array([trns[1], trns[3], trns[5]], np.uint8)
            out[..., 3] = np.where((px == key).all(axis=2), 0, 255)
    elif ctype == 6:
        out[...] = px
    else:
        pal = np.frombuffer(plte, np.uint8).reshape(-1, 3)
        al = np.full(len(pal), 255, np.uint8)
        if trns is not None:
            al[:len(trns)] = np.frombuffer(trns, np.uint8)[:len(pal)]
        out[..., :3] = pal[px[..., 0]]
        out[..., 3] = al[px[..., 0]]
    return w, h, out.tobytes()


def _hx(x) -> str:
    return float(x).hex()


def to_text(sp: SceneParams) -> str:
    """Line-based dump read by oracle/ref_driver.cpp (hex floats: exact float32 values)."""
    out = []
    c = sp.camera
    out.append("camera " + " ".join(_hx(v) for v in (*c.look_from, *c.look_at, c.fov, c.aperture, c.dist_to_focus)))
    out.append("background " + " ".join(_hx(v) for v in sp.background))
    for t in sp.textures:   # before the materials that refer to them; children before their checker
        if t.type == TEX_CONSTANT:
            out.append("texture constant %s %s" % (" ".join(_hx(v) for v in t.color), _hx(t.alpha)))
        elif t.type == TEX_CHECKER:
            out.append("texture checker %d %d %s" % (t.even, t.odd, _hx(t.scale)))
        elif t.type == TEX_PERLIN:
            out.append("texture perlin %s" % _hx(t.scale))
        else:
            out.append("texture image %d %d %s" % (t.width, t.height, t.rgba.hex()))
    if sp.background_texture >= 0:
        out.append("background_texture %d" % sp.background_texture)
    for m in sp.materials:
        out.append("material %d %s %s %s %d %s %s %d" % (m.type, " ".join(_hx(v) for v in m.color), _hx(m.alpha),
                                                        _hx(m.power), int(m.two_sided), _hx(m.fuzz), _hx(m.ior), m.texture))
    for p in sp.prims:
        if p.type == PRIM_RECT:
            out.append("prim rect %d %s %d %d" % (p.mat, " ".join(_hx(v) for v in p.rect), p.plane, int(p.flipped)))
        elif p.type == PRIM_BOX:
            out.append("prim box %d %s" % (p.mat, " ".join(_hx(v) for v in (*p.p0, *p.p1))))
        elif p.type == PRIM_SPHERE:
            out.append("prim sphere %d %s" % (p.mat, " ".join(_hx(v) for v in (*p.center, p.radius))))
        elif p.type == PRIM_VOLUME:
            out.append("prim volume %d %d %s %d" % (p.mat, p.boundary, _hx(p.density), p.phase_mat))
    for i in sp.instances:
        out.append("instance %d %s %d" % (i.prim, " ".join(_hx(v) for v in (*i.scale, *i.rotate, *i.translate)),
                                          int(i.is_light)))
    return "\n".join(out) + "\n"
