"""TEST INFRASTRUCTURE -- ctypes binding of oracle/libpt_oracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import scene_params as sp

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpt_oracle.so")
REF_DRIVER = os.path.join(HERE, "_ref", "ref_driver")

MODE_MT, MODE_STREAM = 0, 1


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("alpha", C.c_float), ("power", C.c_float),
                ("two_sided", C.c_int32), ("fuzz", C.c_float), ("ior", C.c_float), ("texture", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("alpha", C.c_float), ("even", C.c_int32),
                ("odd", C.c_int32), ("scale", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
                ("texel_offset", C.c_int64)]


class Prim(C.Structure):
    _fields_ = [("type", C.c_int32), ("mat", C.c_int32), ("rect", C.c_float * 5), ("plane", C.c_int32),
                ("flipped", C.c_int32), ("p0", C.c_float * 3), ("p1", C.c_float * 3), ("center", C.c_float * 3),
                ("radius", C.c_float), ("boundary", C.c_int32), ("density", C.c_float), ("phase_mat", C.c_int32)]


class Instance(C.Structure):
    _fields_ = [("prim", C.c_int32), ("scale", C.c_float * 3), ("rotate", C.c_float * 3),
                ("translate", C.c_float * 3), ("is_light", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("look_from", C.c_float * 3), ("look_at", C.c_float * 3), ("fov", C.c_float),
                ("aperture", C.c_float), ("dist_to_focus", C.c_float)]


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32), ("max_bounces", C.c_int32),
                ("light_samples", C.c_int32), ("russian_roulette", C.c_int32), ("only_direct", C.c_int32),
                ("block_w", C.c_int32), ("block_h", C.c_int32), ("normal_offset", C.c_float)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "ext_rays", "ext_hits", "shadow_rays", "term_miss", "term_rr",
                                          "term_emitter", "term_pdf", "term_bounce_limit")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force: bool = False) -> None:
    """Compile the restatement (and oracle/_ref when the reference tree is present)."""
    subprocess.run(["make", "-C", HERE] + (["-B"] if force else []), check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.pto_scene_create.restype = C.c_void_p
        L.pto_scene_create.argtypes = [C.POINTER(Material), C.c_int, C.POINTER(Prim), C.c_int, C.POINTER(Instance),
                                       C.c_int, C.POINTER(Camera), C.POINTER(C.c_float)]
        L.pto_scene_create_textured.restype = C.c_void_p
        L.pto_scene_create_textured.argtypes = L.pto_scene_create.argtypes + [C.POINTER(Texture), C.c_int, C.c_char_p,
                                                                            C.c_int64, C.c_int]
        L.pto_perlin_tables.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.pto_texture_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_float),
                                       C.POINTER(C.c_float)]
        for name in ("ptm_sinf", "ptm_acosf"):
            getattr(L, name).argtypes = [C.c_float]
            getattr(L, name).restype = C.c_float
        L.ptm_atan2f.argtypes = [C.c_float, C.c_float]
        L.ptm_atan2f.restype = C.c_float
        L.pto_scene_destroy.argtypes = [C.c_void_p]
        for name in ("pto_scene_num_instances", "pto_scene_num_nodes", "pto_scene_num_lights"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        L.pto_scene_light.argtypes = [C.c_void_p, C.c_int]
        L.pto_scene_light.restype = C.c_int
        fp = C.POINTER(C.c_float)
        L.pto_scene_instance_tables.argtypes = [C.c_void_p, C.c_int, fp, fp, fp]
        L.pto_scene_node.argtypes = [C.c_void_p, C.c_int, fp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.pto_scene_camera.argtypes = [C.c_void_p, C.c_int, C.c_int, fp]
        L.pto_scene_next_random.argtypes = [C.c_void_p]
        L.pto_scene_next_random.restype = C.c_double
        L.pto_rng_after_static_init.argtypes = [C.c_int, C.POINTER(C.c_double)]
        L.pto_stream_u32.argtypes = [C.c_uint32] * 4
        L.pto_stream_u32.restype = C.c_uint32
        L.pto_stream_dims_per_bounce.argtypes = [C.c_void_p, C.c_int]
        L.pto_stream_dims_per_bounce.restype = C.c_int
        L.pto_render_mt.argtypes = [C.c_void_p, C.POINTER(Config), fp, C.POINTER(Counters)]
        L.pto_samples_mt.argtypes = [C.c_void_p, C.POINTER(Config), C.c_int, fp]
        L.pto_hits_mt.argtypes = [C.c_void_p, C.POINTER(Config), C.c_int, fp]
        L.pto_render_stream.argtypes = [C.c_void_p, C.POINTER(Config), C.c_uint32] + [C.c_int] * 7 + [fp, C.POINTER(Counters)]
        L.pto_sample_stream.argtypes = [C.c_void_p, C.POINTER(Config), C.c_uint32, C.c_int, C.c_int, C.c_int, fp,
                                        C.POINTER(Counters)]
        L.pto_world_hit_stream.argtypes = [C.c_void_p, C.c_int64, fp, fp, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.POINTER(C.c_int32), fp, C.POINTER(C.c_int32)]
        L.ptm_sincos_2pi.argtypes = [C.c_float, fp, fp]
        L.ptm_cbrtf.argtypes = [C.c_float]
        L.ptm_cbrtf.restype = C.c_float
        L.ptm_logf.argtypes = [C.c_float]
        L.ptm_logf.restype = C.c_float
        _lib = L
    return _lib


def make_config(width, height, samples, max_bounces=10, light_samples=4, russian_roulette=True, only_direct=False,
                block_w=128, block_h=128, normal_offset=1e-4) -> Config:
    return Config(width, height, samples, max_bounces, light_samples, int(russian_roulette), int(only_direct),
                  block_w, block_h, np.float32(normal_offset))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Scene:
    """Scene runtime of the restatement.  A fresh Scene = a fresh process in reference terms
    (mt19937 re-seeded, static-init and BVH draws consumed)."""

    def __init__(self, params: sp.SceneParams):
        self.params = params
        mats = (Material * len(params.materials))()
        for d, m in zip(mats, params.materials):
            d.type = m.type
            d.color[:] = [float(x) for x in m.color]
            d.alpha, d.power, d.two_sided, d.fuzz, d.ior = float(m.alpha), float(m.power), int(m.two_sided), float(m.fuzz), float(m.ior)
            d.texture = m.texture
        prims = (Prim * len(params.prims))()
        for d, p in zip(prims, params.prims):
            d.type, d.mat = p.type, p.mat
            d.rect[:] = [float(x) for x in p.rect]
            d.plane, d.flipped = p.plane, int(p.flipped)
            d.p0[:] = [float(x) for x in p.p0]
            d.p1[:] = [float(x) for x in p.p1]
            d.center[:] = [float(x) for x in p.center]
            d.radius, d.boundary, d.density, d.phase_mat = float(p.radius), p.boundary, float(p.density), p.phase_mat
        insts = (Instance * len(params.instances))()
        for d, i in zip(insts, params.instances):
            d.prim = i.prim
            d.scale[:] = [float(x) for x in i.scale]
            d.rotate[:] = [float(x) for x in i.rotate]
            d.translate[:] = [float(x) for x in i.translate]
            d.is_light = int(i.is_light)
        c = params.camera
        cam = Camera()
        cam.look_from[:] = [float(x) for x in c.look_from]
        cam.look_at[:] = [float(x) for x in c.look_at]
        cam.fov, cam.aperture, cam.dist_to_focus = float(c.fov), float(c.aperture), float(c.dist_to_focus)
        bg = (C.c_float * 3)(*[float(x) for x in params.background])
        texs = (Texture * max(len(params.textures), 1))()
        blob = b""
        for d, t in zip(texs, params.textures):
            d.type = t.type
            d.color[:] = [float(x) for x in t.color]
            d.alpha, d.even, d.odd, d.scale, d.width, d.height = float(t.alpha), t.even, t.odd, float(t.scale), t.width, t.height
            if t.type == sp.TEX_IMAGE:
                d.texel_offset = len(blob)
                blob += t.rgba
        self._h = lib().pto_scene_create_textured(mats, len(mats), prims, len(prims), insts, len(insts), C.byref(cam), bg,
                                                  texs, len(params.textures), blob, len(blob), params.background_texture)
        if not self._h:
            raise ValueError("pto_scene_create rejected the scene")

    @classmethod
    def from_json(cls, path):
        return cls(sp.load_scene_params(path))

    def close(self):
        if self._h:
            lib().pto_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- tables -------------------------------------------------------------------------
    def instance_tables(self):
        n = lib().pto_scene_num_instances(self._h)
        fwd = np.zeros((n, 12), np.float32)
        inv = np.zeros((n, 12), np.float32)
        bbox = np.zeros((n, 6), np.float32)
        for i in range(n):
            lib().pto_scene_instance_tables(self._h, i, _fp(fwd[i]), _fp(inv[i]), _fp(bbox[i]))
        return fwd, inv, bbox

    def nodes(self):
        n = lib().pto_scene_num_nodes(self._h)
        out = []
        for k in range(n):
            bb = np.zeros(6, np.float32)
            l, r = C.c_int32(), C.c_int32()
            lib().pto_scene_node(self._h, k, _fp(bb), C.byref(l), C.byref(r))
            out.append((bb, l.value, r.value))
        return out

    def lights(self):
        return [lib().pto_scene_light(self._h, k) for k in range(lib().pto_scene_num_lights(self._h))]

    def camera(self, width, height):
        out = np.zeros(22, np.float32)
        lib().pto_scene_camera(self._h, width, height, _fp(out))
        return out

    def texture_eval(self, ti, uvp, mode=MODE_MT):
        """value (rgb) and alpha of texture `ti` at rows (u, v, px, py, pz)."""
        q = np.ascontiguousarray(uvp, np.float32).reshape(-1, 5)
        out = np.zeros((len(q), 4), np.float32)
        for k in range(len(q)):
            p = (C.c_float * 3)(*q[k, 2:5])
            lib().pto_texture_eval(self._h, ti, mode, float(q[k, 0]), float(q[k, 1]), p, _fp(out[k]))
        return out

    def next_random(self) -> float:
        return lib().pto_scene_next_random(self._h)

    def dims_per_bounce(self, light_samples):
        return lib().pto_stream_dims_per_bounce(self._h, light_samples)

    # --- rendering ----------------------------------------------------------------------
    def render_mt(self, cfg: Config):
        fb = np.zeros((cfg.height, cfg.width, 3), np.float32)
        ctr = Counters()
        lib().pto_render_mt(self._h, C.byref(cfg), _fp(fb), C.byref(ctr))
        return fb, ctr.as_dict()

    def samples_mt(self, cfg: Config, n: int):
        out = np.zeros((n, 13), np.float32)
        lib().pto_samples_mt(self._h, C.byref(cfg), n, _fp(out))
        return out

    def hits_mt(self, cfg: Config, n: int):
        out = np.zeros((n, 9), np.float32)
        lib().pto_hits_mt(self._h, C.byref(cfg), n, _fp(out))
        return out

    def render_stream(self, cfg: Config, seed=0, rect=None, s0=0, s1=None, threads=None, fb=None):
        if rect is None:
            rect = (0, 0, cfg.width, cfg.height)
        if s1 is None:
            s1 = cfg.samples
        if threads is None:
            threads = os.cpu_count() or 1
        if fb is None:
            fb = np.zeros((cfg.height, cfg.width, 3), np.float32)
        ctr = Counters()
        lib().pto_render_stream(self._h, C.byref(cfg), seed, rect[0], rect[1], rect[2], rect[3], s0, s1, threads,
                                _fp(fb), C.byref(ctr))
        return fb, ctr.as_dict()

    def world_hit_stream(self, origins, dirs, k0=0, k1=0, vol_dim=8):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        n = o.shape[0]
        hit = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        inst = np.zeros(n, np.int32)
        ip = C.POINTER(C.c_int32)
        lib().pto_world_hit_stream(self._h, n, _fp(o), _fp(d), k0, k1, vol_dim, hit.ctypes.data_as(ip), _fp(t), inst.ctypes.data_as(ip))
        return hit, t, inst

    def sample_stream(self, cfg: Config, i, j, s, seed=0):
        rgb = np.zeros(3, np.float32)
        ctr = Counters()
        lib().pto_sample_stream(self._h, C.byref(cfg), seed, i, j, s, _fp(rgb), C.byref(ctr))
        return rgb, ctr.as_dict()


def perlin_tables():
    """(ranvec 256x3 float32, perm 3x256 int32) as the reference's static initialisers leave them."""
    rv = np.zeros((256, 3), np.float32)
    pm = np.zeros((3, 256), np.int32)
    lib().pto_perlin_tables(_fp(rv), pm.ctypes.data_as(C.POINTER(C.c_int32)))
    return rv, pm


def rng_after_static_init(n: int) -> np.ndarray:
    out = np.zeros(n, np.float64)
    lib().pto_rng_after_static_init(n, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def stream_u32(seed, pixel, sample, dim) -> int:
    return lib().pto_stream_u32(seed, pixel, sample, dim)


# ---- oracle/_ref (the real reference headers) ---------------------------------------------
def ref_available() -> bool:
    return os.path.exists(REF_DRIVER)


def _cfg_args(cfg: Config):
    return [str(cfg.width), str(cfg.height), str(cfg.samples), str(cfg.max_bounces), str(cfg.light_samples),
            str(cfg.russian_roulette), repr(float(cfg.normal_offset)), str(cfg.only_direct), str(cfg.block_w),
            str(cfg.block_h)]


def ref_run(params: sp.SceneParams, mode: str, args, workdir: str):
    """Run oracle/_ref/ref_driver (build container only: oracle/_ref/ is git-ignored and gpurun-ignored, nothing on the GPU box runs it)."""
    ppath = os.path.join(workdir, "scene.params")
    with open(ppath, "w") as f:
        f.write(sp.to_text(params))
    out = subprocess.run([REF_DRIVER, ppath, mode] + list(args), check=True, capture_output=True, cwd=workdir)
    return out.stdout.decode(errors="replace")


def ref_render(params, cfg: Config, workdir: str, ppm_path: str = None):
    out = os.path.join(workdir, "fb.f32")
    txt = ref_run(params, "render", _cfg_args(cfg) + [out] + ([ppm_path] if ppm_path else []), workdir)
    rays = int([l for l in txt.splitlines() if l.startswith("REF_RAYS")][0].split()[1])
    fb = np.fromfile(out, np.float32).reshape(cfg.height, cfg.width, 3)
    return fb, rays


def ref_samples(params, cfg: Config, n: int, workdir: str, mode="samples"):
    out = os.path.join(workdir, mode + ".f32")
    ref_run(params, mode, _cfg_args(cfg) + [str(n), out], workdir)
    return np.fromfile(out, np.float32).reshape(n, 13 if mode == "samples" else 9)
