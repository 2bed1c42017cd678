"""pathtrace_amd -- MI355X-native implementation of gillett-hernandez/pathtrace's per-pixel NEE
path-tracing hot path.

The product is `lib/libpathtrace_hip.so` (hand-written HIP kernels for gfx950 + a C++ host front end behind
the C ABI of include/pathtrace_hip.h).  This module is only the thin ctypes binding the tests and bench.py
use; it contains no rendering code and no CPU fallback: every render call goes through the C ABI into the
HIP kernels, and a missing library or device is an error.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PATHTRACE_HIP_LIB") or os.path.join(HERE, "lib", "libpathtrace_hip.so")

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = range(5)
PRIM_RECT, PRIM_BOX, PRIM_SPHERE, PRIM_VOLUME = range(4)
KERNELS = ("generate", "extend", "shade", "connect", "accumulate")


class PathtraceError(RuntimeError):
    pass


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("alpha", C.c_float), ("power", C.c_float),
                ("two_sided", C.c_int32), ("fuzz", C.c_float), ("ior", C.c_float), ("texture", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("alpha", C.c_float), ("even", C.c_int32),
                ("odd", C.c_int32), ("scale", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
                ("texel_offset", C.c_int64)]


class Primitive(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("rect", C.c_float * 5), ("plane", C.c_int32),
                ("flipped", C.c_int32), ("p0", C.c_float * 3), ("p1", C.c_float * 3), ("center", C.c_float * 3),
                ("radius", C.c_float), ("boundary", C.c_int32), ("density", C.c_float), ("phase_material", C.c_int32)]


class Instance(C.Structure):
    _fields_ = [("primitive", C.c_int32), ("fwd", C.c_float * 12), ("inv", C.c_float * 12), ("bbox", C.c_float * 6)]


class BvhNode(C.Structure):
    _fields_ = [("bbox", C.c_float * 6), ("left", C.c_int32), ("right", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [(n, C.c_float * 3) for n in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")] + \
               [("lens_radius", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("n_materials", C.c_int32), ("materials", C.POINTER(Material)),
                ("n_primitives", C.c_int32), ("primitives", C.POINTER(Primitive)),
                ("n_instances", C.c_int32), ("instances", C.POINTER(Instance)),
                ("n_nodes", C.c_int32), ("nodes", C.POINTER(BvhNode)),
                ("n_lights", C.c_int32), ("lights", C.POINTER(C.c_int32)),
                ("camera", Camera), ("background", C.c_float * 3),
                ("n_textures", C.c_int32), ("textures", C.POINTER(Texture)),
                ("texel_bytes", C.c_int64), ("texels", C.POINTER(C.c_uint8)),
                ("background_texture", C.c_int32),
                ("perlin_ranvec", C.POINTER(C.c_float)), ("perlin_perm", C.POINTER(C.c_int32))]


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_bounces", C.c_int32), ("light_samples", C.c_int32),
                ("russian_roulette", C.c_int32), ("only_direct_illumination", C.c_int32), ("normal_offset", C.c_float),
                ("seed", C.c_uint32), ("device", C.c_int32), ("max_paths_in_flight", C.c_int64)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("camera_samples", "rays", "extension_rays", "extension_hits", "shadow_rays",
                                          "term_miss", "term_rr", "term_emitter", "term_pdf", "term_bounce_limit",
                                          "rays_traced", "shadow_rays_traced")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Plan(C.Structure):
    """pt_plan (ABI v6): how the library cut the last render call into wavefront batches, and the state of the streams."""
    _fields_ = [("pixels", C.c_int64), ("samples", C.c_int32), ("spp_per_batch", C.c_int32), ("batches", C.c_int32), ("lanes", C.c_int32),
                ("paths_per_batch", C.c_int64), ("path_slots", C.c_int64), ("stream_bytes", C.c_int64), ("hbm_free_bytes", C.c_int64),
                ("auto_sized", C.c_int32), ("grown", C.c_int32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class KernelTimes(C.Structure):
    _fields_ = [("launches", C.c_uint64 * 5), ("ms", C.c_double * 5), ("units", C.c_uint64 * 5)]


class HostConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("exposure", C.c_float), ("gamma", C.c_float),
                ("ppm_output_path", C.c_char * 512), ("png_output_path", C.c_char * 512),
                ("traced_paths_output_path", C.c_char * 512), ("traced_paths_2d_output_path", C.c_char * 512),
                ("scene_path", C.c_char * 512), ("should_trace_paths", C.c_int32), ("avg_number_of_paths", C.c_float),
                ("block_width", C.c_int32), ("block_height", C.c_int32), ("trace_probability", C.c_float),
                ("render_type", C.c_int32), ("only_direct_illumination", C.c_int32), ("integrator_type", C.c_int32),
                ("max_bounces", C.c_int32), ("samples", C.c_int32), ("light_samples", C.c_int32), ("threads", C.c_uint32),
                ("normal_offset", C.c_float), ("russian_roulette", C.c_int32)]


# every symbol include/pathtrace_hip.h declares
EXPORTS = ["pt_create", "pt_destroy", "pt_render_async", "pt_render_tiles_async", "pt_reserve", "pt_prime", "pt_get_plan", "pt_plan_batches", "pt_render_seconds", "pt_wait_for",
           "pt_multi_reserve", "pt_multi_render_seconds", "pt_multi_wait_for", "pt_poll", "pt_wait", "pt_read_framebuffer", "pt_snapshot_framebuffer",
           "pt_clear_framebuffer", "pt_get_counters", "pt_device_framebuffer", "pt_set_device_framebuffer",
           "pt_get_stream", "pt_set_stream", "pt_set_profiling", "pt_get_kernel_times", "pt_set_lanes", "pt_measure_tile_costs", "pt_spec_header", "pt_spec_status", "pt_spec_wait", "pt_spec_info", "pt_spec_build_check", "pt_spec_build_info",
           "pt_read_last_batch_radiance", "pt_trace_rays", "pt_last_error", "pt_abi_version", "pt_device_count",
           "pt_multi_create", "pt_multi_destroy", "pt_multi_render_async", "pt_multi_poll", "pt_multi_wait",
           "pt_multi_read_framebuffer", "pt_multi_snapshot_framebuffer", "pt_multi_get_counters", "pt_multi_clear",
           "pt_multi_device_count", "pt_multi_tile_owners", "pt_multi_get_device_counters", "pt_multi_exchange_bytes",
           "pth_config_from_file", "pth_config_from_json", "pth_scene_from_file", "pth_scene_from_json",
           "pth_scene_desc", "pth_scene_free", "pth_spiral_tiles", "pth_write_ppm", "pth_main"]

_lib = None


def lib():
    """Load libpathtrace_hip.so.  Raises if it has not been built (python -m pathtrace_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PathtraceError(f"{LIB_PATH} is missing: build it with `python -m pathtrace_amd.build` "
                             "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, fp = C.c_void_p, C.POINTER(C.c_float)
    L.pt_create.restype = vp
    L.pt_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(Config)]
    L.pt_destroy.argtypes = [vp]
    L.pt_destroy.restype = None
    L.pt_render_async.argtypes = [vp] + [C.c_int32] * 6
    L.pt_render_tiles_async.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32]
    L.pt_reserve.argtypes = [vp, C.c_int64, C.c_int32]
    L.pt_get_plan.argtypes = [vp, C.POINTER(Plan)]
    L.pt_plan_batches.argtypes = [C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_int32)]
    L.pt_plan_batches.restype = C.c_int32
    L.pt_render_seconds.argtypes = [vp]
    L.pt_render_seconds.restype = C.c_double
    L.pt_wait_for.argtypes = [vp, C.c_int32]
    L.pt_multi_reserve.argtypes = [vp, C.c_int32, C.c_int32]
    L.pt_prime.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    L.pt_multi_render_seconds.argtypes = [vp]
    L.pt_multi_render_seconds.restype = C.c_double
    L.pt_multi_wait_for.argtypes = [vp, C.c_int32]
    L.pt_poll.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pt_wait.argtypes = [vp]
    L.pt_read_framebuffer.argtypes = [vp, fp]
    L.pt_snapshot_framebuffer.argtypes = [vp, fp, C.POINTER(C.c_uint64)]
    L.pt_clear_framebuffer.argtypes = [vp]
    L.pt_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.pt_device_framebuffer.argtypes = [vp]
    L.pt_device_framebuffer.restype = vp
    L.pt_set_device_framebuffer.argtypes = [vp, vp, C.c_size_t]
    L.pt_get_stream.argtypes = [vp]
    L.pt_get_stream.restype = vp
    L.pt_set_stream.argtypes = [vp, vp]
    L.pt_set_profiling.argtypes = [vp, C.c_int]
    L.pt_set_lanes.argtypes = [vp, C.c_int32]
    L.pt_measure_tile_costs.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_uint64)]
    L.pt_spec_header.argtypes = [C.POINTER(SceneDesc), C.c_char_p, C.c_size_t]
    L.pt_spec_status.argtypes = [vp]
    L.pt_spec_wait.argtypes = [vp]
    L.pt_spec_info.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.pt_spec_build_check.argtypes = [C.POINTER(SceneDesc), C.c_int32]
    L.pt_spec_build_check.restype = C.c_long
    L.pt_spec_build_info.argtypes = [C.POINTER(SceneDesc), C.c_int32, C.c_char_p, C.c_size_t]
    L.pt_multi_get_device_counters.argtypes = [vp, C.c_int32, C.POINTER(Counters)]
    L.pt_multi_exchange_bytes.argtypes = [vp]
    L.pt_multi_exchange_bytes.restype = C.c_uint64
    L.pt_get_kernel_times.argtypes = [vp, C.POINTER(KernelTimes)]
    L.pt_read_last_batch_radiance.argtypes = [vp, fp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.pt_trace_rays.argtypes = [vp, C.c_int64, C.c_int32, fp, fp, C.c_uint32, C.c_uint32, C.c_uint32, fp, C.POINTER(C.c_int32)]
    L.pt_multi_create.restype = vp
    L.pt_multi_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(Config), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32]
    L.pt_multi_destroy.argtypes = [vp]
    L.pt_multi_destroy.restype = None
    L.pt_multi_render_async.argtypes = [vp, C.c_int32, C.c_int32]
    L.pt_multi_poll.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pt_multi_wait.argtypes = [vp]
    L.pt_multi_read_framebuffer.argtypes = [vp, fp]
    L.pt_multi_snapshot_framebuffer.argtypes = [vp, fp, C.POINTER(C.c_uint64)]
    L.pt_multi_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.pt_multi_clear.argtypes = [vp]
    L.pt_multi_device_count.argtypes = [vp]
    L.pt_multi_tile_owners.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32]
    L.pt_last_error.restype = C.c_char_p
    L.pt_abi_version.restype = C.c_int
    L.pt_device_count.restype = C.c_int
    L.pth_config_from_file.argtypes = [C.c_char_p, C.POINTER(HostConfig)]
    L.pth_config_from_json.argtypes = [C.c_char_p, C.POINTER(HostConfig)]
    L.pth_scene_from_file.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
    L.pth_scene_from_file.restype = vp
    L.pth_scene_from_json.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
    L.pth_scene_from_json.restype = vp
    L.pth_scene_desc.argtypes = [vp]
    L.pth_scene_desc.restype = C.POINTER(SceneDesc)
    L.pth_scene_free.argtypes = [vp]
    L.pth_scene_free.restype = None
    L.pth_spiral_tiles.argtypes = [C.c_int32] * 4 + [C.POINTER(C.c_int32), C.c_int32]
    L.pth_write_ppm.argtypes = [C.c_char_p, fp, C.c_int32, C.c_int32, C.c_int32, C.c_float]
    L.pth_main.argtypes = [C.c_char_p]
    _lib = L
    return L


def last_error() -> str:
    return (lib().pt_last_error() or b"").decode(errors="replace")


def _check(rc, what):
    if rc < 0:
        raise PathtraceError(f"{what}: {last_error()}")
    return rc


class Scene:
    """Flat scene built by the C++ host front end from a reference-format scene JSON."""

    def __init__(self, path: str = None, width: int = 0, height: int = 0, text: str = None):
        if text is not None:
            self._h = lib().pth_scene_from_json(text.encode(), width, height)
        else:
            self._h = lib().pth_scene_from_file(os.fsencode(path), width, height)
        if not self._h:
            raise PathtraceError(f"scene load failed: {last_error()}")
        self.width, self.height = width, height

    @property
    def desc(self) -> SceneDesc:
        return lib().pth_scene_desc(self._h).contents

    def close(self):
        if getattr(self, "_h", None):
            lib().pth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # numpy views for tests
    def instance_tables(self):
        d = self.desc
        n = d.n_instances
        fwd = np.array([list(d.instances[i].fwd) for i in range(n)], np.float32)
        inv = np.array([list(d.instances[i].inv) for i in range(n)], np.float32)
        bbox = np.array([list(d.instances[i].bbox) for i in range(n)], np.float32)
        return fwd, inv, bbox

    def nodes(self):
        d = self.desc
        return [(np.array(list(d.nodes[i].bbox), np.float32), d.nodes[i].left, d.nodes[i].right) for i in range(d.n_nodes)]

    def lights(self):
        d = self.desc
        return [d.lights[i] for i in range(d.n_lights)]

    def textures(self):
        d = self.desc
        return [d.textures[i] for i in range(d.n_textures)]

    def perlin_tables(self):
        d = self.desc
        return (np.array([d.perlin_ranvec[i] for i in range(768)], np.float32).reshape(256, 3),
                np.array([d.perlin_perm[i] for i in range(768)], np.int32).reshape(3, 256))

    def texel_bytes(self) -> bytes:
        d = self.desc
        return bytes(bytearray(d.texels[i] for i in range(d.texel_bytes)))

    def camera(self):
        c = self.desc.camera
        out = []
        for n in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w"):
            out += list(getattr(c, n))
        return np.array(out + [c.lens_radius], np.float32)


def spec_header(scene: "Scene") -> str:
    """The scene's traversal program as the header of the per-scene sweep build (host only)."""
    n = _check(lib().pt_spec_header(C.byref(scene.desc), None, 0), "pt_spec_header")
    buf = C.create_string_buffer(n + 1)
    _check(lib().pt_spec_header(C.byref(scene.desc), buf, n + 1), "pt_spec_header")
    return buf.value.decode()


def spec_build_check(scene: "Scene", light_samples: int = 4) -> int:
    """Compile the scene's own traversal kernels for gfx950 (hiprtc, host only); returns the code object size."""
    n = lib().pt_spec_build_check(C.byref(scene.desc), light_samples)
    if n < 0:
        raise PathtraceError(f"pt_spec_build_check: {last_error()}")
    return n


def spec_build_info(scene: "Scene", light_samples: int = 4) -> dict:
    """The same build, answering with pt_spec_info's record: which compiler a context of this scene gets in this process."""
    import json
    n = lib().pt_spec_build_info(C.byref(scene.desc), light_samples, None, 0)
    if n < 0:
        raise PathtraceError(f"pt_spec_build_info: {last_error()}")
    buf = C.create_string_buffer(n + 1)
    lib().pt_spec_build_info(C.byref(scene.desc), light_samples, buf, n + 1)
    return json.loads(buf.value.decode(errors="replace"))


def load_config(path: str = None, text: str = None) -> HostConfig:
    hc = HostConfig()
    if text is not None:
        _check(lib().pth_config_from_json(text.encode(), C.byref(hc)), "pth_config_from_json")
    else:
        _check(lib().pth_config_from_file(os.fsencode(path), C.byref(hc)), "pth_config_from_file")
    return hc


def plan_batches(pixels: int, samples: int, path_slots: int, lanes: int = 3):
    """The library's launch plan rule (host only): (samples per batch, batches) of a call of pixels x samples."""
    nb = C.c_int32()
    spp = lib().pt_plan_batches(int(pixels), int(samples), int(path_slots), int(lanes), C.byref(nb))
    return int(spp), int(nb.value)


PLAN_MAX_PATHS = 1920 * 1080 * 96   # PT_PLAN_MAX_PATHS of include/pathtrace_hip.h


def spiral_tiles(width, height, bw, bh):
    n = _check(lib().pth_spiral_tiles(width, height, bw, bh, None, 0), "pth_spiral_tiles")
    buf = (C.c_int32 * (4 * n))()
    lib().pth_spiral_tiles(width, height, bw, bh, buf, n)
    return [tuple(buf[4 * i:4 * i + 4]) for i in range(n)]


class Renderer:
    """Device context (pt_ctx) for one scene + config on one GPU."""

    def __init__(self, scene: Scene, max_bounces=10, light_samples=4, russian_roulette=True, only_direct=False,
                 normal_offset=1e-4, seed=0, device=-1, max_paths_in_flight=0, width=None, height=None):
        self.scene = scene
        self.width = width or scene.width
        self.height = height or scene.height
        self.cfg = Config(self.width, self.height, max_bounces, light_samples, int(russian_roulette), int(only_direct),
                          np.float32(normal_offset), seed, device, max_paths_in_flight)
        self._h = lib().pt_create(C.byref(scene.desc), C.byref(self.cfg))
        if not self._h:
            raise PathtraceError(f"pt_create failed: {last_error()}")

    def close(self):
        if getattr(self, "_h", None):
            lib().pt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_async(self, spp_begin, spp_end, rect=None):
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, self.width, self.height)
        _check(lib().pt_render_async(self._h, x0, y0, x1, y1, spp_begin, spp_end), "pt_render_async")

    def render_tiles_async(self, rects, spp_begin, spp_end):
        flat = [int(v) for r in rects for v in r]
        arr = (C.c_int32 * len(flat))(*flat)
        _check(lib().pt_render_tiles_async(self._h, len(rects), arr, spp_begin, spp_end), "pt_render_tiles_async")

    def reserve(self, pixels: int, samples: int):
        """Size the wavefront streams for render calls of pixels x samples by the library's launch plan (pt_reserve):
        set-up, like the scene upload; the first render call does it otherwise."""
        _check(lib().pt_reserve(self._h, int(pixels), int(samples)), "pt_reserve")

    def prime(self, min_ms: int = 50, rects=None):
        """Warm-up before a timed render (pt_prime): one-sample passes on every lane for min_ms, then everything is cleared."""
        flat = [int(v) for r in (rects or []) for v in r]
        arr = (C.c_int32 * len(flat))(*flat) if flat else None
        _check(lib().pt_prime(self._h, len(flat) // 4, arr, int(min_ms)), "pt_prime")

    def plan(self) -> dict:
        """The library's launch plan of the last render call / reserve (pt_get_plan)."""
        p = Plan()
        _check(lib().pt_get_plan(self._h, C.byref(p)), "pt_get_plan")
        return p.as_dict()

    def render_seconds(self) -> float:
        """Wall seconds of the last render call, entry to the device finishing its last batch; < 0 while it runs."""
        return float(lib().pt_render_seconds(self._h))

    def wait_for(self, timeout_ms: int) -> bool:
        return bool(_check(lib().pt_wait_for(self._h, int(timeout_ms)), "pt_wait_for"))

    def poll(self):
        s, r = C.c_uint64(), C.c_uint64()
        done = _check(lib().pt_poll(self._h, C.byref(s), C.byref(r)), "pt_poll")
        return bool(done), s.value, r.value

    def wait(self):
        _check(lib().pt_wait(self._h), "pt_wait")

    def framebuffer(self) -> np.ndarray:
        fb = np.zeros((self.height, self.width, 3), np.float32)
        _check(lib().pt_read_framebuffer(self._h, fb.ctypes.data_as(C.POINTER(C.c_float))), "pt_read_framebuffer")
        return fb

    def snapshot(self):
        """(framebuffer SUM as it stands, camera samples known to be accumulated) without waiting for queued work."""
        fb = np.zeros((self.height, self.width, 3), np.float32)
        n = C.c_uint64()
        _check(lib().pt_snapshot_framebuffer(self._h, fb.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)), "pt_snapshot_framebuffer")
        return fb, n.value

    def clear(self):
        _check(lib().pt_clear_framebuffer(self._h), "pt_clear_framebuffer")

    def counters(self) -> dict:
        c = Counters()
        _check(lib().pt_get_counters(self._h, C.byref(c)), "pt_get_counters")
        return c.as_dict()

    def render(self, samples, rect=None) -> np.ndarray:
        self.render_async(0, samples, rect)
        return self.framebuffer()

    def set_lanes(self, n: int) -> int:
        """Batches rotate over n stream lanes from now on (1 = kernels of consecutive batches run one after the other);
        returns the number of lanes the context owns."""
        return _check(lib().pt_set_lanes(self._h, n), "pt_set_lanes")

    def measure_tile_costs(self, rects, spp: int = 1):
        """World::hit queries of `spp` samples per pixel of every rect, all rects in one pass (the tile planner)."""
        flat = [int(v) for r in rects for v in r]
        arr = (C.c_int32 * len(flat))(*flat)
        out = (C.c_uint64 * len(rects))()
        _check(lib().pt_measure_tile_costs(self._h, len(rects), arr, spp, out), "pt_measure_tile_costs")
        return [int(v) for v in out]

    def spec_status(self) -> int:
        """Per-scene build of the sweep: 1 in use, 0 still building, -1 not available (the generic kernels run)."""
        return lib().pt_spec_status(self._h)

    def spec_wait(self) -> int:
        """Block until the per-scene build has ended; 1 = the context launches the scene's own kernels, -1 = generic."""
        return lib().pt_spec_wait(self._h)

    def spec_info(self) -> dict:
        """Who compiled the module this context launches (pt_spec_info): built_by helper / in-process, the libhiprtc file,
        the code object's producer string and whether that is the compiler the library was built with."""
        import json
        n = lib().pt_spec_info(self._h, None, 0)
        if n < 0:
            raise PathtraceError(f"pt_spec_info: {last_error()}")
        buf = C.create_string_buffer(n + 1)
        lib().pt_spec_info(self._h, buf, n + 1)
        return json.loads(buf.value.decode(errors="replace"))

    def set_profiling(self, on: bool):
        _check(lib().pt_set_profiling(self._h, int(on)), "pt_set_profiling")

    def kernel_times(self) -> dict:
        kt = KernelTimes()
        _check(lib().pt_get_kernel_times(self._h, C.byref(kt)), "pt_get_kernel_times")
        return {k: {"launches": int(kt.launches[i]), "ms": float(kt.ms[i]), "units": int(kt.units[i])}
                for i, k in enumerate(KERNELS)}

    def last_batch_radiance(self, n) -> np.ndarray:
        out = np.zeros((n, 4), np.float32)
        got = C.c_size_t()
        _check(lib().pt_read_last_batch_radiance(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), n, C.byref(got)),
               "pt_read_last_batch_radiance")
        return out[:got.value]

    def trace_rays(self, origins, dirs, k0=0, k1=0, vol_dim=8):
        """World::hit for explicit rays (validation hook).  dirs.shape = (n, 3) or (n, 4, 3)."""
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        n = o.shape[0]
        nr = 1 if d.ndim == 2 else d.shape[1]
        t = np.zeros(n * nr, np.float32)
        ids = np.zeros(n * nr, np.int32)
        fpp = C.POINTER(C.c_float)
        _check(lib().pt_trace_rays(self._h, n, nr, o.ctypes.data_as(fpp), d.ctypes.data_as(fpp), k0, k1, vol_dim,
                                   t.ctypes.data_as(fpp), ids.ctypes.data_as(C.POINTER(C.c_int32))), "pt_trace_rays")
        return (t, ids) if nr == 1 else (t.reshape(n, nr), ids.reshape(n, nr))

    def device_framebuffer_ptr(self) -> int:
        return lib().pt_device_framebuffer(self._h)

    def set_device_framebuffer(self, ptr: int, nbytes: int):
        _check(lib().pt_set_device_framebuffer(self._h, ptr, nbytes), "pt_set_device_framebuffer")

    def stream_ptr(self) -> int:
        return lib().pt_get_stream(self._h)

    def set_stream(self, ptr):
        _check(lib().pt_set_stream(self._h, ptr), "pt_set_stream")


class MultiRenderer:
    """pt_multi: one context per listed device in this process, tiles cost-balanced over them, one sum at the end."""

    def __init__(self, scene: Scene, devices, max_bounces=10, light_samples=4, russian_roulette=True, only_direct=False,
                 normal_offset=1e-4, seed=0, max_paths_in_flight=0, block=128):
        self.scene = scene
        self.width, self.height = scene.width, scene.height
        self.cfg = Config(self.width, self.height, max_bounces, light_samples, int(russian_roulette), int(only_direct),
                          np.float32(normal_offset), seed, -1, max_paths_in_flight)
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        self._h = lib().pt_multi_create(C.byref(scene.desc), C.byref(self.cfg), len(devices), devs, block, block)
        if not self._h:
            raise PathtraceError(f"pt_multi_create failed: {last_error()}")

    def close(self):
        if getattr(self, "_h", None):
            lib().pt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_async(self, spp_begin, spp_end):
        _check(lib().pt_multi_render_async(self._h, spp_begin, spp_end), "pt_multi_render_async")

    def reserve(self, samples: int, prime_ms: int = 0):
        _check(lib().pt_multi_reserve(self._h, int(samples), int(prime_ms)), "pt_multi_reserve")

    def render_seconds(self) -> float:
        return float(lib().pt_multi_render_seconds(self._h))

    def wait_for(self, timeout_ms: int) -> bool:
        return bool(_check(lib().pt_multi_wait_for(self._h, int(timeout_ms)), "pt_multi_wait_for"))

    def wait(self):
        _check(lib().pt_multi_wait(self._h), "pt_multi_wait")

    def poll(self):
        s, r = C.c_uint64(), C.c_uint64()
        done = _check(lib().pt_multi_poll(self._h, C.byref(s), C.byref(r)), "pt_multi_poll")
        return bool(done), s.value, r.value

    def framebuffer(self) -> np.ndarray:
        fb = np.zeros((self.height, self.width, 3), np.float32)
        _check(lib().pt_multi_read_framebuffer(self._h, fb.ctypes.data_as(C.POINTER(C.c_float))), "pt_multi_read_framebuffer")
        return fb

    def snapshot(self):
        fb = np.zeros((self.height, self.width, 3), np.float32)
        n = C.c_uint64()
        _check(lib().pt_multi_snapshot_framebuffer(self._h, fb.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)), "pt_multi_snapshot_framebuffer")
        return fb, n.value

    def counters(self) -> dict:
        c = Counters()
        _check(lib().pt_multi_get_counters(self._h, C.byref(c)), "pt_multi_get_counters")
        return c.as_dict()

    def clear(self):
        _check(lib().pt_multi_clear(self._h), "pt_multi_clear")

    def device_counters(self, index: int) -> dict:
        c = Counters()
        _check(lib().pt_multi_get_device_counters(self._h, index, C.byref(c)), "pt_multi_get_device_counters")
        return c.as_dict()

    def exchange_bytes(self) -> int:
        """Device-to-device bytes moved by the last framebuffer sum."""
        return int(lib().pt_multi_exchange_bytes(self._h))

    def tile_owners(self):
        n = lib().pt_multi_tile_owners(self._h, None, 0)
        buf = (C.c_int32 * n)()
        lib().pt_multi_tile_owners(self._h, buf, n)
        return list(buf)


def write_ppm(path, fb_sum: np.ndarray, samples: int, exposure_field: float = 2.2):
    fb = np.ascontiguousarray(fb_sum, np.float32)
    h, w, _ = fb.shape
    _check(lib().pth_write_ppm(os.fsencode(path), fb.ctypes.data_as(C.POINTER(C.c_float)), w, h, samples,
                               C.c_float(exposure_field)), "pth_write_ppm")


def device_count() -> int:
    return lib().pt_device_count()
