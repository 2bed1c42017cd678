/* pathtrace_hip.h -- C ABI of libpathtrace_hip.so: the MI355X (gfx950) wavefront implementation
 * of the reference's per-pixel NEE path-tracing hot path.
 *
 * This is the drop-in boundary.  In the reference the path is entered one camera
 * sample at a time through  Integrator::color(ray&, int, long*, path*, bool)  (integrator.h:14,
 * implemented by NEEIterative::color integrator.h:176-339) from  Tiled::compute  (renderer.h:626-691).
 * One-ray-per-virtual-call is not a viable device boundary, so the boundary is the Renderer
 * (renderer.h:114-150): a `HipWavefront : Renderer` forwards start_render / sync_progress /
 * is_done / finalize to the entry points below (see INTEGRATION.md for the reference-side stub).
 *
 * Conventions: plain pointers and sizes, caller owns every host buffer, the library owns device
 * memory.  All functions return 0 on success and a negative value on failure unless noted;
 * pt_last_error() gives the message for the calling thread.  One pt_ctx per host thread / GPU.
 * There is NO CPU fallback: without a usable HIP device pt_create fails.
 */
#ifndef PATHTRACE_HIP_H
#define PATHTRACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 6

/* ---- environment variables the library reads (all of them; none is needed for normal use, none changes a pixel) --------------
 *   PATHTRACE_HIP_DEVICES       pth_main / HipWavefront: "0,1,.." or "all" -- the devices that render (default: config.json's
 *                               `threads`, capped by the devices present); an ordinal may repeat ("0,0": a one-GPU rehearsal)
 *   PATHTRACE_HIP_LANES         1..4 stream lanes per context (default 3); 1 = every kernel alone on the chip (profiling)
 *   PATHTRACE_HIP_PLAN          comma list: caller (every render call is one batch as far as the slots allow: probing other launch
 *                               plans), max=<paths> (cap of the library's own sizing, default PT_PLAN_MAX_PATHS), seg=<slots>
 *                               (queue segment size, default 4096, a multiple of 256)
 *   PATHTRACE_HIP_SPEC          per-scene build of the traversal kernels: async (default) | sync | off, optionally ",extend-only"
 *                               (shadow rays stay on the generic k_connect)
 *   PATHTRACE_HIP_SPEC_FLAGS    more compiler options for that build (-DPT_CONNECT_WAVES=4, -mllvm ...)
 *   PATHTRACE_HIP_SPEC_CC       its compile helper (default: pt_spec_cc beside the library); "in-process" = no helper
 *   PATHTRACE_HIP_SPEC_DUMP     file to write the module's code object to
 *   PATHTRACE_HIP_SPEC_BREAK    make the build fail on purpose (the fallback onto the generic kernels, tested)
 *   PATHTRACE_HIP_TRAVERSAL     comma list forcing a traversal form: general (no fast sweep), walk (per-lane walk also for small
 *                               scenes), sweep (never the walk), tree (no flat program) -- the tests hold the forms to each other
 *   PATHTRACE_HIP_SHADE         comma list: sort | nosort (k_shade's chunk sort on / off), nostage (every hit gets a shadow record)
 *   PATHTRACE_HIP_MULTI         comma list for pt_multi: roundrobin (tile k -> device k mod n), rccl (ncclReduce of whole frames)
 *   PATHTRACE_HIP_TRACE_LAUNCH  one stream synchronisation and one stderr line per launch (names the kernel in front of a fault)
 * Everything else that used to be a run-time knob (grid size, segment merging, rays per sweep, ...) is a -D of the build:
 * `python -m pathtrace_amd.build --variant NAME -DPT_...`, loaded by the Python binding through PATHTRACE_HIP_LIB. */

/* material.h:27-277 */
enum { PT_MAT_LAMBERTIAN = 0, PT_MAT_METAL = 1, PT_MAT_DIELECTRIC = 2, PT_MAT_DIFFUSE_LIGHT = 3, PT_MAT_ISOTROPIC = 4 };
/* primitive.h:27-256, volume.h:7-93 */
enum { PT_PRIM_RECT = 0, PT_PRIM_BOX = 1, PT_PRIM_SPHERE = 2, PT_PRIM_VOLUME = 3 };
/* primitive.h:11-16  enum plane_enum { XY, XZ, YZ } */
enum { PT_PLANE_XY = 0, PT_PLANE_XZ = 1, PT_PLANE_YZ = 2 };

/* ---- flat POD scene: what World / bvh_node / instance / material objects hold at render time ---- */

/* texture.h: constant_texture :14-31, checker_texture :33-75, noise_texture :185-196; image.h:7-50 image_texture.
 * A checker's children are indices of EARLIER entries of the table. */
enum { PT_TEX_CONSTANT = 0, PT_TEX_CHECKER = 1, PT_TEX_PERLIN = 2, PT_TEX_IMAGE = 3 };
typedef struct pt_texture {
    int32_t type;
    float color[3];            /* constant_texture::color */
    float alpha;               /* constant_texture::a */
    int32_t even, odd;         /* checker_texture::even / ::odd */
    float scale;               /* checker_texture::scale, noise_texture::scale */
    int32_t width, height;     /* image_texture */
    int64_t texel_offset;      /* image_texture: byte offset of its RGBA8 pixels (what lodepng::decode returns, row 0
                                  first; scene_parser.h:39-55) in pt_scene_desc::texels */
} pt_texture;

typedef struct pt_material {   /* material.h: lambertian / metal / diffuse_light / isotropic */
    int32_t type;
    float color[3];            /* albedo or emit colour (texture.h:14-31) */
    float alpha;               /* constant_texture::a */
    float power;               /* diffuse_light::power (material.h:243) */
    int32_t two_sided;         /* diffuse_light::two_sided */
    float fuzz, ior;           /* metal / dielectric: carried for completeness; neither influences NEEIterative's radiance
                                  (metal is cosine-diffuse material.h:99-108, a dielectric path ends after its NEE) */
    int32_t texture;           /* lambertian / isotropic albedo, diffuse_light emit: index into pt_scene_desc::textures,
                                  or -1 = the constant texture (color, alpha) above */
} pt_material;

typedef struct pt_primitive {
    int32_t type;
    int32_t material;          /* index into materials (rec.mat_ptr) */
    float rect[5];             /* rect: x0 z0 x1 z1 y in the XZ-canonical frame (primitive.h:120-124) */
    int32_t plane;             /* rect: plane_enum */
    int32_t flipped;           /* rect: ctor argument `flipped` (normal = !flipped) */
    float p0[3], p1[3];        /* box (primitive.h:229-242) */
    float center[3], radius;   /* sphere */
    int32_t boundary;          /* volume: index of the boundary primitive (volume.h:10) */
    float density;             /* volume */
    int32_t phase_material;    /* volume: index of its isotropic phase function material */
} pt_primitive;

typedef struct pt_instance {   /* primitive.h:258-348 */
    int32_t primitive;
    float fwd[12];             /* transform3::_transform, rows of the 3x4 affine (transform3.h:69) */
    float inv[12];             /* transform.inverse(), same layout (transform3.h:51-54) */
    float bbox[6];             /* instance::bbox min xyz, max xyz (primitive.h:272-296) */
} pt_instance;

typedef struct pt_bvh_node {   /* bvh.h:6-29; nodes in preorder, node 0 = root */
    float bbox[6];
    int32_t left, right;       /* >= 0: node index;  < 0: ~instance_index */
} pt_bvh_node;

typedef struct pt_camera {     /* camera.h:111-117 for one aspect ratio */
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    float lens_radius;
} pt_camera;

typedef struct pt_scene_desc {
    int32_t n_materials;  const pt_material *materials;
    int32_t n_primitives; const pt_primitive *primitives;
    int32_t n_instances;  const pt_instance *instances;   /* in scene-file order (= hit_record::primitive ids) */
    int32_t n_nodes;      const pt_bvh_node *nodes;
    int32_t n_lights;     const int32_t *lights;          /* World::lights as instance indices (world.h:39) */
    pt_camera camera;
    float background[3];                                  /* World::background, constant colour (world.h:27-30) */
    /* textures; all optional: n_textures = 0, background_texture = -1 */
    int32_t n_textures;   const pt_texture *textures;
    int64_t texel_bytes;  const uint8_t *texels;          /* RGBA8 pixels of the image textures */
    int32_t background_texture;                           /* World::background as a texture index, -1 = the colour above */
    const float *perlin_ranvec;                           /* perlin::ranvec, 256 x 3 (texture.h:176); required iff a perlin */
    const int32_t *perlin_perm;                           /* perm_x, perm_y, perm_z, 3 x 256 (texture.h:177-179)  texture exists */
} pt_scene_desc;

typedef struct pt_config {     /* the Config fields the path reads (config.h:74-96) */
    int32_t width, height;     /* film */
    int32_t max_bounces;       /* integrator.h:186 */
    int32_t light_samples;     /* integrator.h:221 */
    int32_t russian_roulette;  /* integrator.h:287 */
    int32_t only_direct_illumination; /* integrator.h:299 */
    float normal_offset;       /* integrator.h:274 */
    uint32_t seed;             /* stream RNG seed (the reference has one fixed mt19937 seed) */
    int32_t device;            /* HIP device ordinal, -1 = current device */
    int64_t max_paths_in_flight; /* path slots of one wavefront batch.  0 (what a plugin passes) = the library sizes the
                                  * context itself: the LAUNCH PLAN below, from the pixels and samples of the render calls
                                  * and the free HBM of the device (ABI v6; up to v5: a fixed 8 Mi) */
} pt_config;

/* ---- the launch plan (ABI v6) ------------------------------------------------------------------------------
 * Tiled::start_render (renderer.h:553-603) fans the whole job out at full speed from config.json alone; so does
 * pt_render_async.  A render call of `pixels` pixels x `samples` samples per pixel is cut into wavefront batches by
 * one rule, measured on MI355X: as FEW EQUAL batches as the path slots allow, at least TWO (a single
 * batch's thin late bounces have nothing to overlap with), and a multiple of the context's stream lanes when there are
 * more batches than lanes (the last round of batches is then a full one).  With max_paths_in_flight = 0 the slots are
 * the library's to choose: pixels x samples-per-batch of that plan, capped at PT_PLAN_MAX_PATHS and at
 * PT_PLAN_HBM_FRACTION of the device memory free when the streams are allocated ((192 + 24 light_samples) bytes per slot
 * and lane).  The streams are allocated by the first render call (or by pt_reserve, which a plugin calls from its
 * constructor so that the allocation stays out of the timed render) and GROW when a later call's plan wants more; they
 * never shrink.  With max_paths_in_flight > 0 the caller's size stands and the same rule cuts the calls within it. */
#define PT_PLAN_MAX_PATHS 199065600ll      /* 1920 x 1080 x 96: 48 601 segments of 4096 slots; 172 GB at 3 lanes */
#define PT_PLAN_HBM_FRACTION 0.70
typedef struct pt_plan {
    int64_t pixels;            /* of the render call (or pt_reserve) the plan was last made for */
    int32_t samples;           /* samples per pixel of that call */
    int32_t spp_per_batch;     /* samples per pixel of one wavefront batch */
    int32_t batches;           /* batches the call was cut into */
    int32_t lanes;             /* stream lanes the batches rotate over */
    int64_t paths_per_batch;   /* pixels x spp_per_batch */
    int64_t path_slots;        /* slots every lane's streams hold now */
    int64_t stream_bytes;      /* device memory of the wavefront streams, all lanes */
    int64_t hbm_free_bytes;    /* hipMemGetInfo's free bytes when the streams were last sized */
    int32_t auto_sized;        /* 1: max_paths_in_flight = 0, the library chose path_slots */
    int32_t grown;             /* times the streams were re-allocated larger */
} pt_plan;

/* reference counters: rays = World::hit queries (integrator.h:192,247; renderer.h:696-706) */
typedef struct pt_counters {
    uint64_t camera_samples;
    uint64_t rays, extension_rays, extension_hits, shadow_rays;
    uint64_t term_miss, term_rr, term_emitter, term_pdf, term_bounce_limit;
    /* `rays` and `shadow_rays` count what the reference counts: light_samples shadow rays per hit.  A hit none of whose
     * light samples can contribute (every coefficient +-0 or NaN: the surface faces away from the light, the hit lies on
     * the light) is given no shadow record and its shadow rays are NOT traced -- the image is the same bit for bit.  These
     * two leave such rays out: World::hit queries the device actually performed. */
    uint64_t rays_traced, shadow_rays_traced;
} pt_counters;

/* per-kernel device time of the last completed pt_render_async, from HIP events on the render
 * stream (only collected when pt_set_profiling(ctx, 1)); launches and milliseconds per kernel */
#define PT_N_KERNELS 5
enum { PT_K_GENERATE = 0, PT_K_EXTEND = 1, PT_K_SHADE = 2, PT_K_CONNECT = 3, PT_K_ACCUMULATE = 4 };
typedef struct pt_kernel_times {
    uint64_t launches[PT_N_KERNELS];
    double ms[PT_N_KERNELS];
    uint64_t units[PT_N_KERNELS];   /* rays (extend, connect) or path records (others) processed */
} pt_kernel_times;

typedef struct pt_ctx pt_ctx;

/* Copies the flat scene to the device and allocates the wavefront streams.
 * Replaces: Renderer::Renderer + Tiled::Tiled (renderer.h:121-133, 545-551) as far as the device is concerned.
 * Returns NULL on failure (no device, unsupported material/primitive, bad indices). */
pt_ctx *pt_create(const pt_scene_desc *scene, const pt_config *config);
void pt_destroy(pt_ctx *ctx);

/* Enqueue the camera samples  [spp_begin, spp_end) x pixels [x0,x1) x [y0,y1)  (j = 0 is the BOTTOM row,
 * renderer.h:30) and return immediately.  Replaces the body of Tiled::compute (renderer.h:626-691) for one
 * tile: jitter + camera::get_ray + NEEIterative::color + framebuffer[j][i] += de_nan(col). */
int pt_render_async(pt_ctx *ctx, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t spp_begin, int32_t spp_end);
/* Same for a set of pixel rects (image tiles, NaiveSpiral::next queue.h:121-127) rendered together as one
 * wavefront batch: rects = n_rects * {x0, y0, x1, y1}.  This is how one GPU renders the tiles it owns in a
 * tile-partitioned multi-GPU render without paying one small batch per tile. */
int pt_render_tiles_async(pt_ctx *ctx, int32_t n_rects, const int32_t *rects, int32_t spp_begin, int32_t spp_end);
/* ABI v6.  Size (or grow) the wavefront streams for render calls of `pixels` pixels x `samples` samples per pixel, by the
 * launch plan above; what Renderer::Renderer's framebuffer allocation (renderer.h:121-133) is to the reference.  Optional:
 * the first render call does it otherwise, inside the caller's timed region.  Returns 0, < 0 on failure. */
int pt_reserve(pt_ctx *ctx, int64_t pixels, int32_t samples);
/* ABI v6.  Warm the context up before a timed render: one-sample passes over the given rects (n_rects = 0: the whole film) on
 * every lane for at least `min_ms` milliseconds, then framebuffer and counters are cleared (call it before the first render,
 * after pt_reserve).  A fresh process's first render otherwise pays the code objects' first launches, the first touch of the
 * freshly allocated streams and the clock ramp of an idle GPU: +5 ms on a 104 ms render of 1080p x 256 spp, measured
 * (profiles/experiments/r05_cold_start.jsonl); Tiled::start_render's thread spawn is outside its timed region as well
 * (renderer.h:553-603). */
int pt_prime(pt_ctx *ctx, int32_t n_rects, const int32_t *rects, int32_t min_ms);
/* The plan of the last render call / pt_reserve and the state of the streams. */
int pt_get_plan(pt_ctx *ctx, pt_plan *out);
/* The rule itself (host only, no device): samples per batch of a call of pixels x samples with `path_slots` slots per batch on
 * `lanes` lanes; *batches (optional) = how many batches that makes. */
int32_t pt_plan_batches(int64_t pixels, int32_t samples, int64_t path_slots, int32_t lanes, int32_t *batches);
/* Wall seconds of the last render call: from its entry into pt_render_async / pt_render_tiles_async to the moment the
 * device finished its last batch (a host function enqueued behind it stamps the clock), i.e. what a caller polling
 * without delay would measure -- Tiled::finalize's "time taken to compute" (renderer.h:700-706) without the 0.5 s
 * granularity of main.cpp:158-163's loop.  < 0 while the call is still running or before any call. */
double pt_render_seconds(pt_ctx *ctx);
/* Block until everything enqueued has finished or `timeout_ms` have passed (the sleep of main.cpp:162 that ends early):
 * 1 = idle, 0 = timed out, < 0 error. */
int pt_wait_for(pt_ctx *ctx, int32_t timeout_ms);
/* Non-blocking progress (Tiled::sync_progress renderer.h:605-620 reads samples_done[]):
 * returns 1 when everything enqueued so far has finished, 0 if still running, < 0 on error. */
int pt_poll(pt_ctx *ctx, uint64_t *samples_done, uint64_t *rays_done);
/* Block until idle (the reference's main loop spins on is_done(), main.cpp:158-163). */
int pt_wait(pt_ctx *ctx);
/* Copy the linear float framebuffer SUM (not mean; renderer.h:682) to host: height*width*3 floats,
 * row 0 = bottom row, i.e. framebuffer[j][i] of renderer.h:141.  Waits for pending work. */
int pt_read_framebuffer(pt_ctx *ctx, float *rgb_sum);
/* Progressive preview (Tiled::sync_progress renderer.h:605-620 re-writes the PPM from the LIVE framebuffer with the
 * divisor 1 + samples_done / (W*H)): copy the framebuffer SUM as it stands, without waiting for the work that is
 * still queued.  *samples_accumulated = camera samples of the batches known to be fully accumulated when the copy
 * started (a lower bound of what the copy holds; exact once the context is idle).  Same layout as above. */
int pt_snapshot_framebuffer(pt_ctx *ctx, float *rgb_sum, uint64_t *samples_accumulated);
int pt_clear_framebuffer(pt_ctx *ctx);
int pt_get_counters(pt_ctx *ctx, pt_counters *out);   /* totals since pt_create / pt_clear_framebuffer; waits */

/* Interop for multi-GPU reduction: the device address of the framebuffer (float[height*width*4], RGBA with
 * A unused, same row order) so that a caller can hand it to RCCL / torch.distributed without a host copy,
 * and an optional caller-owned device buffer to render into instead. */
void *pt_device_framebuffer(pt_ctx *ctx);
int pt_set_device_framebuffer(pt_ctx *ctx, void *device_rgba, size_t bytes);
/* HIP stream the context launches on (hipStream_t as void*); pt_set_stream(ctx, NULL) = own stream */
void *pt_get_stream(pt_ctx *ctx);
int pt_set_stream(pt_ctx *ctx, void *hip_stream);

/* Batches rotate over n of the context's stream lanes (default: all it owns, 3).  n = 1 runs the kernels of consecutive
 * batches one after the other: per-kernel measurements.  Returns the number of lanes the context owns, < 0 on error. */
int pt_set_lanes(pt_ctx *ctx, int32_t n);
/* Planner of a tile-partitioned render: rays_out[k] = the reference's ray count (pt_counters::rays: extension rays +
 * light_samples shadow rays per hit -- what a tile's time follows; up to ABI v6's first form: rays_traced) for `spp` samples
 * per pixel of rect k, for all rects in one pass (replaces one render + counter read per tile).  Framebuffer and counters
 * are cleared before and after. */
int pt_measure_tile_costs(pt_ctx *ctx, int32_t n_rects, const int32_t *rects, int32_t spp, uint64_t *rays_out);
/* The per-scene build of the traversal sweep: the scene's traversal program as a header text for pt_kernels.hip
 * (PT_SPEC_HEADER).  Host only (no device needed).  Returns the text length; buf receives it when cap > length.
 * < 0: the scene has no program to specialise (the generic kernels run). */
int pt_spec_header(const pt_scene_desc *scene, char *buf, size_t cap);
/* pt_create compiles the traversal kernels once more for the scene at hand (hiprtc; env PATHTRACE_HIP_SPEC = async
 * [default: on a thread of its own, used when ready] | sync | off).  pt_spec_status: 1 = in use, 0 = still building (the
 * generic kernels run meanwhile), -1 = not available (pt_last_error says why; the generic kernels run).  pt_spec_wait
 * blocks until the build has ended and returns the same.  The image is the same bit for bit either way. */
int pt_spec_status(pt_ctx *ctx);
int pt_spec_wait(pt_ctx *ctx);
/* ABI v5.  Provenance of the module: one line of JSON -- {"status", "built_by": "helper" | "in-process", "rtc_lib": path of
 * the libhiprtc that compiled, "producer": the code object's compiler string, "own_compiler": true iff that is the hipcc
 * that built this library}.  A foreign compiler builds correct but measurably slower kernels; bench.py labels such a run.
 * Returns the text length; buf receives it when cap > length. */
int pt_spec_info(pt_ctx *ctx, char *buf, size_t cap);
/* host-only check: compile the scene's module for gfx950 without a device; returns its code size, < 0 on failure */
long pt_spec_build_check(const pt_scene_desc *scene, int32_t light_samples);
/* the same check, answering with pt_spec_info's line for the module it built (ABI v5); < 0 when the build failed */
int pt_spec_build_info(const pt_scene_desc *scene, int32_t light_samples, char *buf, size_t cap);
int pt_set_profiling(pt_ctx *ctx, int enabled);
int pt_get_kernel_times(pt_ctx *ctx, pt_kernel_times *out);
/* debug / parity: radiance of every camera sample of the LAST batch rendered (de_nan not applied):
 * n = (x1-x0)*(y1-y0)*(spp_end-spp_begin) records of 4 floats, sample-major then row-major pixels. */
int pt_read_last_batch_radiance(pt_ctx *ctx, float *rgba, size_t max_records, size_t *n_records);

/* Validation hook: World::hit (world.h:17-20, with the integrator's t range (0.001, FLT_MAX)) for caller-supplied
 * rays, through the very traversal code the render kernels use.  n origins (float[3n]); rays_per_origin = 1 uses the
 * extension-ray instantiation, 2 and 4 the shared-origin instantiations of the shadow rays (dirs = float[3 * n * rays_per_origin]).
 * k0, k1, vol_dim = stream RNG key and dimension base for constant_medium free-flight draws.  Outputs per ray:
 * t and id = instance*8 + face, or -1 for a miss. */
int pt_trace_rays(pt_ctx *ctx, int64_t n, int32_t rays_per_origin, const float *origins, const float *dirs,
                  uint32_t k0, uint32_t k1, uint32_t vol_dim, float *t_out, int32_t *id_out);

/* ---- multi-GPU in one process ------------------------------------------------------------
 * One context per listed device (an ordinal may repeat: two contexts on one GPU rehearse the path on a one-GPU box).
 * The film is cut into block_w x block_h tiles in NaiveSpiral order (queue.h:68-127) and every tile is owned by one
 * device (cost-balanced over measured per-tile ray counts; PATHTRACE_HIP_MULTI=roundrobin: tile k -> device k mod n);
 * pt_multi_render_async enqueues each device's tiles as wavefront batches and returns, there is no communication while
 * rendering, and reading the framebuffer sums the per-device framebuffers into the first device -- every peer sends only
 * the pixels of the tiles it owns (packed, one device-to-device copy per peer over its own link, added on the root), or
 * one RCCL ncclReduce per device (PATHTRACE_HIP_MULTI=rccl).  The result is the
 * single-device image bit for bit.  Replaces the thread fan-out of Tiled::start_render (renderer.h:553-603). */
typedef struct pt_multi pt_multi;
pt_multi *pt_multi_create(const pt_scene_desc *scene, const pt_config *config, int32_t n_devices, const int32_t *devices,
                          int32_t block_w, int32_t block_h);
void pt_multi_destroy(pt_multi *m);
int pt_multi_render_async(pt_multi *m, int32_t spp_begin, int32_t spp_end);
/* ABI v6: pt_reserve + pt_spec_wait + pt_prime (prime_ms, 0 = none) for every device's share of the film; seconds from pt_multi_render_async's entry until the last
 * device finished (< 0 while running); timed wait (1 idle, 0 timed out). */
int pt_multi_reserve(pt_multi *m, int32_t samples, int32_t prime_ms);
double pt_multi_render_seconds(pt_multi *m);
int pt_multi_wait_for(pt_multi *m, int32_t timeout_ms);
int pt_multi_poll(pt_multi *m, uint64_t *samples_done, uint64_t *rays_done);      /* 1 done, 0 running, < 0 error */
int pt_multi_wait(pt_multi *m);
int pt_multi_read_framebuffer(pt_multi *m, float *rgb_sum);                        /* height*width*3, row 0 = bottom row */
int pt_multi_snapshot_framebuffer(pt_multi *m, float *rgb_sum, uint64_t *samples_accumulated);
int pt_multi_get_counters(pt_multi *m, pt_counters *out);                          /* summed over the devices */
int pt_multi_clear(pt_multi *m);
int pt_multi_device_count(pt_multi *m);
int pt_multi_get_device_counters(pt_multi *m, int32_t index, pt_counters *out);     /* of the index-th listed device */
uint64_t pt_multi_exchange_bytes(pt_multi *m);                                      /* device-to-device bytes of the last framebuffer sum */
int pt_multi_tile_owners(pt_multi *m, int32_t *owners, int32_t max_tiles);         /* owner index per spiral tile; returns the tile count */

const char *pt_last_error(void);
int pt_abi_version(void);
int pt_device_count(void);   /* 0 when no HIP device is usable */

/* ---- host front end (C++ inside the library, C ABI outside) -----------------------------------------
 * The reference's input formats: config.json (config.h:98-131) and the scene JSON
 * (scene_parser.h:241-595, main.cpp:86-104).  pth_load builds exactly what main() builds before
 * renderer->start_render(): Config, World (instances, BVH, lights, background) and the camera. */
typedef struct pth_scene pth_scene;

typedef struct pth_config {     /* every Config / s_film field (config.h:11-96) */
    int32_t width, height;
    float exposure, gamma;      /* stored swapped, as config.h:24-25 does */
    char ppm_output_path[512], png_output_path[512];
    char traced_paths_output_path[512], traced_paths_2d_output_path[512];
    char scene_path[512];
    int32_t should_trace_paths;
    float avg_number_of_paths;
    int32_t block_width, block_height;
    float trace_probability;
    int32_t render_type;        /* 0 naive, 1 progressive, 2 tiled (config.h:30-44), 3 hip_wavefront (ours) */
    int32_t only_direct_illumination;
    int32_t integrator_type;    /* config.h:46-72; 4 = INEEPT */
    int32_t max_bounces, samples, light_samples;
    uint32_t threads;
    float normal_offset;
    int32_t russian_roulette;
} pth_config;

/* Parse a config.json text / file.  Missing required keys fail like the reference's .get<>() throws. */
int pth_config_from_file(const char *path, pth_config *out);
int pth_config_from_json(const char *json_text, pth_config *out);
/* Parse a scene JSON file and build the flat scene for the given film size (camera aspect = width/height). */
pth_scene *pth_scene_from_file(const char *path, int32_t width, int32_t height);
pth_scene *pth_scene_from_json(const char *json_text, int32_t width, int32_t height);
const pt_scene_desc *pth_scene_desc(const pth_scene *s);
void pth_scene_free(pth_scene *s);
/* NaiveSpiral tile order (queue.h:68-127): writes up to max_tiles rects (x0,y0,x1,y1) and returns the count */
int pth_spiral_tiles(int32_t width, int32_t height, int32_t block_w, int32_t block_h, int32_t *rects, int32_t max_tiles);
/* Film output (renderer.h:24-55, helpers.h:146-168, tonemap.h:4-24): P6 PPM of the SUM framebuffer */
int pth_write_ppm(const char *path, const float *rgb_sum, int32_t width, int32_t height, int32_t samples, float exposure_field);
/* The whole program of main.cpp:108-168 for render_type "hip_wavefront": config.json in `workdir`. */
int pth_main(const char *workdir);

#ifdef __cplusplus
}
#endif
#endif /* PATHTRACE_HIP_H */
