/* TEST INFRASTRUCTURE -- CPU restatement of the reference's per-pixel NEE path-tracing hot
 * path.  This is the checker, never the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Two RNG/arithmetic modes (SURVEY.md section 7 "parity ladder"):
 *   PTO_MODE_MT      one process-wide mt19937 in the reference's exact draw order, glibc libm,
 *                    the reference's own tile / row / column order.  Pinned bit-for-bit against
 *                    oracle/_ref (the reference's headers compiled in place) -> tests/golden.
 *   PTO_MODE_STREAM  same estimator, but every draw is hash(pixel, sample, dimension) and the
 *                    few transcendentals are the portable ptm_* polynomials below, so that a
 *                    GPU can reproduce it bit-for-bit.  This is the twin the HIP path is
 *                    compared against, and the multi-threaded CPU baseline.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTO_MODE_MT 0
#define PTO_MODE_STREAM 1

enum { PTO_MAT_LAMBERTIAN = 0, PTO_MAT_METAL = 1, PTO_MAT_DIELECTRIC = 2, PTO_MAT_DIFFUSE_LIGHT = 3, PTO_MAT_ISOTROPIC = 4 };
enum { PTO_PRIM_RECT = 0, PTO_PRIM_BOX = 1, PTO_PRIM_SPHERE = 2, PTO_PRIM_VOLUME = 3 };
enum { PTO_PLANE_XY = 0, PTO_PLANE_XZ = 1, PTO_PLANE_YZ = 2 };
enum { PTO_TEX_CONSTANT = 0, PTO_TEX_CHECKER = 1, PTO_TEX_PERLIN = 2, PTO_TEX_IMAGE = 3 };

typedef struct {
    int32_t type;
    float color[3];
    float alpha, power;
    int32_t two_sided;
    float fuzz, ior;
    int32_t texture; /* albedo / emit texture (index into the texture table) when it is not a constant one, else -1 */
} pto_material;

/* texture.h: constant_texture :14-31, checker_texture :33-75, noise_texture :185-196; image.h:7-50 */
typedef struct {
    int32_t type;
    float color[3];
    float alpha;
    int32_t even, odd; /* checker: texture indices (smaller than this texture's own index) */
    float scale;       /* checker, perlin */
    int32_t width, height;
    int64_t texel_offset; /* image: byte offset of its RGBA8 pixels (row 0 first) in the texel blob */
} pto_texture;

typedef struct {
    int32_t type, mat;
    float rect[5]; /* x0 z0 x1 z1 y   (reference primitive.h:120) */
    int32_t plane, flipped;
    float p0[3], p1[3];
    float center[3], radius;
    int32_t boundary;
    float density;
    int32_t phase_mat;
} pto_prim;

typedef struct {
    int32_t prim;
    float scale[3], rotate[3], translate[3];
    int32_t is_light;
} pto_instance;

typedef struct {
    float look_from[3], look_at[3];
    float fov, aperture, dist_to_focus;
} pto_camera;

typedef struct {
    int32_t width, height, samples;
    int32_t max_bounces, light_samples, russian_roulette, only_direct;
    int32_t block_w, block_h;
    float normal_offset;
} pto_config;

/* per-render counters: rays = extension + shadow (reference integrator.h:192,247) */
typedef struct {
    uint64_t rays, ext_rays, ext_hits, shadow_rays;
    uint64_t term_miss, term_rr, term_emitter, term_pdf, term_bounce_limit;
} pto_counters;

typedef struct pto_scene pto_scene;

/* Build the scene runtime (transform3 / instance / bvh_node constructors).  The mt19937 is
 * seeded 5489, the 1533 Perlin static-init draws are consumed (texture.h:180-183), then one
 * draw per bvh_node (bvh.h:135).  Returns NULL on invalid input. */
pto_scene *pto_scene_create(const pto_material *mats, int nmat, const pto_prim *prims, int nprim,
                            const pto_instance *insts, int ninst, const pto_camera *cam,
                            const float background[3]);
/* Same with a texture table: the Perlin tables are the ones the 1533 static-init draws produce. */
pto_scene *pto_scene_create_textured(const pto_material *mats, int nmat, const pto_prim *prims, int nprim,
                                     const pto_instance *insts, int ninst, const pto_camera *cam,
                                     const float background[3], const pto_texture *tex, int ntex,
                                     const uint8_t *texels, int64_t texel_bytes, int background_texture);
void pto_scene_destroy(pto_scene *s);
/* perlin::ranvec (256 x 3) and perm_x / perm_y / perm_z (256 each) after static init (texture.h:180-183) */
void pto_perlin_tables(float ranvec[768], int32_t perm[768]);
/* texture value(u, v, p) and alpha(u, v, p) of texture `ti` (MT mode: libm sinf; stream mode: ptm_sinf): out = r g b alpha */
void pto_texture_eval(const pto_scene *s, int ti, int mode, float u, float v, const float p[3], float out[4]);

/* tables for comparison with oracle/_ref and with the product's flattened scene */
int pto_scene_num_instances(const pto_scene *s);
void pto_scene_instance_tables(const pto_scene *s, int i, float fwd[12], float inv[12], float bbox[6]);
int pto_scene_num_nodes(const pto_scene *s);
/* preorder nodes: bbox[6], left, right; child >= 0 is a node index, child < 0 is ~instance */
void pto_scene_node(const pto_scene *s, int n, float bbox[6], int32_t *left, int32_t *right);
int pto_scene_num_lights(const pto_scene *s);
int pto_scene_light(const pto_scene *s, int k);
/* camera (camera.h:9-36) for a given aspect: origin, llc, horizontal, vertical, u, v, w (21 floats) + lens_radius */
void pto_scene_camera(const pto_scene *s, int width, int height, float out[22]);
/* next value of the global mt stream (advances it) */
double pto_scene_next_random(pto_scene *s);
/* first n random_double() values after static init (i.e. after the 1533 Perlin draws) */
void pto_rng_after_static_init(int n, double *out);
/* stream-mode generator, exposed for tests: the 32-bit draw for (seed, pixel, sample, dim) */
uint32_t pto_stream_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim);
/* dims per bounce in stream mode for this scene/config (layout documented in pt_oracle.c) */
int pto_stream_dims_per_bounce(const pto_scene *s, int light_samples);

/* MT mode: whole image, reference order (renderer.h:626-691 with NaiveSpiral queue.h:68-127),
 * one thread.  fb = height*width*3 floats, row 0 = bottom row, SUM of samples (not mean). */
void pto_render_mt(pto_scene *s, const pto_config *cfg, float *fb, pto_counters *ctr);
/* MT mode, first n camera samples in the same order: 13 floats each
 * (u v  ray.A(3) ray.B(3) time  col(3)  rays) -- same layout as ref_driver "samples". */
void pto_samples_mt(pto_scene *s, const pto_config *cfg, int n, float *out);
/* MT mode: world->hit of the first n camera rays: 9 floats (hit t p(3) n(3) inst). */
void pto_hits_mt(pto_scene *s, const pto_config *cfg, int n, float *out);

/* STREAM mode: accumulate samples [s0,s1) of pixels [x0,x1)x[y0,y1) into fb (full-image
 * layout as above), nthreads worker threads over rows. */
void pto_render_stream(const pto_scene *s, const pto_config *cfg, uint32_t seed, int x0, int y0, int x1,
                       int y1, int s0, int s1, int nthreads, float *fb, pto_counters *ctr);
/* STREAM mode, one camera sample: radiance (after de_nan) and its counters. */
void pto_sample_stream(const pto_scene *s, const pto_config *cfg, uint32_t seed, int i, int j, int sample,
                       float rgb[3], pto_counters *ctr);

/* STREAM mode World::hit (world.h:17-20; t range (0.001, FLT_MAX) as in integrator.h:193) for n explicit rays.  k0, k1 =
 * stream key, vol_dim = dimension of volume ordinal 0.  out: hit flag, t, instance index per ray. */
void pto_world_hit_stream(const pto_scene *s, int64_t n, const float *origins, const float *dirs, uint32_t k0, uint32_t k1,
                          uint32_t vol_dim, int32_t *hit, float *t, int32_t *inst);

/* portable math used by stream mode (exposed for tests against libm) */
void ptm_sincos_2pi(float r, float *s, float *c);
float ptm_cbrtf(float x);
float ptm_logf(float x);
float ptm_sinf(float x);
float ptm_atan2f(float y, float x);
float ptm_acosf(float x);

#ifdef __cplusplus
}
#endif
#endif
