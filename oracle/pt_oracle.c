/* TEST INFRASTRUCTURE -- see pt_oracle.h.  CPU restatement of the reference hot path.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * repository root).  Float/double promotions are written out explicitly where C++ overload
 * resolution in the reference picks a float or a double operation, because MT mode is
 * required to be bit-identical to the g++ -O3 build of the reference (oracle/_ref).
 * Compile with -ffp-contract=off and without -ffast-math (oracle/Makefile).
 */
#define _GNU_SOURCE
#include "pt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * vec3  (vec3.h:11-199)
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }            /* vec3.h:85 */
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }            /* vec3.h:100 */
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }            /* vec3.h:105 */
static inline v3 vscale(float t, v3 v) { return V(t * v.x, t * v.y, t * v.z); }             /* vec3.h:110,115 */
static inline v3 vdivf(v3 v, float t) { return V(v.x / t, v.y / t, v.z / t); }              /* vec3.h:125 */
static inline v3 vneg(v3 v) { return V(-v.x, -v.y, -v.z); }                                 /* vec3.h:42 */
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          /* vec3.h:130 */
static inline v3 vcross(v3 a, v3 b)                                                         /* vec3.h:135 */
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vsqlen(v3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }              /* vec3.h:57 */
static inline float vlen(v3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }         /* vec3.h:56 (std::sqrt(float)) */
static inline v3 vunit(v3 v) { return vdivf(v, vlen(v)); }                                  /* vec3.h:196, :191 */
static inline float vget(v3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
static inline int is_nanf(float x) { return !(x == x); }                                   /* helpers.h:45 */
static inline int v_is_nan(v3 v) { return is_nanf(v.x) || is_nanf(v.y) || is_nanf(v.z); }  /* helpers.h:50 */
static inline v3 de_nan(v3 c)                                                              /* helpers.h:60-76 */
{
    if (is_nanf(c.x)) c.x = 0;
    if (is_nanf(c.y)) c.y = 0;
    if (is_nanf(c.z)) c.z = 0;
    return c;
}
/* vec3::operator*=(float), operator/=(float): vec3.h:167-186 */
static inline v3 vscale_assign(v3 v, float t) { return V(v.x * t, v.y * t, v.z * t); }

/* ------------------------------------------------------------------------------------------
 * mt19937 + uniform_real_distribution<double>  (random.h:9-15; libstdc++ generate_canonical)
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } mt19937;

static void mt_seed(mt19937 *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937 *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
/* generate_canonical<double,53>: two 32-bit draws, (lo + hi*2^32)/2^64, clamped below 1 */
static double mt_double(mt19937 *g)
{
    double lo = (double)mt_next(g);
    double hi = (double)mt_next(g);
    double sum = lo + hi * 4294967296.0;
    double r = sum / 18446744073709551616.0;
    if (r >= 1.0) r = nextafter(1.0, 0.0);
    return r;
}
#define PERLIN_STATIC_DRAWS 1533 /* texture.h:99-110 (256*3) + texture.h:76-97 (3*255), run at texture.h:180-183 */

/* ------------------------------------------------------------------------------------------
 * stream-mode generator and portable math (our definition; the HIP path uses the same)
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t mix_lowbias32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
typedef struct { uint32_t k0, k1; } stream_key;
static inline stream_key stream_make_key(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    stream_key k;
    k.k0 = mix_lowbias32(pixel ^ mix_lowbias32(seed));
    k.k1 = mix_lowbias32(sample ^ mix_lowbias32(seed + 0x632BE5ABu));
    return k;
}
static inline uint32_t stream_u32(stream_key k, uint32_t dim)
{
    uint32_t x = k.k0 + dim * 0x9E3779B9u;
    x ^= x >> 17; x *= 0xed5ad4bbU;
    x ^= k.k1;
    x ^= x >> 11; x *= 0xac4c1b51U;
    x ^= x >> 15; x *= 0x31848babU;
    x ^= x >> 14;
    return x;
}
uint32_t pto_stream_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim)
{
    return stream_u32(stream_make_key(seed, pixel, sample), dim);
}

/* sin and cos of 2*pi*r for r in [0,1]: quadrant reduction on r (exact), then odd/even
 * polynomials on [-pi/4, pi/4].  Pure float +,-,* in a fixed order: identical on CPU and GPU. */
void ptm_sincos_2pi(float r, float *s, float *c)
{
    float t = r * 4.0f;
    float q = floorf(t + 0.5f);
    float f = t - q;                  /* [-0.5, 0.5], exact */
    float a = f * 1.57079637f;        /* (float)(pi/2) */
    float a2 = a * a;
    float sp = a + a * a2 * (-0.16666667f + a2 * (0.0083333310f + a2 * (-0.00019840874f + a2 * 2.7525562e-06f)));
    float cp = 1.0f + a2 * (-0.5f + a2 * (0.041666638f + a2 * (-0.0013888378f + a2 * 2.4760495e-05f)));
    int qi = ((int)q) & 3;
    float ss, cc;
    if (qi == 0) { ss = sp; cc = cp; }
    else if (qi == 1) { ss = cp; cc = -sp; }
    else if (qi == 2) { ss = -sp; cc = -cp; }
    else { ss = -cp; cc = sp; }
    *s = ss;
    *c = cc;
}
/* sinf for checker_texture::sines (texture.h:70-73).  Double arithmetic throughout (+,-,*, rint; no contraction), so the
 * CPU and the GPU agree bit for bit; the float result is within 1 ulp of libm's for |x| < 1e6 (three-term Cody-Waite
 * reduction whose first two products are exact for |k| < 2^20), the same formula with slowly degrading accuracy up to
 * 2^30, and x - x (NaN for inf / NaN, 0 otherwise) beyond. */
static double ptm_sin_reduced(double r, int q)
{
    const double r2 = r * r;
    const double sp = r + r * r2 * (-1.6666666666666666e-01 + r2 * (8.3333333333333332e-03 + r2 * (-1.9841269841269841e-04
                      + r2 * (2.7557319223985893e-06 + r2 * (-2.5052108385441720e-08 + r2 * 1.6059043836821613e-10)))));
    const double cp = 1.0 + r2 * (-0.5 + r2 * (4.1666666666666664e-02 + r2 * (-1.3888888888888889e-03 + r2 * (2.4801587301587302e-05
                      + r2 * (-2.7557319223985888e-07 + r2 * (2.0876756987868100e-09 + r2 * -1.1470745597729725e-11))))));
    return (q & 1) ? ((q & 2) ? -cp : cp) : ((q & 2) ? -sp : sp);
}
float ptm_sinf(float x)
{
    if (!(fabsf(x) < 1073741824.0f)) return x - x;
    const double xd = (double)x;
    const double k = rint(xd * 0.63661977236758138);
    /* pi/2 = C1 + C2 + C3, C1 and C2 with 33 significant bits */
    const double r = ((xd - k * 1.5707963267341256) - k * 6.077100506303966e-11) - k * 2.0222662487959506e-21;
    return (float)ptm_sin_reduced(r, (int)k & 3);
}
/* atan2f / acosf for the environment-map coordinates of a missed ray (integrator.h:327-330); same rules as ptm_sinf */
static double ptm_atan2_d(double y, double x)
{
    if (x != x || y != y) return x + y;
    const double ax = fabs(x), ay = fabs(y);
    const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    double a = (mx > 0.0) ? mn / mx : 0.0;            /* [0, 1] */
    double off = 0.0;
    if (a > 0.41421356237309503) { a = (a - 1.0) / (a + 1.0); off = 0.78539816339744828; }   /* atan a = pi/4 + atan((a-1)/(a+1)) */
    const double z = a * a;
    double p = 1.0 / 27.0;
    p = 1.0 / 25.0 - z * p; p = 1.0 / 23.0 - z * p; p = 1.0 / 21.0 - z * p; p = 1.0 / 19.0 - z * p; p = 1.0 / 17.0 - z * p;
    p = 1.0 / 15.0 - z * p; p = 1.0 / 13.0 - z * p; p = 1.0 / 11.0 - z * p; p = 1.0 / 9.0 - z * p; p = 1.0 / 7.0 - z * p;
    p = 1.0 / 5.0 - z * p; p = 1.0 / 3.0 - z * p; p = 1.0 - z * p;
    double t = off + a * p;                          /* atan(mn / mx) in [0, pi/4] */
    if (ay > ax) t = 1.5707963267948966 - t;
    if (x < 0.0) t = 3.1415926535897931 - t;
    return (y < 0.0) ? -t : t;
}
float ptm_atan2f(float y, float x)
{
    return (float)ptm_atan2_d((double)y, (double)x);
}
float ptm_acosf(float x)
{
    const double xd = (double)x;
    return (float)ptm_atan2_d(sqrt((1.0 - xd) * (1.0 + xd)), xd);   /* NaN outside [-1, 1], like libm */
}
/* cube root for x in [0, 1]: bit-trick seed + 3 Newton steps (float +,-,*,/ only) */
float ptm_cbrtf(float x)
{
    if (!(x > 0.0f)) return 0.0f;
    union { float f; uint32_t u; } b;
    b.f = x;
    b.u = b.u / 3u + 709921077u;
    float y = b.f;
    for (int i = 0; i < 3; i++) y = y - (y * y * y - x) / (3.0f * y * y);
    return y;
}
/* natural log for x in [0, +inf): x = m * 2^e, m in [sqrt(.5), sqrt(2)); log m = 2 atanh((m-1)/(m+1)) */
float ptm_logf(float x)
{
    if (!(x > 0.0f)) return -INFINITY;
    union { float f; uint32_t u; } b;
    b.f = x;
    int e = 0;
    if (b.u < 0x00800000u) { b.f = x * 8388608.0f; e = -23; } /* subnormal */
    e += (int)(b.u >> 23) - 127;
    b.u = (b.u & 0x007fffffu) | 0x3f800000u; /* m in [1,2) */
    float m = b.f;
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float p = 2.0f * s * (1.0f + z * (0.33333334f + z * (0.2f + z * (0.14285715f + z * 0.11111111f))));
    float fe = (float)e;
    return fe * 0.693145752f + (fe * 1.42860677e-06f + p); /* ln2 split hi/lo */
}

/* ------------------------------------------------------------------------------------------
 * scene runtime
 * ---------------------------------------------------------------------------------------- */
typedef struct { float m[3][4]; } affine; /* rows of an Eigen::Affine3f (transform3.h:69) */
typedef struct { v3 mn, mx; } aabb;        /* aabb.h:10-32 */

typedef struct {
    int type;
    v3 color;
    float alpha, power;
    int two_sided;
    int tex; /* texture index when albedo / emit is not a constant texture, else -1 */
} mat_t;

typedef struct {
    int type;
    v3 color;
    float alpha;
    int even, odd;
    float scale;
    int width, height;
    const uint8_t *rgba; /* image: into pto_scene.texels */
} tex_t;

typedef struct { float x0, z0, x1, z1, y; int plane; int normal; /* = !flipped */ int mat; } rect_t;

typedef struct {
    int type, mat;
    rect_t rect;      /* PRIM_RECT */
    rect_t sides[6];  /* PRIM_BOX (primitive.h:232-240) */
    v3 center;        /* PRIM_SPHERE */
    float radius;
    int boundary;     /* PRIM_VOLUME */
    float density;
    int phase_mat;
} prim_t;

typedef struct {
    int prim;
    affine fwd, inv;
    aabb bbox;
    int vol_ordinal; /* ordinal among volume instances, -1 otherwise (stream-mode dimension slot) */
} inst_t;

typedef struct { aabb box; int left, right; } node_t;

struct pto_scene {
    int nmat, nprim, ninst, nnode, nlight, nvol;
    mat_t *mats;
    prim_t *prims;
    inst_t *insts;
    node_t *nodes;
    int *lights;
    pto_camera cam;
    v3 background;
    int background_tex; /* World::background when it is not a constant texture, else -1 */
    int ntex;
    tex_t *tex;
    uint8_t *texels;
    v3 ranvec[256];      /* perlin::ranvec, perm_x/y/z (texture.h:176-183), from the static-init draws */
    int perm[3][256];
    mt19937 rng; /* the process-wide generator of random.h:12 */
};

/* hit_record (hittable.h:10-19) */
typedef struct {
    float t;
    v3 p, normal;
    int inst; /* rec.primitive after instance::hit, else -1 */
    float u, v;
    int mat;
} hitrec;

typedef struct { v3 A, B; } ray_t; /* ray.h:5-29; _time never influences the hot path */

/* RNG context handed down the call tree.  MT mode ignores `dim`, stream mode ignores order. */
typedef struct {
    int mode;
    mt19937 *mt;
    stream_key key;
    uint32_t vol_dim_base; /* stream: dimension of volume-ordinal 0 for the current traversal */
} rngctx;

static inline double rnd(rngctx *c, uint32_t dim)
{
    if (c->mode == PTO_MODE_MT) return mt_double(c->mt);
    return (double)stream_u32(c->key, dim) * (1.0 / 4294967296.0);
}

/* ---- transform3 (transform3.h:19-68, Eigen 3.2.10 code paths per SURVEY.md A.4) ---------- */
typedef struct { float x, y, z, w; } quat;

static quat quat_from_angle_axis(float angle, float ax, float ay, float az)
{   /* Eigen/src/Geometry/Quaternion.h QuaternionBase::operator=(AngleAxis) */
    float ha = 0.5f * angle;
    float s = sinf(ha);
    quat q;
    q.w = cosf(ha);
    q.x = s * ax; q.y = s * ay; q.z = s * az;
    return q;
}
static quat quat_mul(quat a, quat b)
{   /* Eigen/src/Geometry/arch/Geometry_SSE.h:19-41 (SSE float path, lane-wise) */
    quat r;
    r.x = (a.x * b.w - a.z * b.y) + (a.y * b.z + a.w * b.x);
    r.y = (a.y * b.w - a.x * b.z) + (a.z * b.x + a.w * b.y);
    r.z = (a.z * b.w - a.y * b.x) + (a.x * b.y + a.w * b.z);
    r.w = (a.w * b.w - a.x * b.x) + (-(a.z * b.z) + -(a.y * b.y));
    return r;
}
static affine affine_compose(const float scale[3], const float rotate[3], const float translate[3])
{   /* transform3.h:19-25:  t_translate * (AAx * AAy * AAz) * t_scale */
    float ax = (float)((double)rotate[0] * M_PI);
    float ay = (float)((double)rotate[1] * M_PI);
    float az = (float)((double)rotate[2] * M_PI);
    quat q = quat_mul(quat_mul(quat_from_angle_axis(ax, 1, 0, 0), quat_from_angle_axis(ay, 0, 1, 0)),
                      quat_from_angle_axis(az, 0, 0, 1));
    /* Quaternion.h:525-557 toRotationMatrix */
    float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
    float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    float R[3][3];
    R[0][0] = 1.0f - (tyy + tzz); R[0][1] = txy - twz;          R[0][2] = txz + twy;
    R[1][0] = txy + twz;          R[1][1] = 1.0f - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy;          R[2][1] = tyz + twx;          R[2][2] = 1.0f - (txx + tyy);
    affine a;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) a.m[i][j] = R[i][j] * scale[j]; /* Transform * DiagonalMatrix: linear() *= b */
        a.m[i][3] = 0.0f + translate[i];                            /* Translation * Transform: pretranslate */
    }
    return a;
}
static affine affine_inverse(const affine *f)
{   /* Eigen/src/Geometry/Transform.h:1158-1184 (Affine) + LU/Inverse.h:116-159 */
    const float (*m)[4] = f->m;
#define COF(i, j) (m[((i) + 1) % 3][((j) + 1) % 3] * m[((i) + 2) % 3][((j) + 2) % 3] - m[((i) + 1) % 3][((j) + 2) % 3] * m[((i) + 2) % 3][((j) + 1) % 3])
    float c0 = COF(0, 0), c1 = COF(1, 0), c2 = COF(2, 0);
    float det = c0 * m[0][0] + (c1 * m[1][0] + c2 * m[2][0]); /* redux of 3: a0 + (a1 + a2) */
    float invdet = 1.0f / det;
    affine r;
    r.m[0][0] = c0 * invdet; r.m[0][1] = c1 * invdet; r.m[0][2] = c2 * invdet;
    r.m[1][0] = COF(0, 1) * invdet; r.m[1][1] = COF(1, 1) * invdet; r.m[1][2] = COF(2, 1) * invdet;
    r.m[2][0] = COF(0, 2) * invdet; r.m[2][1] = COF(1, 2) * invdet; r.m[2][2] = COF(2, 2) * invdet;
#undef COF
    for (int i = 0; i < 3; i++) /* topRightCorner = (-L^-1) * translation : coefficient-based product */
        r.m[i][3] = ((-r.m[i][0]) * m[0][3] + (-r.m[i][1]) * m[1][3]) + (-r.m[i][2]) * m[2][3];
    return r;
}
/* transform3::operator*(vec3) (transform3.h:65-68): translation + linear*v */
static inline v3 xf_point(const affine *a, v3 p)
{
    return V(a->m[0][3] + ((a->m[0][0] * p.x + a->m[0][1] * p.y) + a->m[0][2] * p.z),
             a->m[1][3] + ((a->m[1][0] * p.x + a->m[1][1] * p.y) + a->m[1][2] * p.z),
             a->m[2][3] + ((a->m[2][0] * p.x + a->m[2][1] * p.y) + a->m[2][2] * p.z));
}
/* transform3::apply_linear (transform3.h:56-59) */
static inline v3 xf_linear(const affine *a, v3 v)
{
    return V((a->m[0][0] * v.x + a->m[0][1] * v.y) + a->m[0][2] * v.z,
             (a->m[1][0] * v.x + a->m[1][1] * v.y) + a->m[1][2] * v.z,
             (a->m[2][0] * v.x + a->m[2][1] * v.y) + a->m[2][2] * v.z);
}
/* transform3::apply_normal (transform3.h:60-63): normalize((L^-1)^T n); Eigen norm = sqrt(x2 + (y2 + z2)) */
static inline v3 xf_normal(const affine *inv, v3 n)
{
    float x = (inv->m[0][0] * n.x + inv->m[1][0] * n.y) + inv->m[2][0] * n.z;
    float y = (inv->m[0][1] * n.x + inv->m[1][1] * n.y) + inv->m[2][1] * n.z;
    float z = (inv->m[0][2] * n.x + inv->m[1][2] * n.y) + inv->m[2][2] * n.z;
    float nrm = sqrtf(x * x + (y * y + z * z));
    return V(x / nrm, y / nrm, z / nrm);
}

/* ---- aabb (aabb.h:34-64) ------------------------------------------------------------------ */
static inline float ffmin(float a, float b) { return a < b ? a : b; }
static inline float ffmax(float a, float b) { return a > b ? a : b; }
static aabb surrounding_box(aabb a, aabb b)
{
    aabb r;
    r.mn = V(ffmin(a.mn.x, b.mn.x), ffmin(a.mn.y, b.mn.y), ffmin(a.mn.z, b.mn.z));
    r.mx = V(ffmax(a.mx.x, b.mx.x), ffmax(a.mx.y, b.mx.y), ffmax(a.mx.z, b.mx.z));
    return r;
}
static int aabb_hit(const aabb *b, const ray_t *r, float tmin, float tmax)
{
    for (int a = 0; a < 3; a++) {
        float invD = 1.0f / vget(r->B, a);
        float t0 = (vget(b->mn, a) - vget(r->A, a)) * invD;
        float t1 = (vget(b->mx, a) - vget(r->A, a)) * invD;
        if (invD < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return 0;
    }
    return 1;
}

/* ---- rect / box / sphere / constant_medium (primitive.h, volume.h) ------------------------- */
static inline v3 shuffle(v3 v, int plane)
{   /* primitive.h:104-121 */
    if (plane == PTO_PLANE_XY) return V(v.x, v.z, v.y);
    if (plane == PTO_PLANE_YZ) return V(v.y, v.x, v.z);
    return v;
}
static int rect_hit(const rect_t *q, const ray_t *r, float t0, float t1, hitrec *rec)
{   /* primitive.h:186-225 */
    v3 o = shuffle(r->A, q->plane);
    v3 d = shuffle(r->B, q->plane);
    float t = (q->y - o.y) / d.y;
    if (t < t0 || t > t1) return 0;
    float xh = o.x + t * d.x;
    float zh = o.z + t * d.z;
    if (xh < q->x0 || xh > q->x1 || zh < q->z0 || zh > q->z1) return 0;
    rec->u = (xh - q->x0) / (q->x1 - q->x0);
    rec->v = (zh - q->x0) / (q->z1 - q->z0); /* sic: x0 (primitive.h:207) */
    rec->t = t;
    rec->mat = q->mat;
    rec->p = vadd(r->A, vscale(t, r->B));
    rec->normal = shuffle(V(0, (float)(2 * q->normal - 1), 0), q->plane);
    if (vdot(r->B, rec->normal) > 0) rec->normal = vneg(rec->normal); /* two_sided is always true */
    rec->inst = -1;
    return 1;
}
static aabb rect_bbox(const rect_t *q)
{   /* primitive.h:140-149 */
    v3 a = V(q->x0, (float)((double)q->y - 0.001), q->z0);
    v3 b = V(q->x1, (float)((double)q->y + 0.001), q->z1);
    aabb r;
    r.mn = shuffle(a, q->plane);
    r.mx = shuffle(b, q->plane);
    return r;
}
static int box_hit(const prim_t *p, const ray_t *r, float t0, float t1, hitrec *rec)
{   /* primitive.h:243-246 -> hittable_list.h:21-38 */
    hitrec tmp;
    int hit_anything = 0;
    double closest = t1;
    for (int i = 0; i < 6; i++) {
        if (rect_hit(&p->sides[i], r, t0, (float)closest, &tmp)) {
            hit_anything = 1;
            closest = tmp.t;
            *rec = tmp;
        }
    }
    return hit_anything;
}
static int sphere_hit(const prim_t *s, const ray_t *r, float t_min, float t_max, hitrec *rec)
{   /* primitive.h:64-95 */
    v3 oc = vsub(r->A, s->center);
    float a = vdot(r->B, r->B);
    float b = vdot(oc, r->B);
    float c = vdot(oc, oc) - s->radius * s->radius;
    float disc = b * b - a * c;
    if (disc > 0) {
        float temp = (-b - sqrtf(disc)) / a;
        for (int k = 0; k < 2; k++) {
            if (temp < t_max && temp > t_min) {
                rec->t = temp;
                rec->p = vadd(r->A, vscale(temp, r->B));
                rec->normal = vdivf(vsub(rec->p, s->center), s->radius);
                rec->mat = s->mat;
                rec->inst = -1;
                rec->u = rec->v = 0;
                return 1;
            }
            temp = (-b + sqrtf(disc)) / a;
        }
    }
    return 0;
}
static int prim_hit(const pto_scene *sc, const prim_t *p, const ray_t *r, float t0, float t1, hitrec *rec,
                    rngctx *rc, int vol_ordinal);

/* Free-flight draws one constant_medium::hit call can make: its own, and -- when its boundary is a constant_medium itself
   (volume.h:10 takes any hittable) -- those of the two boundary->hit calls in front of it.  Stream mode gives every draw of
   a traversal a dimension of its own: slot_base .. + draws - 1, the first boundary call's first, then the second's, then
   this medium's own (the order the reference makes them in). */
static int volume_draws(const pto_scene *sc, const prim_t *p)
{
    if (p->type != PTO_PRIM_VOLUME) return 0;
    return 2 * volume_draws(sc, &sc->prims[p->boundary]) + 1;
}
static int volume_hit(const pto_scene *sc, const prim_t *p, const ray_t *r, float t_min, float t_max, hitrec *rec,
                      rngctx *rc, int vol_ordinal)
{   /* volume.h:29-93 */
    hitrec rec1, rec2;
    const prim_t *b = &sc->prims[p->boundary];
    const int nb = volume_draws(sc, b);
    if (prim_hit(sc, b, r, -FLT_MAX, FLT_MAX, &rec1, rc, vol_ordinal)) {
        if (prim_hit(sc, b, r, (float)((double)rec1.t + 0.0001), FLT_MAX, &rec2, rc, vol_ordinal + nb)) {
            if (rec1.t < t_min) rec1.t = t_min;
            if (rec2.t > t_max) rec2.t = t_max;
            if (rec1.t >= rec2.t) return 0;
            if (rec1.t < 0) rec1.t = 0;
            float dlen = vlen(r->B);
            float distance_inside = (rec2.t - rec1.t) * dlen;
            float hit_distance;
            if (rc->mode == PTO_MODE_MT) {
                hit_distance = (float)((double)(-(1 / p->density)) * log(rnd(rc, 0)));
            } else {
                float u = (float)rnd(rc, rc->vol_dim_base + (uint32_t)(vol_ordinal + 2 * nb));
                hit_distance = (-(1 / p->density)) * ptm_logf(u);
            }
            if (hit_distance < distance_inside) {
                rec->t = rec1.t + hit_distance / dlen;
                rec->p = vadd(r->A, vscale(rec->t, r->B));
                rec->normal = V(1, 0, 0);
                rec->mat = p->phase_mat;
                rec->inst = -1;
                rec->u = rec->v = 0;
                return 1;
            }
        }
    }
    return 0;
}
static int prim_hit(const pto_scene *sc, const prim_t *p, const ray_t *r, float t0, float t1, hitrec *rec,
                    rngctx *rc, int vol_ordinal)
{
    switch (p->type) {
    case PTO_PRIM_RECT: return rect_hit(&p->rect, r, t0, t1, rec);
    case PTO_PRIM_BOX: return box_hit(p, r, t0, t1, rec);
    case PTO_PRIM_SPHERE: return sphere_hit(p, r, t0, t1, rec);
    case PTO_PRIM_VOLUME: return volume_hit(sc, p, r, t0, t1, rec, rc, vol_ordinal);
    }
    return 0;
}
static aabb prim_bbox(const pto_scene *sc, const prim_t *p)
{
    switch (p->type) {
    case PTO_PRIM_RECT: return rect_bbox(&p->rect);
    case PTO_PRIM_BOX: { /* primitive.h:247-252 -> hittable_list.h:40-68 */
        aabb b = rect_bbox(&p->sides[0]);
        for (int i = 1; i < 6; i++) b = surrounding_box(b, rect_bbox(&p->sides[i]));
        return b;
    }
    case PTO_PRIM_SPHERE: { /* primitive.h:97-102 */
        aabb b;
        b.mn = vsub(p->center, V(p->radius, p->radius, p->radius));
        b.mx = vadd(p->center, V(p->radius, p->radius, p->radius));
        return b;
    }
    default: return prim_bbox(sc, &sc->prims[p->boundary]); /* volume.h:20-23 */
    }
}

/* ---- instance (primitive.h:258-348) ------------------------------------------------------- */
static int instance_hit(const pto_scene *sc, int ii, const ray_t *r, float t_min, float t_max, hitrec *rec, rngctx *rc)
{   /* primitive.h:298-312; ray::apply ray.h:20-24.  The per-call transform.inverse() is the
       same function of the same constants every time, so the stored inverse is bit-identical. */
    const inst_t *in = &sc->insts[ii];
    ray_t local;
    local.A = xf_point(&in->inv, r->A);
    local.B = xf_linear(&in->inv, r->B);
    if (prim_hit(sc, &sc->prims[in->prim], &local, t_min, t_max, rec, rc, in->vol_ordinal)) {
        rec->p = xf_point(&in->fwd, rec->p);
        rec->normal = xf_normal(&in->inv, rec->normal);
        rec->inst = ii;
        return 1;
    }
    return 0;
}
static float prim_pdf_value(const pto_scene *sc, const prim_t *p, v3 o, v3 v, int mode)
{
    hitrec rec;
    ray_t r = {o, v};
    if (p->type == PTO_PRIM_RECT) { /* primitive.h:151-166 */
        const rect_t *q = &p->rect;
        if (rect_hit(q, &r, (float)0.001, FLT_MAX, &rec)) {
            float area = (q->x1 - q->x0) * (q->z1 - q->z0);
            float vl = vlen(v);
            float d2 = (mode == PTO_MODE_MT) ? powf(rec.t * vl, (float)2.0) : (rec.t * vl) * (rec.t * vl);
            float cosine = fabsf(vdot(v, rec.normal) / vl);
            return d2 / (cosine * area);
        }
        return 0;
    }
    if (p->type == PTO_PRIM_SPHERE) { /* primitive.h:37-51 */
        if (sphere_hit(p, &r, (float)0.001, FLT_MAX, &rec)) {
            float cos_theta_max = sqrtf(1 - p->radius * p->radius / vsqlen(vsub(p->center, o)));
            float solid_angle = (float)(2 * M_PI * (double)(1 - cos_theta_max));
            return 1 / solid_angle;
        }
        return 0;
    }
    (void)sc;
    return 0.0f; /* hittable.h:27: box and constant_medium do not override */
}
static float instance_pdf_value(const pto_scene *sc, int ii, v3 o, v3 v, int mode)
{   /* primitive.h:319-337 */
    const inst_t *in = &sc->insts[ii];
    return prim_pdf_value(sc, &sc->prims[in->prim], xf_point(&in->inv, o), xf_linear(&in->inv, v), mode);
}
/* onb::build_from_w (helpers.h:127-136) */
typedef struct { v3 u, v, w; } onb;
static onb onb_from_w(v3 n)
{
    onb b;
    b.w = vunit(n);
    v3 a = (fabsf(b.w.x) > 0.9) ? V(0, 1, 0) : V(1, 0, 0);
    b.v = vunit(vcross(b.w, a));
    b.u = vcross(b.w, b.v);
    return b;
}
static inline v3 onb_local(const onb *b, v3 a)
{   /* helpers.h:123 */
    return vadd(vadd(vscale(a.x, b->u), vscale(a.y, b->v)), vscale(a.z, b->w));
}
static v3 prim_random(const pto_scene *sc, const prim_t *p, v3 o, rngctx *rc, uint32_t dim)
{
    (void)sc;
    if (p->type == PTO_PRIM_RECT) { /* primitive.h:168-175; g++ evaluates the z argument first */
        const rect_t *q = &p->rect;
        double rz = rnd(rc, dim + 0);
        double rx = rnd(rc, dim + 1);
        float pz = (float)((double)q->z0 + rz * (double)(q->z1 - q->z0));
        float px = (float)((double)q->x0 + rx * (double)(q->x1 - q->x0));
        return vsub(shuffle(V(px, q->y, pz), q->plane), o);
    }
    if (p->type == PTO_PRIM_SPHERE) { /* primitive.h:52-59 + random.h:45-55 */
        v3 direction = vsub(p->center, o);
        float d2 = vsqlen(direction);
        onb uvw = onb_from_w(direction);
        float r1 = (float)rnd(rc, dim + 0);
        float r2 = (float)rnd(rc, dim + 1);
        float z = 1 + r2 * (sqrtf(1 - p->radius * p->radius / d2) - 1);
        float s, c;
        if (rc->mode == PTO_MODE_MT) {
            float phi = (float)(2 * M_PI * (double)r1);
            c = cosf(phi); s = sinf(phi);
        } else {
            ptm_sincos_2pi(r1, &s, &c);
        }
        float x = c * sqrtf(1 - z * z);
        float y = s * sqrtf(1 - z * z);
        return onb_local(&uvw, V(x, y, z));
    }
    return V(1, 0, 0); /* hittable.h:28 */
}
static v3 instance_random(const pto_scene *sc, int ii, v3 o, rngctx *rc, uint32_t dim)
{   /* primitive.h:338-342 */
    const inst_t *in = &sc->insts[ii];
    return xf_linear(&in->fwd, prim_random(sc, &sc->prims[in->prim], xf_point(&in->inv, o), rc, dim));
}

/* ---- bvh_node (bvh.h) -------------------------------------------------------------------- */
static int node_or_leaf_hit(const pto_scene *sc, int child, const ray_t *r, float t_min, float t_max, hitrec *rec, rngctx *rc);

static int bvh_hit(const pto_scene *sc, int n, const ray_t *r, float t_min, float t_max, hitrec *rec, rngctx *rc)
{   /* bvh.h:31-69 */
    const node_t *nd = &sc->nodes[n];
    if (aabb_hit(&nd->box, r, t_min, t_max)) {
        hitrec lrec, rrec;
        int hl = node_or_leaf_hit(sc, nd->left, r, t_min, t_max, &lrec, rc);
        int hr = node_or_leaf_hit(sc, nd->right, r, t_min, t_max, &rrec, rc);
        if (hl && hr) {
            if (lrec.t < rrec.t) *rec = lrec; else *rec = rrec;
            return 1;
        } else if (hl) { *rec = lrec; return 1; }
        else if (hr) { *rec = rrec; return 1; }
        return 0;
    }
    return 0;
}
static int node_or_leaf_hit(const pto_scene *sc, int child, const ray_t *r, float t_min, float t_max, hitrec *rec, rngctx *rc)
{
    if (child >= 0) return bvh_hit(sc, child, r, t_min, t_max, rec, rc);
    return instance_hit(sc, ~child, r, t_min, t_max, rec, rc);
}
static inline int world_hit(const pto_scene *sc, const ray_t *r, hitrec *rec, rngctx *rc)
{   /* world.h:17-20 with the integrator's (0.001, MAXFLOAT) (integrator.h:193,246) */
    return bvh_hit(sc, 0, r, (float)0.001, FLT_MAX, rec, rc);
}

void pto_world_hit_stream(const pto_scene *s, int64_t n, const float *origins, const float *dirs, uint32_t k0, uint32_t k1,
                          uint32_t vol_dim, int32_t *hit, float *t, int32_t *inst)
{
    for (int64_t i = 0; i < n; i++) {
        rngctx rc;
        memset(&rc, 0, sizeof(rc));
        rc.mode = PTO_MODE_STREAM;
        rc.key.k0 = k0; rc.key.k1 = k1;
        rc.vol_dim_base = vol_dim;
        ray_t r;
        r.A = V(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
        r.B = V(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        hitrec rec;
        memset(&rec, 0, sizeof(rec));
        hit[i] = world_hit(s, &r, &rec, &rc);
        t[i] = hit[i] ? rec.t : 0.0f;
        inst[i] = hit[i] ? rec.inst : -1;
    }
}

/* BVH build: bvh.h:71-175.  qsort here is glibc's merge sort (msort.c): top-down, left run
 * element taken iff cmp <= 0; the comparator returns only -1/+1 (never 0). */
typedef struct { pto_scene *sc; int axis; } sortctx;
static int box_compare(const sortctx *c, int a, int b)
{   /* bvh.h:71-131: (left.min[axis] - right.min[axis] < 0.0) ? -1 : 1 */
    float l = vget(c->sc->insts[a].bbox.mn, c->axis);
    float r = vget(c->sc->insts[b].bbox.mn, c->axis);
    return ((double)(l - r) < 0.0) ? -1 : 1;
}
static void msort(const sortctx *c, int *b, int n, int *tmp)
{
    if (n <= 1) return;
    int n1 = n / 2, n2 = n - n1;
    int *b1 = b, *b2 = b + n1;
    msort(c, b1, n1, tmp);
    msort(c, b2, n2, tmp);
    int *t = tmp;
    while (n1 > 0 && n2 > 0) {
        if (box_compare(c, *b1, *b2) <= 0) { *t++ = *b1++; n1--; }
        else { *t++ = *b2++; n2--; }
    }
    if (n1 > 0) memcpy(t, b1, (size_t)n1 * sizeof(int));
    memcpy(b, tmp, (size_t)(n - n2) * sizeof(int));
}
static int build_node(pto_scene *sc, int *l, int n, int *tmp)
{   /* bvh.h:133-175; returns node index (preorder: parent before children, left before right) */
    int me = sc->nnode++;
    int axis = (int)(3 * mt_double(&sc->rng));
    sortctx c = {sc, axis <= 0 ? 0 : (axis == 1 ? 1 : 2)};
    msort(&c, l, n, tmp);
    int left, right;
    if (n == 1) { left = right = ~l[0]; }
    else if (n == 2) { left = ~l[0]; right = ~l[1]; }
    else {
        left = build_node(sc, l, n / 2, tmp);
        right = build_node(sc, l + n / 2, n - n / 2, tmp);
    }
    aabb bl = left >= 0 ? sc->nodes[left].box : sc->insts[~left].bbox;
    aabb br = right >= 0 ? sc->nodes[right].box : sc->insts[~right].bbox;
    sc->nodes[me].left = left;
    sc->nodes[me].right = right;
    sc->nodes[me].box = surrounding_box(bl, br);
    return me;
}

/* ---- scene construction ------------------------------------------------------------------ */
static rect_t make_rect(float x0, float z0, float x1, float z1, float y, int mat, int plane, int flipped)
{
    rect_t r = {x0, z0, x1, z1, y, plane, !flipped, mat};
    return r;
}

static void perlin_static_init(mt19937 *g, v3 ranvec[256], int perm[3][256]);
static int tex_uses_uv(const pto_scene *sc, int ti)
{   /* does the tree below texture ti reach an image texture (the only kind that reads u, v)? */
    const tex_t *t = &sc->tex[ti];
    if (t->type == PTO_TEX_IMAGE) return 1;
    if (t->type == PTO_TEX_CHECKER) return tex_uses_uv(sc, t->even) || tex_uses_uv(sc, t->odd);
    return 0;
}
pto_scene *pto_scene_create(const pto_material *mats, int nmat, const pto_prim *prims, int nprim,
                            const pto_instance *insts, int ninst, const pto_camera *cam, const float background[3])
{
    return pto_scene_create_textured(mats, nmat, prims, nprim, insts, ninst, cam, background, NULL, 0, NULL, 0, -1);
}
pto_scene *pto_scene_create_textured(const pto_material *mats, int nmat, const pto_prim *prims, int nprim,
                                     const pto_instance *insts, int ninst, const pto_camera *cam,
                                     const float background[3], const pto_texture *tex, int ntex,
                                     const uint8_t *texels, int64_t texel_bytes, int background_texture)
{
    if (ninst < 1 || nprim < 1 || nmat < 1 || ntex < 0 || background_texture >= ntex) return NULL;
    pto_scene *sc = (pto_scene *)calloc(1, sizeof(*sc));
    sc->ntex = ntex;
    sc->background_tex = background_texture < 0 ? -1 : background_texture;
    sc->tex = (tex_t *)calloc((size_t)ntex + 1, sizeof(tex_t));
    sc->texels = (uint8_t *)malloc((size_t)(texel_bytes > 0 ? texel_bytes : 1));
    if (texel_bytes > 0) memcpy(sc->texels, texels, (size_t)texel_bytes);
    for (int i = 0; i < ntex; i++) {
        tex_t *t = &sc->tex[i];
        t->type = tex[i].type;
        t->color = V(tex[i].color[0], tex[i].color[1], tex[i].color[2]);
        t->alpha = tex[i].alpha;
        t->even = tex[i].even; t->odd = tex[i].odd; t->scale = tex[i].scale;
        t->width = tex[i].width; t->height = tex[i].height;
        if (t->type == PTO_TEX_CHECKER && (t->even < 0 || t->even >= i || t->odd < 0 || t->odd >= i)) goto fail;
        if (t->type == PTO_TEX_IMAGE) {
            if (t->width < 1 || t->height < 1 || tex[i].texel_offset < 0 ||
                tex[i].texel_offset + 4 * (int64_t)t->width * t->height > texel_bytes) goto fail;
            t->rgba = sc->texels + tex[i].texel_offset;
        }
        if (t->type < PTO_TEX_CONSTANT || t->type > PTO_TEX_IMAGE) goto fail;
    }
    sc->nmat = nmat; sc->nprim = nprim; sc->ninst = ninst;
    sc->mats = (mat_t *)calloc((size_t)nmat, sizeof(mat_t));
    sc->prims = (prim_t *)calloc((size_t)nprim, sizeof(prim_t));
    sc->insts = (inst_t *)calloc((size_t)ninst, sizeof(inst_t));
    sc->nodes = (node_t *)calloc(2 * (size_t)ninst + 2, sizeof(node_t)) /* a one-leaf range still makes a node (bvh.h:142): up to 2n - 1 nodes */;
    sc->lights = (int *)calloc((size_t)ninst, sizeof(int));
    sc->cam = *cam;
    sc->background = V(background[0], background[1], background[2]);
    for (int i = 0; i < nmat; i++) {
        sc->mats[i].type = mats[i].type;
        sc->mats[i].color = V(mats[i].color[0], mats[i].color[1], mats[i].color[2]);
        sc->mats[i].alpha = mats[i].alpha;
        sc->mats[i].power = mats[i].power;
        sc->mats[i].two_sided = mats[i].two_sided;
        sc->mats[i].tex = mats[i].texture;
        if (mats[i].texture >= ntex) goto fail;
        if (mats[i].texture < 0) sc->mats[i].tex = -1;
        /* an image-textured emitter: the NEE ray of a path that lands on the light lies in the light's plane, rect::hit
           accepts its NaN t (Q8) and image_texture::alpha indexes with NaN u, v (image.h:46) -- the reference crashes */
        if (sc->mats[i].tex >= 0 && sc->mats[i].type == PTO_MAT_DIFFUSE_LIGHT && tex_uses_uv(sc, sc->mats[i].tex)) goto fail;
    }
    for (int i = 0; i < nprim; i++) {
        const pto_prim *p = &prims[i];
        prim_t *q = &sc->prims[i];
        q->type = p->type; q->mat = p->mat;
        if (p->mat < 0 || p->mat >= nmat) goto fail;
        /* sphere::hit / constant_medium::hit leave rec.u, rec.v unset: an image texture there reads indeterminate values */
        if (sc->mats[p->mat].tex >= 0 && tex_uses_uv(sc, sc->mats[p->mat].tex) && p->type != PTO_PRIM_RECT && p->type != PTO_PRIM_BOX) goto fail;
        switch (p->type) {
        case PTO_PRIM_RECT:
            q->rect = make_rect(p->rect[0], p->rect[1], p->rect[2], p->rect[3], p->rect[4], p->mat, p->plane, p->flipped);
            break;
        case PTO_PRIM_BOX: { /* primitive.h:232-240 */
            const float *a = p->p0, *b = p->p1;
            q->sides[0] = make_rect(a[0], a[1], b[0], b[1], a[2], p->mat, PTO_PLANE_XY, 1);
            q->sides[1] = make_rect(a[0], a[1], b[0], b[1], b[2], p->mat, PTO_PLANE_XY, 0);
            q->sides[2] = make_rect(a[1], a[2], b[1], b[2], a[0], p->mat, PTO_PLANE_YZ, 1);
            q->sides[3] = make_rect(a[1], a[2], b[1], b[2], b[0], p->mat, PTO_PLANE_YZ, 0);
            q->sides[4] = make_rect(a[0], a[2], b[0], b[2], a[1], p->mat, PTO_PLANE_XZ, 1);
            q->sides[5] = make_rect(a[0], a[2], b[0], b[2], b[1], p->mat, PTO_PLANE_XZ, 0);
            break;
        }
        case PTO_PRIM_SPHERE:
            q->center = V(p->center[0], p->center[1], p->center[2]);
            q->radius = p->radius;
            break;
        case PTO_PRIM_VOLUME:
            if (p->boundary < 0 || p->boundary >= i || p->phase_mat < 0 || p->phase_mat >= nmat) goto fail;
            q->boundary = p->boundary; q->density = p->density; q->phase_mat = p->phase_mat;
            break;
        default: goto fail;
        }
    }
    for (int i = 0; i < ninst; i++) {
        const pto_instance *p = &insts[i];
        inst_t *q = &sc->insts[i];
        if (p->prim < 0 || p->prim >= nprim) goto fail;
        q->prim = p->prim;
        q->fwd = affine_compose(p->scale, p->rotate, p->translate);
        q->inv = affine_inverse(&q->fwd);
        /* first stream-mode draw slot of this instance's traversal; nvol = slots of all volume instances */
        q->vol_ordinal = -1;
        if (sc->prims[p->prim].type == PTO_PRIM_VOLUME) { q->vol_ordinal = sc->nvol; sc->nvol += volume_draws(sc, &sc->prims[p->prim]); }
        /* instance ctor: bbox of the 8 transformed corners (primitive.h:266-296) */
        aabb pb = prim_bbox(sc, &sc->prims[p->prim]);
        v3 mn = V(FLT_MAX, FLT_MAX, FLT_MAX), mx = V(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++)
                for (int c = 0; c < 2; c++) {
                    float x = a * pb.mx.x + (1 - a) * pb.mn.x;
                    float y = b * pb.mx.y + (1 - b) * pb.mn.y;
                    float z = c * pb.mx.z + (1 - c) * pb.mn.z;
                    v3 t = xf_point(&q->fwd, V(x, y, z));
                    if (t.x > mx.x) mx.x = t.x;
                    if (t.x < mn.x) mn.x = t.x;
                    if (t.y > mx.y) mx.y = t.y;
                    if (t.y < mn.y) mn.y = t.y;
                    if (t.z > mx.z) mx.z = t.z;
                    if (t.z < mn.z) mn.z = t.z;
                }
        q->bbox.mn = mn; q->bbox.mx = mx;
        if (p->is_light) sc->lights[sc->nlight++] = i;
    }
    mt_seed(&sc->rng, 5489u);
    perlin_static_init(&sc->rng, sc->ranvec, sc->perm); /* consumes the PERLIN_STATIC_DRAWS values */
    {
        int *order = (int *)malloc((size_t)ninst * sizeof(int));
        int *tmp = (int *)malloc((size_t)ninst * sizeof(int));
        for (int i = 0; i < ninst; i++) order[i] = i;
        sc->nnode = 0;
        build_node(sc, order, ninst, tmp);
        free(order); free(tmp);
    }
    return sc;
fail:
    pto_scene_destroy(sc);
    return NULL;
}
void pto_scene_destroy(pto_scene *s)
{
    if (!s) return;
    free(s->mats); free(s->prims); free(s->insts); free(s->nodes); free(s->lights);
    free(s->tex); free(s->texels);
    free(s);
}
int pto_scene_num_instances(const pto_scene *s) { return s->ninst; }
void pto_scene_instance_tables(const pto_scene *s, int i, float fwd[12], float inv[12], float bbox[6])
{
    memcpy(fwd, s->insts[i].fwd.m, 12 * sizeof(float));
    memcpy(inv, s->insts[i].inv.m, 12 * sizeof(float));
    const aabb *b = &s->insts[i].bbox;
    bbox[0] = b->mn.x; bbox[1] = b->mn.y; bbox[2] = b->mn.z; bbox[3] = b->mx.x; bbox[4] = b->mx.y; bbox[5] = b->mx.z;
}
int pto_scene_num_nodes(const pto_scene *s) { return s->nnode; }
void pto_scene_node(const pto_scene *s, int n, float bbox[6], int32_t *left, int32_t *right)
{
    const aabb *b = &s->nodes[n].box;
    bbox[0] = b->mn.x; bbox[1] = b->mn.y; bbox[2] = b->mn.z; bbox[3] = b->mx.x; bbox[4] = b->mx.y; bbox[5] = b->mx.z;
    *left = s->nodes[n].left; *right = s->nodes[n].right;
}
int pto_scene_num_lights(const pto_scene *s) { return s->nlight; }
int pto_scene_light(const pto_scene *s, int k) { return s->lights[k]; }
double pto_scene_next_random(pto_scene *s) { return mt_double(&s->rng); }
void pto_rng_after_static_init(int n, double *out)
{
    mt19937 g;
    mt_seed(&g, 5489u);
    for (int i = 0; i < PERLIN_STATIC_DRAWS; i++) (void)mt_double(&g);
    for (int i = 0; i < n; i++) out[i] = mt_double(&g);
}

/* ---- camera (camera.h:9-47, main.cpp:86-104) ---------------------------------------------- */
typedef struct { v3 origin, llc, horizontal, vertical, u, v, w; float lens_radius; } camera_t;

static camera_t make_camera(const pto_camera *c, int width, int height)
{
    camera_t cam;
    float aspect = (float)width / (float)height;                 /* main.cpp:148 */
    v3 lookfrom = V(c->look_from[0], c->look_from[1], c->look_from[2]);
    v3 lookat = V(c->look_at[0], c->look_at[1], c->look_at[2]);
    v3 vup = V(0, 1, 0);
    cam.lens_radius = c->aperture / 2;
    float theta = (float)((double)c->fov * M_PI / 180);
    float half_height = tanf(theta / 2);
    float half_width = aspect * half_height;
    cam.origin = lookfrom;
    cam.w = vunit(vsub(lookfrom, lookat));
    cam.u = vunit(vcross(vup, cam.w));
    cam.v = vcross(cam.w, cam.u);
    float fd = c->dist_to_focus;
    cam.llc = vsub(vsub(vsub(cam.origin, vscale(half_width * fd, cam.u)), vscale(half_height * fd, cam.v)), vscale(fd, cam.w));
    cam.horizontal = vscale(2 * half_width * fd, cam.u);
    cam.vertical = vscale(2 * half_height * fd, cam.v);
    return cam;
}
void pto_scene_camera(const pto_scene *s, int width, int height, float out[22])
{
    camera_t c = make_camera(&s->cam, width, height);
    const v3 *vs[7] = {&c.origin, &c.llc, &c.horizontal, &c.vertical, &c.u, &c.v, &c.w};
    for (int i = 0; i < 7; i++) { out[3 * i] = vs[i]->x; out[3 * i + 1] = vs[i]->y; out[3 * i + 2] = vs[i]->z; }
    out[21] = c.lens_radius;
}

/* stream-mode dimension layout ------------------------------------------------------------
 *   0,1   pixel jitter u,v            (renderer.h:648-649)
 *   2,3   lens disk angle, radius     (camera.h:41; only drawn when lens_radius != 0)
 *   8 + b*D + ...   bounce b, with NV = volume instances, L = light_samples, D = NV + L*(3+NV) + 4:
 *     [0, NV)                         free-flight draw of volume v on the extension ray (volume.h:70)
 *     NV + k*(3+NV) + 0               light pick of light sample k (world.h:33)
 *     NV + k*(3+NV) + 1, 2            the two light-surface draws, in the reference's draw order
 *     NV + k*(3+NV) + 3 + v           free-flight draw of volume v on shadow ray k
 *     NV + L*(3+NV) + 0,1,2           material generate() draws (random.h:17-44)
 *     NV + L*(3+NV) + 3               russian roulette (integrator.h:289)
 * The camera time draw (camera.h:43) and the trace-probability draw (renderer.h:652) never
 * influence the radiance and are not drawn in stream mode. */
#define DIM_JITTER_U 0u
#define DIM_JITTER_V 1u
#define DIM_LENS 2u
#define DIM_BOUNCE0 8u
int pto_stream_dims_per_bounce(const pto_scene *s, int light_samples)
{
    return s->nvol + light_samples * (3 + s->nvol) + 4;
}

static ray_t camera_get_ray(const camera_t *cam, float s, float t, rngctx *rc)
{   /* camera.h:38-47 + random.h:27-34 */
    v3 offset;
    if (rc->mode == PTO_MODE_MT) {
        float u = (float)(rnd(rc, 0) * (2 * M_PI));
        float v = powf((float)rnd(rc, 0), (float)(1.0 / 2.0));
        v3 p = V(cosf(u) * v, sinf(u) * v, 0);
        v3 rd = vscale(cam->lens_radius, p);
        offset = vadd(vscale(rd.x, cam->u), vscale(rd.y, cam->v));
        (void)rnd(rc, 0); /* time = time0 + random_double()*(time1-time0): unused downstream */
    } else if (cam->lens_radius != 0.0f) {
        float su, cu;
        ptm_sincos_2pi((float)rnd(rc, DIM_LENS), &su, &cu);
        float v = sqrtf((float)rnd(rc, DIM_LENS + 1));
        v3 rd = vscale(cam->lens_radius, V(cu * v, su * v, 0));
        offset = vadd(vscale(rd.x, cam->u), vscale(rd.y, cam->v));
    } else {
        offset = V(0, 0, 0);
    }
    ray_t r;
    r.A = vadd(cam->origin, offset);
    r.B = vsub(vsub(vadd(vadd(cam->llc, vscale(s, cam->horizontal)), vscale(t, cam->vertical)), cam->origin), offset);
    return r;
}

/* ---- materials / pdfs (material.h, pdf.h, random.h) --------------------------------------- */
static inline float power_heuristic(float fPdf, float gPdf, int mode)
{   /* helpers.h:138-144 with nf = ng = 1 */
    float f = 1 * fPdf, g = 1 * gPdf;
    if (mode == PTO_MODE_MT) {
        float fp = powf(f, 2.0f);
        return fp / (fp + powf(g, 2.0f));
    }
    float fp = f * f;
    return fp / (fp + g * g);
}
static inline float cosine_pdf_value(v3 normal, v3 direction)
{   /* pdf.h:18-29 with uvw.w() = unit_vector(normal) */
    float cosine = vdot(vunit(direction), vunit(normal));
    if (cosine > 0) return (float)((double)cosine / M_PI);
    return 0;
}
static float material_value(const mat_t *m, v3 normal, v3 direction)
{
    switch (m->type) {
    case PTO_MAT_LAMBERTIAN:
    case PTO_MAT_METAL: return cosine_pdf_value(normal, direction);   /* material.h:66-69, 105-108 */
    case PTO_MAT_ISOTROPIC: return (float)(1 / (4 * M_PI));            /* pdf.h:41-44 */
    default: return 0;                                                /* void_pdf pdf.h:72-75 */
    }
}
static v3 random_in_unit_sphere(rngctx *rc, uint32_t dim)
{   /* random.h:17-24 */
    if (rc->mode == PTO_MODE_MT) {
        float u = (float)(rnd(rc, 0) * (2 * M_PI));
        float v = (float)acos(2 * rnd(rc, 0) - 1);
        float w = powf((float)rnd(rc, 0), (float)(1.0 / 3.0));
        return V(cosf(u) * sinf(v) * w, cosf(v) * w, sinf(u) * sinf(v) * w);
    }
    float su, cu;
    ptm_sincos_2pi((float)rnd(rc, dim + 0), &su, &cu);
    float cv = (float)(2 * rnd(rc, dim + 1) - 1); /* cos(acos(x)) = x */
    float sv2 = 1.0f - cv * cv;
    float sv = sqrtf(sv2 > 0.0f ? sv2 : 0.0f);
    float w = ptm_cbrtf((float)rnd(rc, dim + 2));
    return V(cu * sv * w, cv * w, su * sv * w);
}
static v3 material_generate(const mat_t *m, v3 normal, rngctx *rc, uint32_t dim)
{
    if (m->type == PTO_MAT_LAMBERTIAN || m->type == PTO_MAT_METAL) {
        /* cosine_pdf(rec.normal).generate(): pdf.h:30-33, random.h:36-44 */
        onb uvw = onb_from_w(normal);
        float r1 = (float)rnd(rc, dim + 0);
        float r2 = (float)rnd(rc, dim + 1);
        float z = sqrtf(1 - r2);
        float s, c;
        if (rc->mode == PTO_MODE_MT) {
            float phi = (float)(2 * M_PI * (double)r1);
            c = cosf(phi); s = sinf(phi);
        } else {
            ptm_sincos_2pi(r1, &s, &c);
        }
        float x = c * sqrtf(r2);
        float y = s * sqrtf(r2);
        return onb_local(&uvw, V(x, y, z));
    }
    if (m->type == PTO_MAT_DIELECTRIC) {
        /* dielectric::generate material.h:125-166: reflect / refract / schlick, then ONE draw picks between them.  Its
           value() is void_pdf = 0 (material.h:167-170), so NEEIterative breaks at scatter_pdf_s < 1e-7
           (integrator.h:301-304) before the direction is ever used: only the draw is observable (MT stream position). */
        (void)rnd(rc, dim + 0);
        return V(0, 0, 0);
    }
    return random_in_unit_sphere(rc, dim); /* isotropic material.h:267-270 (and void_pdf) */
}
/* ---- textures (texture.h, image.h) -------------------------------------------------------- */
/* perlin_generate / perlin_generate_perm / permute (texture.h:76-110) in static-initialisation order (:180-183):
 * ranvec (256 x 3 draws), then perm_x, perm_y, perm_z (255 draws each) = the 1533 draws before main(). */
static void perlin_static_init(mt19937 *g, v3 ranvec[256], int perm[3][256])
{
    for (int i = 0; i < 256; i++) {
        double xr = 2 * mt_double(g) - 1;
        double yr = 2 * mt_double(g) - 1;
        double zr = 2 * mt_double(g) - 1;
        ranvec[i] = vunit(V((float)xr, (float)yr, (float)zr));
    }
    for (int k = 0; k < 3; k++) {
        int *p = perm[k];
        for (int i = 0; i < 256; i++) p[i] = i;
        for (int i = 255; i > 0; i--) {
            int target = (int)(mt_double(g) * (i + 1));
            int tmp = p[i];
            p[i] = p[target];
            p[target] = tmp;
        }
    }
}
void pto_perlin_tables(float ranvec[768], int32_t perm[768])
{
    mt19937 g;
    v3 rv[256];
    int pm[3][256];
    mt_seed(&g, 5489u);
    perlin_static_init(&g, rv, pm);
    for (int i = 0; i < 256; i++) { ranvec[3 * i] = rv[i].x; ranvec[3 * i + 1] = rv[i].y; ranvec[3 * i + 2] = rv[i].z; }
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 256; i++) perm[k * 256 + i] = pm[k][i];
}
/* perlin::noise (texture.h:134-160) + perlin_interp (:111-131): the Hermite smoothing is applied in noise() AND again in
 * perlin_interp(), whose weight vector also uses the already-smoothed u, v, w -- restated as written. */
static float perlin_noise(const pto_scene *sc, v3 p)
{
    float u = p.x - floorf(p.x);
    float v = p.y - floorf(p.y);
    float w = p.z - floorf(p.z);
    u = u * u * (3 - 2 * u);
    v = v * v * (3 - 2 * v);
    w = w * w * (3 - 2 * w);
    int i = (int)floorf(p.x);
    int j = (int)floorf(p.y);
    int k = (int)floorf(p.z);
    float uu = u * u * (3 - 2 * u);
    float vv = v * v * (3 - 2 * v);
    float ww = w * w * (3 - 2 * w);
    float accum = 0;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                v3 c = sc->ranvec[sc->perm[0][(i + di) & 255] ^ sc->perm[1][(j + dj) & 255] ^ sc->perm[2][(k + dk) & 255]];
                v3 weight_v = V(u - di, v - dj, w - dk);
                accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) *
                         vdot(c, weight_v);
            }
    return accum;
}
/* texture::value(u, v, p) and ::alpha(u, v, p).  checker_texture picks the same child for both (texture.h:43-68). */
static void tex_eval(const pto_scene *sc, int ti, float u, float v, v3 p, int mode, v3 *color, float *alpha)
{
    for (;;) {
        const tex_t *t = &sc->tex[ti];
        if (t->type == PTO_TEX_CHECKER) { /* sines(): texture.h:70-73, sin(float) = sinf */
            float sx, sy, sz;
            if (mode == PTO_MODE_MT) { sx = sinf(t->scale * p.x); sy = sinf(t->scale * p.y); sz = sinf(t->scale * p.z); }
            else { sx = ptm_sinf(t->scale * p.x); sy = ptm_sinf(t->scale * p.y); sz = ptm_sinf(t->scale * p.z); }
            ti = (sx * sy * sz > 0) ? t->odd : t->even;
            continue;
        }
        if (t->type == PTO_TEX_PERLIN) { /* noise_texture::value texture.h:190-193; alpha(): base class, 1.0 */
            float n = perlin_noise(sc, vscale(t->scale, p));
            *color = V(1 * n, 1 * n, 1 * n);
            *alpha = 1.0f;
            return;
        }
        if (t->type == PTO_TEX_IMAGE) { /* image.h:15-49 */
            v -= (int)v;
            if (v < 0) v += 1;
            u -= (int)u;
            if (u < 0) u += 1;
            int y = (int)(v * t->height);
            int x = (int)(u * t->width);
            if (y > t->height - 1) y = t->height - 1; /* u or v == 1.0f after the wrap: the reference reads out of bounds */
            if (x > t->width - 1) x = t->width - 1;
            const uint8_t *px = t->rgba + 4 * ((size_t)y * t->width + x);
            *color = V((float)(px[0] / 255.0), (float)(px[1] / 255.0), (float)(px[2] / 255.0)); /* image.h:58-62 */
            *alpha = (float)(px[3] / 255.0);
            return;
        }
        *color = t->color;
        *alpha = t->alpha;
        return;
    }
}
void pto_texture_eval(const pto_scene *s, int ti, int mode, float u, float v, const float p[3], float out[4])
{
    v3 c = V(0, 0, 0);
    float a = 0;
    if (ti >= 0 && ti < s->ntex) tex_eval(s, ti, u, v, V(p[0], p[1], p[2]), mode, &c, &a);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = a;
}
/* albedo->value(rec.u, rec.v, rec.p) of a lambertian / isotropic (material.h:44, 259) */
static v3 material_albedo(const pto_scene *sc, const mat_t *m, const hitrec *rec, int mode)
{
    if (m->tex < 0) return m->color;
    v3 c; float a;
    tex_eval(sc, m->tex, rec->u, rec->v, rec->p, mode, &c, &a);
    return c;
}
static v3 material_emitted(const pto_scene *sc, const mat_t *m, v3 ray_dir, const hitrec *rec, int mode)
{   /* material.h:211-229; everything else material.h:21-24 (isotropic's 3-arg emitted never overrides) */
    if (m->type != PTO_MAT_DIFFUSE_LIGHT) return V(0, 0, 0);
    int aligned = vdot(rec->normal, ray_dir) > 0;
    if (!aligned || m->two_sided) {
        v3 c = m->color; float a = m->alpha;
        if (m->tex >= 0) tex_eval(sc, m->tex, rec->u, rec->v, rec->p, mode, &c, &a);   /* emitted(r, rec, rec.u, rec.v, rec.p) */
        return vscale(a, vscale(m->power, c));
    }
    return V(0, 0, 0);
}

/* ---- NEEIterative::color (integrator.h:176-339) ------------------------------------------- */
static v3 integrator_color(const pto_scene *sc, const pto_config *cfg, ray_t r, rngctx *rc, pto_counters *ctr)
{
    hitrec rec;
    v3 sum = V(0, 0, 0);
    v3 attenuation = V(0, 0, 0);
    v3 hit_emission;
    float last_bsdf_pdf = -1;
    v3 beta = V(1.0f, 1.0f, 1.0f);
    const int mode = rc->mode;
    const uint32_t L = (uint32_t)cfg->light_samples, NV = (uint32_t)sc->nvol;
    const uint32_t D = NV + L * (3 + NV) + 4;
    int i;
    for (i = 0; i < cfg->max_bounces; i++) {
        const uint32_t base = DIM_BOUNCE0 + (uint32_t)i * D;
        ctr->rays++; ctr->ext_rays++;
        rc->vol_dim_base = base;
        if (world_hit(sc, &r, &rec, rc)) {
            ctr->ext_hits++;
            const mat_t *m = &sc->mats[rec.mat];
            /* scatter(): lambertian material.h:39-53, metal :90-98, dielectric :118-124,
               diffuse_light :187-191 (leaves attenuation stale), isotropic :252-261 */
            int did_scatter = 1;
            switch (m->type) {
            case PTO_MAT_LAMBERTIAN:
                if (vdot(r.B, rec.normal) < 0) attenuation = vdivf(material_albedo(sc, m, &rec, mode), (float)M_PI);
                else attenuation = V(0, 0, 0);
                break;
            case PTO_MAT_METAL: attenuation = vdivf(m->color, (float)M_PI); break;
            case PTO_MAT_DIELECTRIC: attenuation = V(1.0f, 1.0f, 1.0f); break;
            case PTO_MAT_DIFFUSE_LIGHT: did_scatter = 0; break;
            case PTO_MAT_ISOTROPIC: attenuation = material_albedo(sc, m, &rec, mode); break;
            }
            float cos_i = fabsf(vdot(vunit(r.B), vunit(rec.normal)));
            hit_emission = material_emitted(sc, m, r.B, &rec, mode);
            if (vsqlen(hit_emission) > 0.000001) {
                if (last_bsdf_pdf <= 0) {
                    sum = vadd(sum, vmul(beta, hit_emission));
                } else {
                    /* hittable_pdf(rec.primitive, r.origin()).value(rec.p): a POSITION as direction (Q4) */
                    float lp = instance_pdf_value(sc, rec.inst, r.A, rec.p, mode);
                    float weight = power_heuristic(last_bsdf_pdf, lp, mode);
                    sum = vadd(sum, vscale(weight, vmul(beta, hit_emission)));
                }
            }
            v3 light_contribution = V(0, 0, 0);
            for (uint32_t k = 0; k < L; k++) {
                const uint32_t kb = base + NV + k * (3 + NV);
                int idx = (int)(rnd(rc, kb + 0) * (double)(size_t)sc->nlight); /* world.h:31-35 */
                int light = sc->lights[idx];
                float pick_pdf = (float)sc->nlight;
                ray_t light_ray;
                light_ray.A = rec.p;
                light_ray.B = instance_random(sc, light, rec.p, rc, kb + 1);
                float cos_l = vdot(vunit(light_ray.B), vunit(rec.normal));
                float light_pdf_l = instance_pdf_value(sc, light, rec.p, light_ray.B, mode);
                float scatter_pdf_l = material_value(m, rec.normal, light_ray.B);
                float weight_l = power_heuristic(light_pdf_l, scatter_pdf_l, mode);
                hitrec lrec;
                rc->vol_dim_base = kb + 3;
                int did_light_hit = world_hit(sc, &light_ray, &lrec, rc);
                ctr->rays++; ctr->shadow_rays++;
                if (did_light_hit && (double)vlen(attenuation) > 0.0001) {
                    v3 le = material_emitted(sc, &sc->mats[lrec.mat], light_ray.B, &lrec, mode);
                    float dropoff = (mode == PTO_MODE_MT) ? (float)fmax((double)cos_l, 0.0) : (cos_l > 0.0f ? cos_l : 0.0f);
                    /* attenuation * beta * weight_l / light_pdf_l * dropoff * light_emission / pick_pdf */
                    v3 c = vmul(attenuation, beta);
                    c = vscale(weight_l, c);
                    c = vdivf(c, light_pdf_l);
                    c = vscale(dropoff, c);
                    c = vmul(c, le);
                    c = vdivf(c, pick_pdf);
                    if (!v_is_nan(c)) light_contribution = vadd(light_contribution, c);
                }
            }
            sum = vadd(sum, vdivf(light_contribution, (float)cfg->light_samples));
            if (did_scatter) {
                const uint32_t gb = base + NV + L * (3 + NV);
                ray_t scattered;
                scattered.A = vadd(rec.p, vscale(cfg->normal_offset, rec.normal));
                scattered.B = material_generate(m, rec.normal, rc, gb);
                float scatter_pdf_s = material_value(m, rec.normal, scattered.B);
                float pin = (beta.y < beta.z) ? beta.z : beta.y; /* std::max(a,b) = (a < b) ? b : a */
                float p = (beta.x < pin) ? pin : beta.x;
                if (cfg->russian_roulette && p <= 1 && 0.001 < (double)p) {
                    if (rnd(rc, gb + 3) > (double)p) { ctr->term_rr++; break; }
                    beta = vscale_assign(beta, 1 / p);
                }
                if (!cfg->only_direct) {
                    if ((double)scatter_pdf_s < 0.0000001) { ctr->term_pdf++; break; }
                    beta = vmul(beta, vdivf(vscale(fabsf(cos_i), attenuation), scatter_pdf_s));
                    last_bsdf_pdf = scatter_pdf_s;
                    r = scattered;
                } else {
                    break;
                }
            } else {
                sum = vadd(sum, vmul(beta, hit_emission)); /* second addition (Q3) */
                ctr->term_emitter++;
                break;
            }
        } else {
            /* integrator.h:325-336: world->value(u, v, unit_direction); a constant background ignores all three
               (world.h:27-30 -> texture.h:21-24).  TAU is "2 * M_PI" unparenthesised (random.h:7), so u = ((pi + atan2) / 2) * pi */
            v3 bg = sc->background;
            if (sc->background_tex >= 0) {
                v3 ud = vunit(r.B);
                float eu, ev, a;
                if (mode == PTO_MODE_MT) {
                    eu = (float)((M_PI + (double)atan2f(ud.y, ud.x)) / 2 * M_PI);
                    ev = (float)((double)acosf(ud.z) / M_PI);
                } else {
                    eu = (float)((M_PI + (double)ptm_atan2f(ud.y, ud.x)) / 2 * M_PI);
                    ev = (float)((double)ptm_acosf(ud.z) / M_PI);
                }
                tex_eval(sc, sc->background_tex, eu, ev, ud, mode, &bg, &a);
            }
            sum = vadd(sum, vmul(beta, bg));
            ctr->term_miss++;
            break;
        }
    }
    if (i == cfg->max_bounces) ctr->term_bounce_limit++;
    return sum;
}

/* ---- NaiveSpiral (queue.h:68-127) --------------------------------------------------------- */
typedef struct { int x0, y0, x1, y1; } tile_t;
static int spiral_tiles(int width, int height, int bw, int bh, tile_t **out)
{
    int tw = (int)ceilf((float)width / bw), th = (int)ceilf((float)height / bh);
    tile_t *tiles = (tile_t *)malloc((size_t)tw * th * sizeof(tile_t));
    int n = 0, radius = 1;
    int x = (tw % 2 == 0) ? (tw / 2 - 1) : (tw / 2);
    int y = (th % 2 == 0) ? (th / 2 - 1) : (th / 2);
    int furthest = tw > th ? tw : th;
    int dx = 1, dy = 0, count = radius;
    while (radius <= furthest) {
        if (x >= 0 && y >= 0 && x < tw && y < th) {
            tile_t t = {x * bw, y * bh, 0, 0};
            t.x1 = t.x0 + bw < width ? t.x0 + bw : width;
            t.y1 = t.y0 + bh < height ? t.y0 + bh : height;
            tiles[n++] = t;
        }
        x += dx; y += dy; count--;
        if (count <= 0) {
            if (dx == 0 && dy == 1) { dx = -1; dy = 0; radius++; }
            else if (dx == 1 && dy == 0) { dx = 0; dy = 1; }
            else if (dx == -1 && dy == 0) { dx = 0; dy = -1; }
            else if (dx == 0 && dy == -1) { dx = 1; dy = 0; radius++; }
            count = radius;
        }
    }
    *out = tiles;
    return n;
}

/* ---- Tiled::compute in the reference's order (renderer.h:626-691), MT mode --------------- */
typedef void (*sample_cb)(void *user, int i, int j, float u, float v, const ray_t *r, v3 col, uint64_t rays, const hitrec *h, int hit);

static void run_mt(pto_scene *sc, const pto_config *cfg, float *fb, pto_counters *ctr, int max_samples, int hits_only,
                   float *out)
{
    camera_t cam = make_camera(&sc->cam, cfg->width, cfg->height);
    rngctx rc;
    memset(&rc, 0, sizeof(rc));
    rc.mode = PTO_MODE_MT;
    rc.mt = &sc->rng;
    tile_t *tiles;
    int ntiles = spiral_tiles(cfg->width, cfg->height, cfg->block_w, cfg->block_h, &tiles);
    int done = 0;
    for (int ti = 0; ti < ntiles; ti++) {
        tile_t t = tiles[ti];
        for (int s = 0; s < cfg->samples; s++)
            for (int j = t.y1 - 1; j >= t.y0; j--)
                for (int i = t.x0; i < t.x1; i++) {
                    if (max_samples >= 0 && done >= max_samples) goto out;
                    float u = (float)((double)i + rnd(&rc, 0)) / (float)cfg->width;
                    float v = (float)((double)j + rnd(&rc, 0)) / (float)cfg->height;
                    ray_t r = camera_get_ray(&cam, u, v, &rc);
                    (void)rnd(&rc, 0); /* random_double() < trace_probability (renderer.h:652) */
                    if (hits_only) {
                        hitrec h;
                        int hit = world_hit(sc, &r, &h, &rc);
                        float *o = out + (size_t)done * 9;
                        o[0] = hit ? 1.f : 0.f;
                        for (int k = 1; k < 8; k++) o[k] = 0;
                        o[8] = -1;
                        if (hit) {
                            o[1] = h.t; o[2] = h.p.x; o[3] = h.p.y; o[4] = h.p.z;
                            o[5] = h.normal.x; o[6] = h.normal.y; o[7] = h.normal.z; o[8] = (float)h.inst;
                        }
                    } else {
                        pto_counters c1;
                        memset(&c1, 0, sizeof(c1));
                        ray_t r0 = r;
                        v3 col = de_nan(integrator_color(sc, cfg, r, &rc, ctr ? ctr : &c1));
                        if (fb) {
                            float *px = fb + ((size_t)j * cfg->width + i) * 3;
                            px[0] += col.x; px[1] += col.y; px[2] += col.z;
                        }
                        if (out) {
                            float *o = out + (size_t)done * 13;
                            o[0] = u; o[1] = v;
                            o[2] = r0.A.x; o[3] = r0.A.y; o[4] = r0.A.z; o[5] = r0.B.x; o[6] = r0.B.y; o[7] = r0.B.z;
                            o[8] = 0; /* time: not modelled */
                            o[9] = col.x; o[10] = col.y; o[11] = col.z;
                            o[12] = (float)c1.rays;
                        }
                    }
                    done++;
                }
    }
out:
    free(tiles);
}
void pto_render_mt(pto_scene *s, const pto_config *cfg, float *fb, pto_counters *ctr)
{
    memset(ctr, 0, sizeof(*ctr));
    run_mt(s, cfg, fb, ctr, -1, 0, NULL);
}
void pto_samples_mt(pto_scene *s, const pto_config *cfg, int n, float *out) { run_mt(s, cfg, NULL, NULL, n, 0, out); }
void pto_hits_mt(pto_scene *s, const pto_config *cfg, int n, float *out) { run_mt(s, cfg, NULL, NULL, n, 1, out); }

/* ---- stream mode --------------------------------------------------------------------------- */
static v3 stream_sample(const pto_scene *sc, const pto_config *cfg, const camera_t *cam, uint32_t seed, int i, int j,
                        int s, pto_counters *ctr)
{
    rngctx rc;
    memset(&rc, 0, sizeof(rc));
    rc.mode = PTO_MODE_STREAM;
    rc.key = stream_make_key(seed, (uint32_t)(j * cfg->width + i), (uint32_t)s);
    float u = (float)((double)i + rnd(&rc, DIM_JITTER_U)) / (float)cfg->width;
    float v = (float)((double)j + rnd(&rc, DIM_JITTER_V)) / (float)cfg->height;
    ray_t r = camera_get_ray(cam, u, v, &rc);
    return de_nan(integrator_color(sc, cfg, r, &rc, ctr));
}
void pto_sample_stream(const pto_scene *s, const pto_config *cfg, uint32_t seed, int i, int j, int sample, float rgb[3],
                       pto_counters *ctr)
{
    camera_t cam = make_camera(&s->cam, cfg->width, cfg->height);
    pto_counters c;
    memset(&c, 0, sizeof(c));
    v3 col = stream_sample(s, cfg, &cam, seed, i, j, sample, &c);
    rgb[0] = col.x; rgb[1] = col.y; rgb[2] = col.z;
    if (ctr) *ctr = c;
}
typedef struct {
    const pto_scene *sc; const pto_config *cfg; camera_t cam; uint32_t seed;
    int x0, y0, x1, y1, s0, s1; float *fb;
    int next_row; pthread_mutex_t mu; pto_counters total;
} stream_job;

static void ctr_add(pto_counters *a, const pto_counters *b)
{
    a->rays += b->rays; a->ext_rays += b->ext_rays; a->ext_hits += b->ext_hits; a->shadow_rays += b->shadow_rays;
    a->term_miss += b->term_miss; a->term_rr += b->term_rr; a->term_emitter += b->term_emitter;
    a->term_pdf += b->term_pdf; a->term_bounce_limit += b->term_bounce_limit;
}
static void *stream_worker(void *arg)
{
    stream_job *jb = (stream_job *)arg;
    pto_counters local;
    memset(&local, 0, sizeof(local));
    for (;;) {
        pthread_mutex_lock(&jb->mu);
        int j = jb->next_row++;
        pthread_mutex_unlock(&jb->mu);
        if (j >= jb->y1) break;
        for (int i = jb->x0; i < jb->x1; i++) {
            float *px = jb->fb + ((size_t)j * jb->cfg->width + i) * 3;
            for (int s = jb->s0; s < jb->s1; s++) { /* sample order = accumulation order */
                v3 col = stream_sample(jb->sc, jb->cfg, &jb->cam, jb->seed, i, j, s, &local);
                px[0] += col.x; px[1] += col.y; px[2] += col.z;
            }
        }
    }
    pthread_mutex_lock(&jb->mu);
    ctr_add(&jb->total, &local);
    pthread_mutex_unlock(&jb->mu);
    return NULL;
}
void pto_render_stream(const pto_scene *s, const pto_config *cfg, uint32_t seed, int x0, int y0, int x1, int y1, int s0,
                       int s1, int nthreads, float *fb, pto_counters *ctr)
{
    stream_job jb;
    memset(&jb, 0, sizeof(jb));
    jb.sc = s; jb.cfg = cfg; jb.cam = make_camera(&s->cam, cfg->width, cfg->height); jb.seed = seed;
    jb.x0 = x0; jb.y0 = y0; jb.x1 = x1; jb.y1 = y1; jb.s0 = s0; jb.s1 = s1; jb.fb = fb; jb.next_row = y0;
    pthread_mutex_init(&jb.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, stream_worker, &jb);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&jb.mu);
    if (ctr) *ctr = jb.total;
}
