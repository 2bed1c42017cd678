// C-ABI implementation (include/pathtrace_hip.h): device context, scene upload, batch scheduling.
// Compiled with hipcc as host code.  No CPU rendering path exists here: every entry point that produces
// pixels launches the gfx950 kernels in pt_kernels.hip or fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <deque>
#include <vector>

#include "../../../include/pathtrace_hip.h"
#include "pt_device.h"
#include "pt_spec.h"

namespace ptd {
void launch_generate(const DScene &S, const DStreams &st, const DBatch &b, hipStream_t s);
void launch_extend(const DScene &S, const DStreams &st, const DBatch &b, int qi, int bounce, hipStream_t s, SpecJob *spec);
void launch_shade(const DScene &S, const DStreams &st, const DBatch &b, int qi, int bounce, hipStream_t s);
void launch_connect(const DScene &S, const DStreams &st, const DBatch &b, int bounce, hipStream_t s, SpecJob *spec);
void launch_accumulate(const DScene &S, const DStreams &st, const DBatch &b, hipStream_t s);
void launch_trace(const DScene &S, const DStreams &st, long long n, int nr, const float *org, const float *dir, uint32_t k0, uint32_t k1,
                  uint32_t vol_dim, float *t_out, int *id_out, hipStream_t s, SpecJob *spec);
void launch_tally(const DScene &S, const DStreams &st, const DBatch &b, int qi, int bounce, unsigned long long *cost, hipStream_t s);
int launch_fuses_generate();
int launch_qmax_words();
int launch_grid_max();
}  // namespace ptd

using namespace ptd;

static thread_local std::string g_err;
// An environment variable that holds a comma-separated list of tokens (PATHTRACE_HIP_PLAN, _SPEC, _TRAVERSAL, _SHADE: the table in
// include/pathtrace_hip.h): is `token` one of them?  `value` (optional) receives what follows "token=".
static bool env_token(const char *var, const char *token, std::string *value = nullptr)
{
    const char *e = getenv(var);
    if (!e) return false;
    const size_t tl = strlen(token);
    for (const char *p = e; *p;) {
        const char *q = strchr(p, ',');
        const size_t n = q ? (size_t)(q - p) : strlen(p);
        if (n >= tl && !strncmp(p, token, tl) && (n == tl || p[tl] == '=')) {
            if (value) *value = (n > tl) ? std::string(p + tl + 1, n - tl - 1) : std::string();
            return true;
        }
        p += n + (q ? 1 : 0);
    }
    return false;
}
static void set_err(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
#define HIP_TRY(x)                                                                         \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            set_err("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                     \
        }                                                                                  \
    } while (0)

extern "C" const char *pt_last_error(void) { return g_err.c_str(); }
// Segments stop merging at this many: every wave of k_shade reserves its output ranges with returning atomics on the output
// segment's two counters, and with a handful of segments left (32 at bounce 9 of a 66 M-path batch) those serialise: k_shade's
// launches of bounces 7 - 9 took 228 / 204 / 219 us, with 507 segments kept 174 / 132 / 108.
// (A build-time constant: -DPT_MERGE_MIN_SEGMENTS=... for measurements; floors of 256 - 1024 are equal, 0 and 4096 worse.)
#ifndef PT_MERGE_MIN_SEGMENTS
#define PT_MERGE_MIN_SEGMENTS 512
#endif
void pth_set_error(const std::string &m) { g_err = m; }   // used by the host front end (pt_host.cpp)
extern "C" int pt_abi_version(void) { return PT_ABI_VERSION; }
extern "C" int pt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

struct TimedLaunch { int kind; hipEvent_t a, b; };

// A lane = one HIP stream + one private set of wavefront streams.  Consecutive batches go to consecutive lanes, so
// the low-occupancy late bounces of one batch overlap the wide early bounces of the next (the kernels of one batch
// are a dependent chain; the chip is only full when several chains are in flight).
#define PT_MAX_LANES 4
struct Lane {
    hipStream_t stream = nullptr;
    DStreams st{};
    hipEvent_t acc_done = nullptr;   // recorded after this lane's last k_accumulate (framebuffer order across lanes)
};

struct pt_ctx {
    int device = 0;
    pt_config cfg{};
    DScene S{};
    DStreams st{};
    int64_t P = 0;       // path slots a batch may use
    int64_t P_phys = 0;  // slots allocated per stream: P + room for a batch cut into segments of seg_cap + 256 (render_group)
    int seg_cap = 4096;
    int n_seg_max = 0;
    std::vector<void *> allocs;
    std::vector<void *> stream_allocs;   // the P-dependent wavefront streams of all lanes (size_streams: freed and re-made when they grow)
    // the launch plan (include/pathtrace_hip.h, ABI v6)
    bool auto_size = false;          // max_paths_in_flight = 0: the library sizes the streams from the render calls and the free HBM
    pt_plan plan{};
    // wall time of the last render call: entry -> the host function behind its last batch (pt_render_seconds, pt_wait_for)
    struct Clock {
        std::mutex mu;
        std::condition_variable cv;
        std::chrono::steady_clock::time_point t_begin, t_end;
        uint64_t issued = 0, landed = 0;   // render calls enqueued / whose last batch has finished
    } *clock = nullptr;                     // on the heap and never freed before the streams are idle (the callback holds it)
    hipStream_t own_stream = nullptr, stream = nullptr;   // lane 0's stream (stream may be caller-owned)
    hipEvent_t done_ev = nullptr;
    Lane lanes[PT_MAX_LANES];
    int n_lanes = 1;                 // lanes batches are scheduled on
    int n_lanes_alloc = 1;           // lanes that own streams / events / buffers (teardown, synchronisation)
    int next_lane = 0;
    int last_lane = -1;              // lane of the most recent batch (its acc_done orders the next accumulate)
    int last_batch_lane = 0;
    bool fb_external = false;
    float4 *fb_own = nullptr;
    DCounters *host_ctr = nullptr;   // pinned mirror of all banks, refreshed after every batch
    DCounters host_sum{};            // banks summed (by sum_counters)
    // progressive preview (pt_snapshot_framebuffer): one event per batch after its k_accumulate, in enqueue order
    struct Mark { hipEvent_t ev; uint64_t cum_samples; };
    std::deque<Mark> marks;
    std::vector<hipEvent_t> mark_free;
    uint64_t enqueued_samples = 0, accumulated_samples = 0;
    hipStream_t snap_stream = nullptr;
    float4 *snap_host = nullptr;     // pinned staging for the live-framebuffer copy
    DBatch last_batch{};
    bool have_last = false;
    // tile table of the current / last pt_render_tiles_async call (device copy + host mirror for reuse)
    DTile *d_tiles = nullptr;
    size_t d_tiles_cap = 0;
    std::vector<DTile> h_tiles;
    // tile-cost planner (pt_measure_tile_costs): one word per entry of the tile table while a planner call runs
    unsigned long long *tally = nullptr, *tally_buf = nullptr;
    size_t tally_cap = 0;
    std::vector<int> band_rect;      // rect index of every band of the current pt_render_tiles_async call
    std::vector<size_t> band_table;  // its index in the tile table
    // the scene's own build of the sweep (pt_spec.cpp): null = off / not applicable; launches use it once spec_poll says 1
    SpecJob *spec = nullptr;
    // profiling
    bool profiling = false;
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> ev_pool;
    pt_kernel_times ktimes{};
    DCounters ctr_at_profile_start{};
};

template <typename T>
static int dev_alloc(pt_ctx *c, T **p, size_t n)
{
    void *q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    c->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}
template <typename T>
static int dev_upload(pt_ctx *c, const T **dst, const std::vector<T> &v)
{
    T *p = nullptr;
    if (dev_alloc(c, &p, v.size())) return -1;
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = p;
    return 0;
}

static uint32_t mix_lowbias32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

static DRect make_rect(float x0, float z0, float x1, float z1, float y, int mat, int plane, int flipped)
{
    DRect r;
    r.x0 = x0; r.z0 = z0; r.x1 = x1; r.z1 = z1; r.y = y;
    r.plane = plane;
    r.ny = (float)(2 * (!flipped) - 1);   // primitive.h:212 with normal = !flipped (primitive.h:122)
    r.mat = mat;
    return r;
}

// Free-flight draws one constant_medium::hit can make (volume.h:70): its own and, when its boundary is a medium too, those of
// the two boundary->hit calls in front of it -- the dimension slots a volume instance's traversal owns in stream mode, in the
// reference's order: first boundary call, second boundary call, own draw (oracle/pt_oracle.c volume_draws, same rule).
static int volume_draws(const pt_scene_desc *sc, int pi)
{
    int n = 0;
    for (int depth = 0; pi >= 0 && pi < sc->n_primitives && sc->primitives[pi].type == PT_PRIM_VOLUME && depth < 8; depth++) {
        n = 2 * n + 1;
        pi = sc->primitives[pi].boundary;
    }
    return n;
}

// Does the plane "local coordinate `axis` = k" of instance ii WALL THE SCENE IN (DOp::slot bit 5 for a rect, bits 8-13 for the six sides
// of a box; pt_kernels.hip world_hit_fast_rb SHADOW)?  In the instance's own frame: every light's bounding box lies strictly on ONE
// side of the plane -- at least 2^-10 of the farthest any point of the scene gets from it, so that "beyond the sample by 2^-12 of the
// ray's length" holds for every origin -- and no corner of any other instance's bounding box, nor the primitive itself ([self_lo,
// self_hi] along the axis), lies on the far side by more than rounding (2^-13 of the lights' distance).  A shadow ray, which joins a
// point of some instance to a point of a light, then cannot cross the plane between t_min and the sample except by rounding at its
// own ends, and that case the kernel tests per ray.  A performance choice only: the kernel's proof is per ray.
static bool plane_walls_scene(const pt_scene_desc *sc, int ii, int axis, double k, double self_lo, double self_hi)
{
    const pt_instance &in = sc->instances[ii];
    if (sc->n_lights < 1 || sc->n_nodes < 1) return false;
    auto local = [&](const float *bbox, int c) {   // the corner's coordinate along the plane's axis, relative to the plane
        const double x = bbox[(c & 1) ? 3 : 0], y = bbox[(c & 2) ? 4 : 1], z = bbox[(c & 4) ? 5 : 2];
        return (double)in.inv[4 * axis] * x + (double)in.inv[4 * axis + 1] * y + (double)in.inv[4 * axis + 2] * z + (double)in.inv[4 * axis + 3] - k;
    };
    double side = 0.0, dmin = INFINITY, far = 0.0;
    for (int l = 0; l < sc->n_lights; l++) {
        const int li = sc->lights[l];
        if (li < 0 || li >= sc->n_instances || li == ii) return false;
        // the sample sits at t = 1 for a rect light only (rect::random returns point - origin, primitive.h:168-175); a sphere light
        // is sampled by direction (its hit lies at t = distance: every ray would go to the general sweep), other shapes not at all
        if (sc->primitives[sc->instances[li].primitive].type != PT_PRIM_RECT) return false;
        for (int c = 0; c < 8; c++) {
            const double v = local(sc->instances[li].bbox, c);
            if (!(std::fabs(v) > 0.0) || !std::isfinite(v)) return false;
            if (side == 0.0) side = v > 0 ? 1.0 : -1.0;
            if (v * side < 0.0) return false;
            dmin = std::min(dmin, std::fabs(v));
        }
    }
    for (int c = 0; c < 8; c++) far = std::max(far, std::fabs(local(sc->nodes[0].bbox, c)));
    if (!(dmin >= far * 0x1p-10)) return false;
    const double tol = dmin * 0x1p-13;
    if (!((self_lo - k) * side >= -tol) || !((self_hi - k) * side >= -tol)) return false;
    for (int j = 0; j < sc->n_instances; j++)
        for (int c = 0; c < 8 && j != ii; c++) {   // (the instance's own box is padded by 0.001; its own extent was given exactly)
            const double v = local(sc->instances[j].bbox, c);
            if (!(v * side >= -tol)) return false;
        }
    return true;
}
static bool emits(const pt_scene_desc *sc, int material)
{
    return material < 0 || material >= sc->n_materials || sc->materials[material].type == PT_MAT_DIFFUSE_LIGHT;
}

// bvh_node tree -> sweep program (pt_device.h).  `pending_push` is the slot the NEXT emitted op must push into.
static int emit_ops(const pt_scene_desc *sc, int child, int depth, std::vector<DOp> &ops, int &max_depth, int &pending_push)
{
    if (child < 0) {
        const int ii = ~child;
        const pt_instance &in = sc->instances[ii];
        const pt_primitive &p = sc->primitives[in.primitive];
        DOp op{};
        op.a = ii;
        op.push_slot = pending_push;
        pending_push = -1;
        memcpy(op.f, in.inv, 12 * sizeof(float));
        {   // shape of the inverse's linear part (pt_device.h DOp::slot): 1 = exactly the identity (pure translation: the
            // sweep adds the translation and skips 30 multiplies); 2 / 3 / 4 = the x / y / z axis is mapped to itself
            // (row and column of that axis zero off the diagonal: a rotation about one axis, with any scaling)
            const float *m = in.inv;
            const bool zx = m[1] == 0.0f && m[2] == 0.0f && m[4] == 0.0f && m[8] == 0.0f;
            const bool zy = m[4] == 0.0f && m[6] == 0.0f && m[1] == 0.0f && m[9] == 0.0f;
            const bool zz = m[8] == 0.0f && m[9] == 0.0f && m[2] == 0.0f && m[6] == 0.0f;
            if (zx && zy && zz && m[0] == 1.0f && m[5] == 1.0f && m[10] == 1.0f) op.slot = 1;
            else if (zx) op.slot = 2;
            else if (zy) op.slot = 3;
            else if (zz) op.slot = 4;
            else op.slot = 0;
        }
        switch (p.type) {
        case PT_PRIM_RECT:
            op.kind = p.plane == PT_PLANE_XY ? OP_LEAF_RECT_XY : (p.plane == PT_PLANE_YZ ? OP_LEAF_RECT_YZ : OP_LEAF_RECT_XZ);
            memcpy(op.g, p.rect, 5 * sizeof(float));
            if (!emits(sc, p.material) && plane_walls_scene(sc, ii, p.plane == PT_PLANE_XY ? 2 : (p.plane == PT_PLANE_YZ ? 0 : 1), p.rect[4], p.rect[4], p.rect[4]))
                op.slot |= 32;
            break;
        case PT_PRIM_BOX:
            op.kind = OP_LEAF_BOX;
            memcpy(op.g, p.p0, 12); memcpy(op.g + 3, p.p1, 12);
            // box::hit's sides in its order XY(p0.z) XY(p1.z) YZ(p0.x) YZ(p1.x) XZ(p0.y) XZ(p1.y) (primitive.h:229-242): bits 8-13
            for (int f = 0; f < 6 && !emits(sc, p.material); f++) {
                const int axis = f < 2 ? 2 : (f < 4 ? 0 : 1);
                if (plane_walls_scene(sc, ii, axis, (f & 1) ? p.p1[axis] : p.p0[axis], p.p0[axis], p.p1[axis])) op.slot |= 256 << f;
            }
            break;
        case PT_PRIM_SPHERE:
            op.kind = OP_LEAF_SPHERE;
            memcpy(op.g, p.center, 12); op.g[3] = p.radius;
            break;
        case PT_PRIM_VOLUME: {
            if (p.boundary < 0 || p.boundary >= sc->n_primitives) return -1;
            const pt_primitive &bd = sc->primitives[p.boundary];
            if (bd.type == PT_PRIM_VOLUME && (bd.boundary < 0 || bd.boundary >= sc->n_primitives)) return -1;
            if (bd.type == PT_PRIM_BOX) {
                op.kind = OP_LEAF_VOLBOX;
                memcpy(op.g, bd.p0, 12); memcpy(op.g + 3, bd.p1, 12);
                op.g[6] = p.density;
            } else if (bd.type == PT_PRIM_SPHERE) {
                op.kind = OP_LEAF_VOLSPHERE;
                memcpy(op.g, bd.center, 12); op.g[3] = bd.radius;
                op.g[6] = p.density;
            } else if (bd.type == PT_PRIM_RECT) {
                op.kind = OP_LEAF_NONE;
            } else if (bd.type == PT_PRIM_VOLUME) {
                // a medium whose boundary is a medium: the outer's two boundary queries are two scattering events of the inner
                // one (volume.h:29-93 twice); one level of nesting is carried, the inner's own boundary being a box or a sphere
                const pt_primitive &ib = sc->primitives[bd.boundary];
                int32_t shape = 0;
                if (ib.type == PT_PRIM_BOX) { shape = 1; memcpy(op.g, ib.p0, 12); memcpy(op.g + 3, ib.p1, 12); }
                else if (ib.type == PT_PRIM_SPHERE) { shape = 2; memcpy(op.g, ib.center, 12); op.g[3] = ib.radius; }
                else if (ib.type == PT_PRIM_RECT) shape = 0;   // the inner medium is never hit, so neither is the outer one
                else return -2;                                // three media deep: refused loudly by the caller
                op.kind = shape ? OP_LEAF_VOLVOL : OP_LEAF_NONE;
                op.g[6] = p.density; op.g[8] = bd.density;
                memcpy(&op.g[9], &shape, 4);
            } else return -2;
            break;
        }
        default: return -1;
        }
        // Wave-level leaf cull (pt_kernels.hip wave_skips_cullable): a box, or a medium on a box, carries the instance's world
        // bounding box (instance::bbox, primitive.h:272-296) ENLARGED by a margin that dominates every rounding between the exact
        // line and the hit points rect::hit computes (g[10..15]; bit 6 of slot): a wave none of whose rays touches it skips the
        // leaf.  Only for coordinates within 2^12 and scales within 2^+-4, where the margin's argument holds.
        if (op.kind == OP_LEAF_BOX || op.kind == OP_LEAF_VOLBOX) {
            bool ok = true;
            float ext = 0.0f;
            for (int k = 0; k < 3; k++) {
                ok = ok && std::isfinite(in.bbox[k]) && std::isfinite(in.bbox[k + 3]) && std::fabs(in.bbox[k]) <= 4096.0f && std::fabs(in.bbox[k + 3]) <= 4096.0f;
                ext = std::max(ext, in.bbox[k + 3] - in.bbox[k]);
            }
            for (int r = 0; r < 3; r++) {   // column norms of the forward linear part: how far a local rounding error can travel in world space
                const float c = std::fabs(in.fwd[r]) + std::fabs(in.fwd[4 + r]) + std::fabs(in.fwd[8 + r]);
                ok = ok && c >= 0.0625f && c <= 16.0f;
            }
            if (ok) {
                const float m = 0.25f + ext * 0.015625f;
                for (int k = 0; k < 3; k++) { op.g[10 + k] = in.bbox[k] - m; op.g[13 + k] = in.bbox[k + 3] + m; }
                op.slot |= 64;
            }
        }
        ops.push_back(op);
        return 0;
    }
    if (child >= sc->n_nodes) return -1;
    const pt_bvh_node &n = sc->nodes[child];
    size_t me = ops.size();
    DOp en{};
    en.kind = OP_ENTER;
    en.push_slot = pending_push;
    pending_push = -1;
    memcpy(en.f, n.bbox, 6 * sizeof(float));
    ops.push_back(en);
    if (emit_ops(sc, n.left, depth, ops, max_depth, pending_push)) return -1;
    pending_push = depth;   // the left result is parked in slot `depth` by whatever op starts eval(right)
    max_depth = std::max(max_depth, depth + 1);
    if (emit_ops(sc, n.right, depth + 1, ops, max_depth, pending_push)) return -1;
    DOp co{};
    co.kind = OP_COMBINE;
    co.slot = depth;
    co.push_slot = -1;
    ops.push_back(co);
    ops[me].a = (int)ops.size();
    return 0;
}

// ---- per-face hit record table (pt_device.h PT_FACE_F4).  Plain float arithmetic in the association of the device code
// (pt_kernels.hip xf_normal / vunit / onb_from_w); this file is compiled with -ffp-contract=off like the kernels.
namespace {
struct hv3 { float x, y, z; };
inline hv3 h_xf_normal(const float *inv, hv3 n)
{   // normalize((L^-1)^T n), Eigen's norm association x2 + (y2 + z2)   (transform3.h:60-63)
    const float x = (inv[0] * n.x + inv[4] * n.y) + inv[8] * n.z;
    const float y = (inv[1] * n.x + inv[5] * n.y) + inv[9] * n.z;
    const float z = (inv[2] * n.x + inv[6] * n.y) + inv[10] * n.z;
    const float nrm = sqrtf(x * x + (y * y + z * z));
    return hv3{x / nrm, y / nrm, z / nrm};
}
inline hv3 h_vunit(hv3 v)
{   // vec3.h unit_vector: v / sqrt(x*x + y*y + z*z)
    const float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return hv3{v.x / l, v.y / l, v.z / l};
}
inline hv3 h_cross(hv3 a, hv3 b) { return hv3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline hv3 h_shuffle(hv3 v, int plane)
{   // primitive.h:104-121
    if (plane == 0) return hv3{v.x, v.z, v.y};
    if (plane == 2) return hv3{v.y, v.x, v.z};
    return v;
}
void face_side(const float *inv, hv3 nl, float4 *out3)
{
    const hv3 n = h_xf_normal(inv, nl);
    const hv3 w = h_vunit(n);                                             // helpers.h:127-136
    const hv3 a = (fabsf(w.x) > 0.9) ? hv3{0.0f, 1.0f, 0.0f} : hv3{1.0f, 0.0f, 0.0f};
    const hv3 v = h_vunit(h_cross(w, a));
    const hv3 u = h_cross(w, v);
    out3[0] = make_float4(n.x, n.y, n.z, w.x);
    out3[1] = make_float4(w.y, w.z, u.x, u.y);
    out3[2] = make_float4(u.z, v.x, v.y, v.z);
}
}  // namespace
static void build_faces(const std::vector<DInst> &insts, const std::vector<DPrim> &prims, const std::vector<DMat> &mats, std::vector<float4> &faces)
{
    faces.assign(insts.size() * 8 * PT_FACE_F4, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t i = 0; i < insts.size(); i++) {
        const DPrim &pr = prims[insts[i].prim];
        for (int f = 0; f < 8; f++) {
            float4 *out = &faces[(i * 8 + f) * PT_FACE_F4];
            hv3 nl{1.0f, 0.0f, 0.0f};                                      // constant_medium: volume.h:85
            if (pr.type <= 1) {                                             // rect (face 0) / box side: primitive.h:212
                const DRect &q = pr.r[(pr.type == 0) ? 0 : std::min(f, 5)];
                nl = h_shuffle(hv3{0.0f, q.ny, 0.0f}, q.plane);
            }
            // shading class of a hit on this face (k_shade's sort key): 0 untextured lambertian / metal on a rect or box,
            // 2 emitter, 1 everything else
            const DMat &fm = mats[pr.hit_mat[f]];
            const int cls = (fm.type == PT_MAT_DIFFUSE_LIGHT) ? 2 : (((fm.type == PT_MAT_LAMBERTIAN || fm.type == PT_MAT_METAL) && fm.tex < 0 && pr.type <= 1) ? 0 : 1);
            const int32_t head = pr.hit_mat[f] | (pr.type << 24) | (cls << 28);
            float hb;
            memcpy(&hb, &head, 4);
            out[0] = make_float4(hb, nl.x, nl.y, nl.z);
            static_assert(sizeof(DMat) == 48 && PT_FACE_F4 >= 11, "the face table embeds the material record");
            memcpy(out + 8, &fm, sizeof(DMat));
            if (pr.type == 2) continue;                                     // sphere: the normal depends on the hit point
            face_side(insts[i].inv, nl, out + 1);
            face_side(insts[i].inv, hv3{-nl.x, -nl.y, -nl.z}, out + 4);     // vec3 operator-: (-x, -y, -z)
        }
    }
}

// The traversal programs of a scene (pt_device.h DOp), built on the host alone: the general program (ENTER / LEAF /
// COMBINE), behind it the fast program (pt_kernels.hip world_hit_fast), the parent-box table of the flat program and the
// flags the kernels are picked by.  No device call: pt_spec_source (the per-scene build of the sweep) uses it without a GPU.
struct HostProgram {
    std::vector<DOp> ops;          // general program, one padding op, fast program, one padding op
    std::vector<float4> chains;
    int n_general = 0, n_fast = 0, n_chain = 0, max_depth = 0, tame = 0, geom_all = 0;
};
static int build_program(const pt_scene_desc *sc, HostProgram &hp)
{
    if (!sc || sc->n_instances < 1 || sc->n_primitives < 1 || sc->n_nodes < 1) { set_err("pt_create: empty scene"); return -1; }
    std::vector<int32_t> vol_ordinal(sc->n_instances, -1);   // first stream RNG dimension slot of a volume instance's traversal
    for (int i = 0, nvol = 0; i < sc->n_instances; i++) {
        const int pi = sc->instances[i].primitive;
        if (pi < 0 || pi >= sc->n_primitives) { set_err("pt_create: instance %d: bad primitive", i); return -1; }
        if (sc->primitives[pi].type == PT_PRIM_VOLUME) { vol_ordinal[i] = nvol; nvol += volume_draws(sc, pi); }
    }
    std::vector<DOp> &ops = hp.ops;
    int max_depth = 0;
    // validate child indices first
    for (int i = 0; i < sc->n_nodes; i++)
        for (int ch : {sc->nodes[i].left, sc->nodes[i].right})
            if ((ch >= 0 && (ch >= sc->n_nodes || ch <= i)) || (ch < 0 && ~ch >= sc->n_instances)) {
                set_err("pt_create: bvh node %d: bad child %d (nodes must be in preorder)", i, ch);
                return -1;
            }
    int pending_push = -1;
    if (emit_ops(sc, 0, 0, ops, max_depth, pending_push)) {
        set_err("pt_create: malformed BVH, or constant_mediums nested three deep (a medium whose boundary is a medium whose boundary is a medium)");
        return -1;
    }
    if (max_depth > 64) {
        set_err("pt_create: BVH needs %d short-stack slots, the sweep traversal holds 64", max_depth);
        return -1;
    }
    for (DOp &op : ops)
        if (op.kind == OP_LEAF_VOLBOX || op.kind == OP_LEAF_VOLSPHERE || op.kind == OP_LEAF_VOLVOL) { int32_t vo = vol_ordinal[op.a]; memcpy(&op.g[7], &vo, 4); }
    if (ops.size() >= (1u << 22)) { set_err("pt_create: traversal program too long (%zu ops)", ops.size()); return -1; }
    ops.push_back(DOp{});   // padding op (never executed)
    // the fast program (pt_kernels.hip world_hit_fast): the same list without the COMBINE ops (the fold over the leaves
    // needs neither them nor the pushes), ENTER's skip target re-indexed; stored behind the general program
    const int n_general = (int)ops.size() - 1;
    // flat mode (pt_device.h DScene::chains): few instances, shallow tree -> leaves only, ancestors checked afterwards
    const bool flat = sc->n_instances <= PT_FLAT_MAX_INSTANCES && max_depth <= PT_MAX_STACK && !env_token("PATHTRACE_HIP_TRAVERSAL", "tree");
    // (the root's ENTER stays: a wave of camera rays that miss the scene's box leaves the program after one op)
    auto in_fast = [&](const DOp &o) { return o.kind != OP_COMBINE && !(flat && o.kind == OP_ENTER && &o != &ops[0]); };
    std::vector<int> fast_index(n_general + 1, 0);
    for (int i = 0, k = 0; i <= n_general; i++) { fast_index[i] = k; if (i < n_general && in_fast(ops[i])) k++; }
    const int n_fast = fast_index[n_general];
    std::vector<float4> &chains = hp.chains;
    int n_chain = 0;
    if (flat) {
        // One box per leaf suffices: its PARENT's.  A node's box is surrounding_box of its children's (aabb.h:55-64:
        // componentwise fmin / fmax), so every ancestor's box contains the parent's bound for bound, and the slab test is
        // monotone in the bounds -- (b - o) * invD and the min / max over them only move outwards when a bound does,
        // rounding included -- hence a ray that hits the parent's box hits every ancestor's.  Leaves hanging off the root
        // get an all-space box (the root's ENTER op is in the program).
        n_chain = 1;
        chains.assign((size_t)sc->n_instances * 2, make_float4(0.f, 0.f, 0.f, 0.f));
        for (size_t k = 0; k < chains.size(); k += 2) {
            chains[k] = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, 0.f);
            chains[k + 1] = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
        }
        std::vector<int> open_nodes;   // ENTER ops whose subtree holds the current op
        for (int i = 0; i < n_general; i++) {
            while (!open_nodes.empty() && ops[open_nodes.back()].a <= i) open_nodes.pop_back();
            if (ops[i].kind == OP_ENTER) open_nodes.push_back(i);
            else if (ops[i].kind >= OP_LEAF_RECT_XY && open_nodes.size() > 1) {
                const float *bb = ops[open_nodes.back()].f;
                chains[(size_t)ops[i].a * 2] = make_float4(bb[0], bb[1], bb[2], 0.f);
                chains[(size_t)ops[i].a * 2 + 1] = make_float4(bb[3], bb[4], bb[5], 0.f);
            }
        }
    } else chains.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
    for (int i = 0; i < n_general; i++) {
        if (!in_fast(ops[i])) continue;
        DOp f = ops[i];
        {   // the fast sweep compares positions as floats: its own index rides in push_slot (unused there), ENTER's target in f[6]
            const float self = (float)fast_index[i];
            memcpy(&f.push_slot, &self, 4);
        }
        if (f.kind == OP_ENTER) { f.a = fast_index[ops[i].a]; f.f[6] = (float)f.a; }
        ops.push_back(f);
    }
    ops.push_back(DOp{});   // padding
    hp.geom_all = 0;
    for (const DOp &op : ops) hp.geom_all |= (op.kind >= OP_LEAF_SPHERE);
    {   // precondition of the sweep's unscaled division (pt_fdiv.h, pt_kernels.hip world_hit_fast): a leaf's linear part is
        // zero or within [2^-44, 2^13] per entry (float residues of a rotation by a multiple of pi/2 are ~1e-8), its
        // translation zero or within [2^-44, 2^20], its bounds zero or within [2^-20, 2^20].  PATHTRACE_HIP_TRAVERSAL=general (the
        // tests hold the sweeps to each other with it) or an unordered / non-finite node box send the whole scene to the general sweep.
        auto in_range = [](float x, int lo, int hi) {
            uint32_t u;
            memcpy(&u, &x, 4);
            u &= 0x7fffffffu;
            return u == 0u || (u - ((uint32_t)(127 + lo) << 23)) <= ((uint32_t)(hi - lo) << 23);
        };
        hp.tame = env_token("PATHTRACE_HIP_TRAVERSAL", "general") ? 0 : 1;
        for (const DOp &op : ops) if (op.kind == OP_LEAF_VOLVOL) hp.tame = 0;   // a medium in a medium: the general sweep carries it
        for (int oi = 0; oi < (int)ops.size(); oi++) {
            DOp &op = ops[oi];
            if (op.kind == OP_ENTER) {   // world_hit_fast takes min / max of the slab products: the box must be ordered and finite
                for (int i = 0; i < 3; i++) hp.tame &= (op.f[i] <= op.f[i + 3] && std::isfinite(op.f[i]) && std::isfinite(op.f[i + 3])) ? 1 : 0;
                continue;
            }
            if (op.kind < OP_LEAF_RECT_XY) continue;
            // per leaf: outside the ranges (e.g. the 1e-15 .. 1e-22 residues of rotations composed about several axes) the leaf
            // keeps its IEEE divisions inside the fast sweep (bit 4 of DOp::slot), everything else of the sweep stays
            bool leaf_tame = true;
            for (int i = 0; i < 12; i++) leaf_tame = leaf_tame && in_range(op.f[i], -44, (i & 3) == 3 ? 20 : 13);
            const int np = (op.kind <= OP_LEAF_RECT_YZ) ? 5 : ((op.kind == OP_LEAF_BOX || op.kind == OP_LEAF_VOLBOX) ? 6 : 0);   // rect x0 z0 x1 z1 y; box p0 p1; spheres divide the IEEE way
            for (int i = 0; i < np; i++) leaf_tame = leaf_tame && in_range(op.g[i], -20, 20);
            if (!leaf_tame) op.slot |= 16;
        }
    }
    hp.n_general = n_general; hp.n_fast = n_fast; hp.n_chain = n_chain; hp.max_depth = max_depth;
    return 0;
}

// The scene's fast program as the compile-time table of the per-scene sweep build (pt_kernels.hip PT_SPEC_HEADER): a
// header text.  Host only.  Returns the text length (without the terminator), or < 0 when the scene has no fast program
// to specialise (not tame, walked, or too long to unroll).
#define PT_SPEC_MAX_OPS 96
static int spec_header_text(const pt_scene_desc *sc, std::string &out)
{
    HostProgram hp;
    if (build_program(sc, hp)) return -1;
    if (!hp.tame) { set_err("pt_spec_header: the scene does not take the fast sweep"); return -2; }
    if (hp.n_fast > PT_SPEC_MAX_OPS || sc->n_instances > PT_WALK_MIN_INSTANCES) { set_err("pt_spec_header: %d ops are too many to unroll", hp.n_fast); return -2; }
    char line[64];
    out.clear();
    snprintf(line, sizeof line, "#define PT_SPEC_N %d\n", hp.n_fast);
    out += line;
    snprintf(line, sizeof line, "#define PT_SPEC_GA %d\n", hp.geom_all ? 1 : 0);
    out += line;
    {   // flat program: the root's ENTER is the only one (pt_kernels.hip world_hit_fast_rb drops the skip positions)
        int enters = 0;
        for (int i = hp.n_general + 1; i < hp.n_general + 1 + hp.n_fast; i++) enters += hp.ops[i].kind == OP_ENTER;
        snprintf(line, sizeof line, "#define PT_SPEC_FLAT %d\n", (enters == 1 && hp.ops[hp.n_general + 1].kind == OP_ENTER) ? 1 : 0);
        out += line;
    }
    {   // leaves a whole wave may skip (bit 6 of DOp::slot): only worth a test when some leaf stays and some can go
        int ncull = 0, nleaf = 0;
        for (int i = hp.n_general + 1; i < hp.n_general + 1 + hp.n_fast; i++) {
            const DOp &o = hp.ops[i];
            if (o.kind < OP_LEAF_RECT_XY) continue;
            nleaf++;
            if ((o.slot & 64) && !(o.slot & 16)) ncull++;
        }
        snprintf(line, sizeof line, "#define PT_SPEC_NCULL %d\n", (ncull < nleaf) ? ncull : 0);
        out += line;
        int nwall = 0;   // rects and box sides that wall the scene in (bits 5 / 8-13 of DOp::slot): k_connect proves them unreachable instead of testing them
        for (int i = hp.n_general + 1; i < hp.n_general + 1 + hp.n_fast; i++) {
            const DOp &o = hp.ops[i];
            nwall += o.kind >= OP_LEAF_RECT_XY && o.kind <= OP_LEAF_RECT_YZ && (o.slot & 48) == 32;
            if (o.kind == OP_LEAF_BOX && !(o.slot & 16)) nwall += __builtin_popcount((o.slot >> 8) & 63);
        }
        snprintf(line, sizeof line, "#define PT_SPEC_NWALL %d\n", nwall);
        out += line;
    }
    out += "static __device__ constexpr int kSpecW[PT_SPEC_N][32] = {\n";
    for (int i = hp.n_general + 1; i < hp.n_general + 1 + hp.n_fast; i++) {
        int32_t w[32];
        memcpy(w, &hp.ops[i], 128);
        out += "  {";
        for (int k = 0; k < 32; k++) { snprintf(line, sizeof line, "%d%s", w[k], k < 31 ? ", " : ""); out += line; }
        out += "},\n";
    }
    out += "};\n";
    return (int)out.size();
}
extern "C" int pt_spec_header(const pt_scene_desc *scene, char *buf, size_t cap)
{
    std::string text;
    const int n = spec_header_text(scene, text);
    if (n < 0) return n;
    if (buf && cap > (size_t)n) memcpy(buf, text.c_str(), (size_t)n + 1);
    return n;
}

static int build_scene(pt_ctx *c, const pt_scene_desc *sc)
{
    if (!sc || sc->n_instances < 1 || sc->n_primitives < 1 || sc->n_materials < 1 || sc->n_nodes < 1) {
        set_err("pt_create: empty scene");
        return -1;
    }
    if (sc->n_instances >= (1 << 25)) {   // hit id = instance * 8 + face below bit 28 (bits 28-29 carry the shading class)
        set_err("pt_create: %d instances exceed 2^25", sc->n_instances);
        return -1;
    }
    std::vector<DMat> mats(sc->n_materials);
    for (int i = 0; i < sc->n_materials; i++) {
        const pt_material &m = sc->materials[i];
        if (m.type < 0 || m.type > PT_MAT_ISOTROPIC) { set_err("pt_create: bad material type %d", m.type); return -1; }
        DMat d{};
        d.type = m.type; d.r = m.color[0]; d.g = m.color[1]; d.b = m.color[2];
        d.alpha = m.alpha; d.power = m.power; d.two_sided = m.two_sided;
        d.tex = -1;
        // lambertian::scatter / metal::scatter: attenuation = albedo->value(..) / M_PI with M_PI narrowed to float by
        // vec3::operator/(float) (material.h:46, 94; vec3.h): the same three float divisions the kernel performed per hit
        d.att[0] = d.r / 3.14159274f; d.att[1] = d.g / 3.14159274f; d.att[2] = d.b / 3.14159274f;
        if (m.texture >= 0) {
            if (m.texture >= sc->n_textures) { set_err("pt_create: material %d: bad texture index %d", i, m.texture); return -1; }
            // metal's albedo is a plain colour and a dielectric has none (material.h:79, 113-117)
            if (m.type == PT_MAT_LAMBERTIAN || m.type == PT_MAT_DIFFUSE_LIGHT || m.type == PT_MAT_ISOTROPIC) d.tex = m.texture;
        }
        mats[i] = d;
    }
    // ---- textures (texture.h, image.h) ----
    std::vector<DTex> texs((size_t)std::max(sc->n_textures, 0));
    std::vector<float4> texels;
    bool any_perlin = false;
    for (int i = 0; i < sc->n_textures; i++) {
        const pt_texture &t = sc->textures[i];
        DTex d{};
        d.type = t.type; d.even = t.even; d.odd = t.odd; d.scale = t.scale;
        d.r = t.color[0]; d.g = t.color[1]; d.b = t.color[2]; d.a = t.alpha;
        switch (t.type) {
        case PT_TEX_CONSTANT: break;
        case PT_TEX_CHECKER:
            if (t.even < 0 || t.even >= i || t.odd < 0 || t.odd >= i) {
                set_err("pt_create: texture %d: checker children must be earlier entries of the table", i);
                return -1;
            }
            d.uses_uv = texs[t.even].uses_uv | texs[t.odd].uses_uv;
            break;
        case PT_TEX_PERLIN: any_perlin = true; break;
        case PT_TEX_IMAGE: {
            const int64_t n = (int64_t)t.width * t.height;
            if (t.width < 1 || t.height < 1 || t.texel_offset < 0 || !sc->texels || t.texel_offset + 4 * n > sc->texel_bytes ||
                (int64_t)texels.size() + n > INT32_MAX) {
                set_err("pt_create: texture %d: bad image extent", i);
                return -1;
            }
            d.width = t.width; d.height = t.height; d.texel0 = (int32_t)texels.size(); d.uses_uv = 1;
            const uint8_t *px = sc->texels + t.texel_offset;
            for (int64_t k = 0; k < n; k++)   // from_4byte_vector image.h:52-69: byte / 255.0 in double, stored as float
                texels.push_back(make_float4((float)(px[4 * k] / 255.0), (float)(px[4 * k + 1] / 255.0),
                                             (float)(px[4 * k + 2] / 255.0), (float)(px[4 * k + 3] / 255.0)));
            break;
        }
        default: set_err("pt_create: texture %d: bad type %d", i, t.type); return -1;
        }
        texs[i] = d;
    }
    for (int i = 0; i < sc->n_materials; i++)
        if (mats[i].type == PT_MAT_DIFFUSE_LIGHT && mats[i].tex >= 0 && texs[mats[i].tex].uses_uv) {
            // the NEE ray of a path that lands on the light lies in the light's plane, rect::hit accepts its NaN t and
            // image_texture::alpha indexes with NaN u, v (image.h:46): the reference crashes on such a scene
            set_err("pt_create: material %d: an image-textured diffuse_light is not renderable by the reference (NaN u, v "
                    "index the image out of bounds)", i);
            return -1;
        }
    if (any_perlin && (!sc->perlin_ranvec || !sc->perlin_perm)) {
        set_err("pt_create: a perlin texture needs pt_scene_desc::perlin_ranvec / perlin_perm");
        return -1;
    }
    if (sc->background_texture >= sc->n_textures) { set_err("pt_create: bad background_texture %d", sc->background_texture); return -1; }
    std::vector<DPrim> prims(sc->n_primitives);
    for (int i = 0; i < sc->n_primitives; i++) {
        const pt_primitive &p = sc->primitives[i];
        DPrim d{};
        d.type = p.type; d.mat = p.material;
        if (p.material < 0 || p.material >= sc->n_materials) { set_err("pt_create: primitive %d: bad material", i); return -1; }
        switch (p.type) {
        case PT_PRIM_RECT:
            d.r[0] = make_rect(p.rect[0], p.rect[1], p.rect[2], p.rect[3], p.rect[4], p.material, p.plane, p.flipped);
            break;
        case PT_PRIM_BOX: {   // primitive.h:232-240
            const float *a = p.p0, *b = p.p1;
            d.r[0] = make_rect(a[0], a[1], b[0], b[1], a[2], p.material, PT_PLANE_XY, 1);
            d.r[1] = make_rect(a[0], a[1], b[0], b[1], b[2], p.material, PT_PLANE_XY, 0);
            d.r[2] = make_rect(a[1], a[2], b[1], b[2], a[0], p.material, PT_PLANE_YZ, 1);
            d.r[3] = make_rect(a[1], a[2], b[1], b[2], b[0], p.material, PT_PLANE_YZ, 0);
            d.r[4] = make_rect(a[0], a[2], b[0], b[2], a[1], p.material, PT_PLANE_XZ, 1);
            d.r[5] = make_rect(a[0], a[2], b[0], b[2], b[1], p.material, PT_PLANE_XZ, 0);
            break;
        }
        case PT_PRIM_SPHERE:
            d.cx = p.center[0]; d.cy = p.center[1]; d.cz = p.center[2]; d.radius = p.radius;
            break;
        case PT_PRIM_VOLUME:
            if (p.boundary < 0 || p.boundary >= i || p.phase_material < 0 || p.phase_material >= sc->n_materials) {
                set_err("pt_create: volume primitive %d: bad boundary / phase material", i);
                return -1;
            }
            d.boundary = p.boundary; d.density = p.density; d.phase_mat = p.phase_material;
            break;
        default: set_err("pt_create: primitive %d: unsupported type %d", i, p.type); return -1;
        }
        for (int f = 0; f < 8; f++) d.hit_mat[f] = (p.type == PT_PRIM_VOLUME) ? p.phase_material : p.material;
        {   // sphere::hit and constant_medium::hit leave rec.u / rec.v unset (primitive.h:63-102, volume.h:29-93): an
            // image texture there reads indeterminate values in the reference
            const DMat &hm = mats[d.hit_mat[0]];
            if (hm.tex >= 0 && texs[hm.tex].uses_uv && p.type != PT_PRIM_RECT && p.type != PT_PRIM_BOX) {
                set_err("pt_create: primitive %d: an image texture needs the u, v of a rect / box hit", i);
                return -1;
            }
        }
        prims[i] = d;
    }
    std::vector<DInst> insts(sc->n_instances);
    int nvol = 0;
    for (int i = 0; i < sc->n_instances; i++) {
        const pt_instance &p = sc->instances[i];
        if (p.primitive < 0 || p.primitive >= sc->n_primitives) { set_err("pt_create: instance %d: bad primitive", i); return -1; }
        DInst d{};
        memcpy(d.inv, p.inv, sizeof d.inv);
        memcpy(d.fwd, p.fwd, sizeof d.fwd);
        d.prim = p.primitive;
        d.vol_ordinal = -1;
        if (prims[p.primitive].type == PT_PRIM_VOLUME) { d.vol_ordinal = nvol; nvol += volume_draws(sc, p.primitive); }
        {
            auto is_ident = [](const float *m) {
                return m[0] == 1.0f && m[5] == 1.0f && m[10] == 1.0f && m[1] == 0.0f && m[2] == 0.0f && m[4] == 0.0f &&
                       m[6] == 0.0f && m[8] == 0.0f && m[9] == 0.0f;
            };
            d.ident = is_ident(p.inv) && is_ident(p.fwd);
        }
        insts[i] = d;
    }
    std::vector<int32_t> lights(sc->lights, sc->lights + sc->n_lights);
    for (int l : lights)
        if (l < 0 || l >= sc->n_instances) { set_err("pt_create: bad light index %d", l); return -1; }
    if (c->cfg.light_samples > 0 && lights.empty()) {
        set_err("pt_create: light_samples > 0 but the scene has no lights (the reference indexes lights[0] "
                "unconditionally, world.h:31-35)");
        return -1;
    }
    HostProgram hp;
    if (build_program(sc, hp)) return -1;
    std::vector<DOp> &ops = hp.ops;
    const std::vector<float4> &chains = hp.chains;
    const int n_general = hp.n_general, n_fast = hp.n_fast, n_chain = hp.n_chain, max_depth = hp.max_depth;
    // emitted radiance by hit id (instance*8 + face): power * emit->value * emit->alpha (material.h:219), the same two
    // float multiplications the kernels would do (this file is compiled with -ffp-contract=off)
    std::vector<float4> emit((size_t)sc->n_instances * 8, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int i = 0; i < sc->n_instances; i++)
        for (int f = 0; f < 8; f++) {
            const DMat &m = mats[prims[insts[i].prim].hit_mat[f]];
            if (m.type != PT_MAT_DIFFUSE_LIGHT) continue;
            const float pr = m.power * m.r, pg = m.power * m.g, pb = m.power * m.b;
            emit[(size_t)i * 8 + f] = make_float4(m.alpha * pr, m.alpha * pg, m.alpha * pb,
                                                  (m.two_sided ? 0.f : 1.f) + (m.tex >= 0 ? 2.f : 0.f));
        }
    std::vector<float4> ranvec(256, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<int32_t> perm(768, 0);
    if (sc->perlin_ranvec && sc->perlin_perm) {
        for (int i = 0; i < 256; i++) ranvec[i] = make_float4(sc->perlin_ranvec[3 * i], sc->perlin_ranvec[3 * i + 1], sc->perlin_ranvec[3 * i + 2], 0.f);
        for (int i = 0; i < 768; i++) {
            if (sc->perlin_perm[i] < 0 || sc->perlin_perm[i] > 255) { set_err("pt_create: perlin_perm[%d] out of range", i); return -1; }
            perm[i] = sc->perlin_perm[i];
        }
    }
    if (texs.empty()) texs.push_back(DTex{});
    if (texels.empty()) texels.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
    DScene &S = c->S;
    std::vector<float4> faces;
    if (mats.size() >= (1u << 20)) { set_err("pt_create: too many materials"); return -1; }
    build_faces(insts, prims, mats, faces);
    if (dev_upload(c, &S.faces, faces)) return -1;
    if (dev_upload(c, &S.emit, emit)) return -1;
    if (dev_upload(c, &S.tex, texs) || dev_upload(c, &S.texels, texels) || dev_upload(c, &S.ranvec, ranvec) || dev_upload(c, &S.perm, perm))
        return -1;
    S.bg_tex = sc->background_texture < 0 ? -1 : sc->background_texture;
    S.geom_all = hp.geom_all;
    S.tame = hp.tame;
    S.textured = S.bg_tex >= 0;
    for (const DMat &m : mats) S.textured |= m.tex >= 0;
    if (dev_upload(c, &S.insts, insts) || dev_upload(c, &S.prims, prims) || dev_upload(c, &S.mats, mats) ||
        dev_upload(c, &S.ops, ops) || dev_upload(c, &S.lights, lights))
        return -1;
    S.n_insts = (int)insts.size(); S.n_prims = (int)prims.size(); S.n_mats = (int)mats.size();
    S.n_ops = n_general; S.ops_fast_off = n_general + 1; S.n_ops_fast = n_fast;
    S.n_chain = n_chain;
    if (dev_upload(c, &S.chains, chains)) return -1;
    {   // the walk's node table (pt_device.h DScene::wnodes): the caller's preorder nodes, a leaf child pointing at the leaf's
        // DOp of the fast program
        std::vector<int> leaf_op(sc->n_instances, -1);
        for (int i = S.ops_fast_off; i < S.ops_fast_off + n_fast; i++)
            if (ops[i].kind >= OP_LEAF_RECT_XY) leaf_op[ops[i].a] = i;
        std::vector<float4> wn((size_t)sc->n_nodes * 2);
        bool ok = true;
        for (int i = 0; i < sc->n_nodes; i++) {
            const pt_bvh_node &n = sc->nodes[i];
            auto child = [&](int ch) { if (ch >= 0) return ch; const int li = leaf_op[~ch]; ok = ok && li >= 0; return ~li; };
            const int32_t l = child(n.left), r = child(n.right);
            float lf, rf;
            memcpy(&lf, &l, 4); memcpy(&rf, &r, 4);
            wn[2 * i] = make_float4(n.bbox[0], n.bbox[1], n.bbox[2], lf);
            wn[2 * i + 1] = make_float4(n.bbox[3], n.bbox[4], n.bbox[5], rf);
        }
        if (!ok) { set_err("pt_create: a bvh leaf without an op"); return -1; }
        S.walk = (sc->n_instances > PT_WALK_MIN_INSTANCES && S.tame && !env_token("PATHTRACE_HIP_TRAVERSAL", "sweep")) ? 1 : 0;
        if (env_token("PATHTRACE_HIP_TRAVERSAL", "walk") && S.tame) S.walk = 1;   // measurement / test knob: walk small scenes too
        if (dev_upload(c, &S.wnodes, wn)) return -1;
    }
    S.n_lights = (int)lights.size(); S.n_vol = nvol;
    S.stack_depth = std::max(max_depth, 1);
    const pt_camera &cm = sc->camera;
    memcpy(S.cam.origin, cm.origin, 12); memcpy(S.cam.llc, cm.lower_left_corner, 12);
    memcpy(S.cam.horizontal, cm.horizontal, 12); memcpy(S.cam.vertical, cm.vertical, 12);
    memcpy(S.cam.u, cm.u, 12); memcpy(S.cam.v, cm.v, 12);
    S.cam.lens_radius = cm.lens_radius;
    memcpy(S.bg, sc->background, 12);
    S.width = c->cfg.width; S.height = c->cfg.height;
    S.max_bounces = c->cfg.max_bounces; S.light_samples = c->cfg.light_samples;
    S.russian_roulette = c->cfg.russian_roulette; S.only_direct = c->cfg.only_direct_illumination;
    S.normal_offset = c->cfg.normal_offset;
    S.seed_k0 = mix_lowbias32(c->cfg.seed);
    S.seed_k1 = mix_lowbias32(c->cfg.seed + 0x632BE5ABu);
    return 0;
}

// ---- the launch plan (include/pathtrace_hip.h, ABI v6; measured in DESIGN.md 6) -----------------------------------
// A call of `npix` pixels x `total_spp` samples with `cap` path slots per batch: as few EQUAL batches as the slots allow,
// at least two (one batch's thin late bounces overlap nothing: a rank of N = 8 at K = 20 takes 18.6 ms in one batch, 17.6
// in two, 18.2 in three), a multiple of the lanes beyond the lanes (the last round of batches is a full one: N = 1,
// K = 20: 4 batches 133.2 ms, 6 batches 129.5).  Returns samples per batch; *n_batches = how many.
static int plan_batches(int64_t npix, int total_spp, int64_t cap, int lanes, int *n_batches)
{
    const int64_t spp_cap = std::max<int64_t>(1, cap / std::max<int64_t>(npix, 1));
    int64_t batches = std::max<int64_t>(2, (total_spp + spp_cap - 1) / spp_cap);
    if (batches > lanes) batches = (batches + lanes - 1) / lanes * lanes;
    const int64_t want = (total_spp + batches - 1) / batches;   // equal batches (the last may be a few samples short)
    const int spp = (int)std::max<int64_t>(1, std::min<int64_t>(spp_cap, want));
    if (n_batches) *n_batches = (total_spp + spp - 1) / spp;
    return spp;
}
extern "C" int32_t pt_plan_batches(int64_t pixels, int32_t samples, int64_t path_slots, int32_t lanes, int32_t *batches)
{
    int nb = 0;
    const int spp = plan_batches(std::max<int64_t>(pixels, 1), std::max(samples, 1), std::max<int64_t>(path_slots, 1), std::max(lanes, 1), &nb);
    if (batches) *batches = nb;
    return spp;
}
static size_t slot_bytes(const pt_ctx *c) { return 192 + 24 * (size_t)std::max(c->cfg.light_samples, 1); }   // pt_device.h DQueue x 2, DShadowQueue, hit, radiance, pending

// What is shared by the lanes or does not depend on the number of path slots: framebuffer, counters, the lanes' HIP
// streams and events, the live-count words, the general sweep's global stack.
static int alloc_fixed(pt_ctx *c)
{
    if ((int64_t)c->cfg.width * c->cfg.height > ((int64_t)1 << 30)) {
        set_err("pt_create: film of %d x %d pixels exceeds 2^30", c->cfg.width, c->cfg.height);
        return -1;
    }
    c->seg_cap = 4096;   // measured on cornell_box 1080p: 1024 -2.4 %, 2048 -1.4 %, 4096 best, 8192 / 16384 -0.3 %
    if (dev_alloc(c, &c->fb_own, (size_t)c->cfg.width * c->cfg.height)) return -1;
    DCounters *ctr = nullptr;
    if (dev_alloc(c, &ctr, PT_COUNTER_BANKS)) return -1;
    HIP_TRY(hipMemset(c->fb_own, 0, sizeof(float4) * (size_t)c->cfg.width * c->cfg.height));
    HIP_TRY(hipMemset(ctr, 0, sizeof(DCounters) * PT_COUNTER_BANKS));
    // every lane owns a full set of stream buffers.  Measured on cornell_box 1080p (round 2): 1 lane 23.9, 2 lanes 28.9,
    // 3 lanes 29.3, 4 lanes 27.9 Grays/s -- one lane leaves the thin late bounces exposed, with three the chip always has a
    // wide kernel to run
    const char *env = getenv("PATHTRACE_HIP_LANES");
    c->n_lanes = env ? std::max(1, std::min(PT_MAX_LANES, atoi(env))) : 3;
    c->n_lanes_alloc = c->n_lanes;
    for (int l = 0; l < c->n_lanes; l++) {
        Lane &ln = c->lanes[l];
        DStreams &st = ln.st;
        if (dev_alloc(c, &st.qmax, (size_t)launch_qmax_words())) return -1;
        st.gstack = nullptr;
        st.gstack_stride = 0;
        if (c->S.stack_depth > PT_MAX_STACK) {   // deeper than the LDS short stack: a global one per lane (kernels of two lanes overlap)
            st.gstack_stride = launch_grid_max() * 256;
            if (dev_alloc(c, &st.gstack, (size_t)c->S.stack_depth * 4 * (size_t)st.gstack_stride)) return -1;
        }
        st.fb = c->fb_own;
        st.counters = ctr;
        if (l == 0) ln.stream = c->own_stream;
        else HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ln.acc_done, hipEventDisableTiming));
    }
    c->st = c->lanes[0].st;
    HIP_TRY(hipHostMalloc((void **)&c->host_ctr, sizeof(DCounters) * PT_COUNTER_BANKS));
    memset(c->host_ctr, 0, sizeof(DCounters) * PT_COUNTER_BANKS);
    return 0;
}

template <typename T>
static int stream_alloc(pt_ctx *c, T **p, size_t n)
{
    void *q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    c->stream_allocs.push_back(q);
    *p = (T *)q;
    return 0;
}

// The wavefront streams of every lane for `want` path slots per batch.  Called by pt_create (an explicit
// max_paths_in_flight), by pt_reserve and by the first render call of an auto-sized context -- and again when a later call's
// plan wants more slots than the context owns: every lane goes idle, the old streams are freed, the new ones allocated.
static int size_streams_once(pt_ctx *c, int64_t want)
{
    want = std::max<int64_t>(want, 64);
    // a path slot travels as an int32 whose bit 31 is the `pending` flag (k_shade / k_connect), segment counts and pixel
    // counts are int32: refuse what would overflow them instead of corrupting slot ids
    if (want > ((int64_t)1 << 30)) {
        set_err("pt_create: max_paths_in_flight = %lld exceeds 2^30 path slots per batch", (long long)want);
        return -1;
    }
    if (!c->stream_allocs.empty()) {
        for (int l = 0; l < c->n_lanes_alloc; l++) HIP_TRY(hipStreamSynchronize(c->lanes[l].stream));
        for (void *p : c->stream_allocs) (void)hipFree(p);
        c->stream_allocs.clear();
        c->have_last = false;
        c->plan.grown++;
    }
    int seg = c->seg_cap;
    { std::string sv; if (env_token("PATHTRACE_HIP_PLAN", "seg", &sv)) { const int v = atoi(sv.c_str()); if (v >= 256 && v <= (1 << 20) && v % 256 == 0) seg = v; } }
    c->seg_cap = seg;
    c->n_seg_max = (int)((want + c->seg_cap - 1) / c->seg_cap);
    c->P = (int64_t)c->n_seg_max * c->seg_cap;
    c->n_seg_max += 2;
    c->P_phys = c->P + 2 * (int64_t)(c->seg_cap + 256);
    const size_t P = (size_t)c->P_phys;
    const size_t L = (size_t)std::max(c->cfg.light_samples, 1);
    size_t hfree = 0, htotal = 0;
    if (hipMemGetInfo(&hfree, &htotal) == hipSuccess) c->plan.hbm_free_bytes = (int64_t)hfree;
    auto salloc = [&](auto **p, size_t n) -> int { return stream_alloc(c, p, n); };
    for (int l = 0; l < c->n_lanes_alloc; l++) {
        DStreams &st = c->lanes[l].st;
        for (int i = 0; i < 2; i++) {
            if (salloc(&st.q[i].r0, P) || salloc(&st.q[i].r1, P) || salloc(&st.q[i].s0, P) || salloc(&st.q[i].s1, P) ||
                salloc(&st.q[i].count, (size_t)c->n_seg_max))
                return -1;
        }
        if (salloc(&st.sq.p0, P) || salloc(&st.sq.key, P) || salloc(&st.sq.d, P * L) || salloc(&st.sq.e, P * L) ||
            salloc(&st.sq.count, (size_t)c->n_seg_max))
            return -1;
        if (salloc(&st.hit, P) || salloc(&st.radiance, P) || salloc(&st.pending, P)) return -1;
    }
    {   // the caller's framebuffer (pt_set_device_framebuffer) stays the one the batches add to
        float4 *fb = c->st.fb;
        c->st = c->lanes[0].st;
        c->st.fb = fb;
    }
    c->plan.path_slots = c->P;
    c->plan.stream_bytes = (int64_t)(P * slot_bytes(c) * (size_t)c->n_lanes_alloc);
    return 0;
}

static void drop_streams(pt_ctx *c)
{
    for (void *p : c->stream_allocs) (void)hipFree(p);
    c->stream_allocs.clear();
    for (int l = 0; l < PT_MAX_LANES; l++) {   // no stale plane pointers: a lane without streams has none
        DStreams &st = c->lanes[l].st;
        for (int i = 0; i < 2; i++) st.q[i] = DQueue{};
        st.sq = DShadowQueue{};
        st.hit = nullptr; st.radiance = nullptr; st.pending = nullptr;
    }
    c->P = 0; c->P_phys = 0; c->n_seg_max = 0;
    c->plan.path_slots = 0; c->plan.stream_bytes = 0;
}
// An allocation that fails half-way leaves nothing behind; an auto-sized context then tries again with half the slots (another
// process may hold memory hipMemGetInfo did not show as taken yet), an explicit size fails loudly.
static int size_streams(pt_ctx *c, int64_t want)
{
    for (;;) {
        if (size_streams_once(c, want) == 0) return 0;
        const std::string why = g_err;
        (void)hipGetLastError();
        drop_streams(c);
        if (!c->auto_size || want <= ((int64_t)1 << 20)) { g_err = why; return -1; }
        want /= 2;
    }
}

// Path slots the plan wants for calls of `pixels` x `samples` on an auto-sized context: pixels x its samples per batch, within
// PT_PLAN_MAX_PATHS and PT_PLAN_HBM_FRACTION of the memory that is free now plus what the context's streams already hold.
static int64_t plan_slots(pt_ctx *c, int64_t pixels, int samples)
{
    size_t hfree = 0, htotal = 0;
    int64_t cap = PT_PLAN_MAX_PATHS;
    { std::string mv; if (env_token("PATHTRACE_HIP_PLAN", "max", &mv)) { const long long v = atoll(mv.c_str()); if (v >= 64) cap = v; } }   // measurement knob: PATHTRACE_HIP_PLAN=max=<paths>
    if (hipMemGetInfo(&hfree, &htotal) == hipSuccess) {
        const double avail = ((double)hfree + (double)c->plan.stream_bytes) * PT_PLAN_HBM_FRACTION;
        const int64_t by_mem = (int64_t)(avail / ((double)slot_bytes(c) * c->n_lanes_alloc)) - 2 * (int64_t)(c->seg_cap + 256);
        cap = std::max<int64_t>(64, std::min(cap, by_mem));
    }
    const int spp = plan_batches(pixels, samples, cap, c->n_lanes, nullptr);
    return std::min<int64_t>(cap, std::max<int64_t>(pixels, 1) * spp);
}

// Before a render call of `pixels` x `samples`: an auto-sized context gets (or grows to) the slots the plan wants.
static int ensure_streams(pt_ctx *c, int64_t pixels, int samples)
{
    if (!c->auto_size) return 0;
    const int64_t want = plan_slots(c, pixels, samples);
    if (c->stream_allocs.empty() || want > c->P) return size_streams(c, want);
    return 0;
}

extern "C" int pt_reserve(pt_ctx *c, int64_t pixels, int32_t samples)
{
    if (!c || pixels < 1 || samples < 1) { set_err("pt_reserve: bad argument"); return -1; }
    HIP_TRY(hipSetDevice(c->device));
    if (ensure_streams(c, pixels, samples)) return -1;
    int nb = 0;
    c->plan.pixels = pixels; c->plan.samples = samples;
    c->plan.spp_per_batch = plan_batches(pixels, samples, c->P, c->n_lanes, &nb);
    c->plan.batches = nb;
    c->plan.paths_per_batch = pixels * c->plan.spp_per_batch;
    return 0;
}

extern "C" int pt_get_plan(pt_ctx *c, pt_plan *out)
{
    if (!c || !out) { set_err("pt_get_plan: null argument"); return -1; }
    *out = c->plan;
    out->lanes = c->n_lanes;
    out->auto_sized = c->auto_size ? 1 : 0;
    return 0;
}

extern "C" pt_ctx *pt_create(const pt_scene_desc *scene, const pt_config *config)
{
    if (!scene || !config) { set_err("pt_create: null argument"); return nullptr; }
    if (config->width < 1 || config->height < 1 || config->max_bounces < 0 || config->light_samples < 1) {
        // light_samples = 0 makes the reference divide by zero (integrator.h:268) and trip its NaN assertion
        set_err("pt_create: bad config (width/height >= 1, max_bounces >= 0, light_samples >= 1 required)");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_err("pt_create: no HIP device available (this library has no CPU fallback)");
        return nullptr;
    }
    pt_ctx *c = new pt_ctx();
    c->cfg = *config;
    auto fail = [&]() { std::string e = g_err; pt_destroy(c); g_err = e; return (pt_ctx *)nullptr; };
    if (config->device >= 0) {
        if (hipSetDevice(config->device) != hipSuccess) { set_err("pt_create: hipSetDevice(%d) failed", config->device); return fail(); }
    }
    if (hipGetDevice(&c->device) != hipSuccess) { set_err("pt_create: hipGetDevice failed"); return fail(); }
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { set_err("pt_create: stream"); return fail(); }
    c->stream = c->own_stream;
    if (hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming) != hipSuccess) { set_err("pt_create: event"); return fail(); }
    if (build_scene(c, scene)) return fail();
    c->clock = new pt_ctx::Clock();
    if (alloc_fixed(c)) return fail();
    // max_paths_in_flight = 0: the streams are sized by the launch plan from the first render call / pt_reserve (ABI v6)
    c->auto_size = c->cfg.max_paths_in_flight <= 0;
    if (!c->auto_size && size_streams(c, c->cfg.max_paths_in_flight)) return fail();
    {   // the per-scene build of the sweep: PATHTRACE_HIP_SPEC = async (default: built on a thread of its own, used when ready),
        // sync (built before pt_create returns), off.  Scenes without a fast program keep the generic kernels.
        const bool off = env_token("PATHTRACE_HIP_SPEC", "off");
        std::string table;
        if (!off && !c->S.walk && spec_header_text(scene, table) > 0) {
            const int L = c->cfg.light_samples;
            const int spec_nr = (L % 2 == 0) ? 2 : 1;   // rays per sweep of the module's k_connect (4: measured slower, LOGBOOK)
            c->spec = spec_start(table, c->S.geom_all != 0, c->S.textured != 0, spec_nr, c->device, env_token("PATHTRACE_HIP_SPEC", "sync"));
        }
        g_err.clear();   // spec_header_text leaves a message for scenes it does not serve: not an error of pt_create
    }
    return c;
}

extern "C" void pt_destroy(pt_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (int l = 0; l < c->n_lanes_alloc; l++)
        if (c->lanes[l].stream) (void)hipStreamSynchronize(c->lanes[l].stream);
    for (int l = 0; l < c->n_lanes_alloc; l++) {
        if (c->lanes[l].acc_done) (void)hipEventDestroy(c->lanes[l].acc_done);
        if (l > 0 && c->lanes[l].stream) (void)hipStreamDestroy(c->lanes[l].stream);
    }
    spec_destroy(c->spec);
    for (void *p : c->stream_allocs) (void)hipFree(p);
    for (void *p : c->allocs) (void)hipFree(p);
    delete c->clock;   // every lane is idle (synchronised above): no host function of a render call is pending
    if (c->host_ctr) (void)hipHostFree(c->host_ctr);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto &m : c->marks) (void)hipEventDestroy(m.ev);
    for (auto e : c->mark_free) (void)hipEventDestroy(e);
    if (c->snap_host) (void)hipHostFree(c->snap_host);
    if (c->snap_stream) (void)hipStreamDestroy(c->snap_stream);
    if (c->done_ev) (void)hipEventDestroy(c->done_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

// sum of the counter banks in the pinned mirror (monotone words written by the device; a poll may see a mix of two
// consecutive copies, which is still a valid lower bound)
static const DCounters &sum_counters(pt_ctx *c)
{
    DCounters s{};
    unsigned long long *d = (unsigned long long *)&s;
    for (int b = 0; b < PT_COUNTER_BANKS; b++) {
        const unsigned long long *h = (const unsigned long long *)&c->host_ctr[b];
        for (int k = 0; k < PT_N_COUNTERS; k++) d[k] += h[k];
    }
    c->host_sum = s;
    return c->host_sum;
}

// batches accumulate in enqueue order (run_batch chains the k_accumulate launches), so the finished ones are a prefix
static void retire_marks(pt_ctx *c)
{
    while (!c->marks.empty() && hipEventQuery(c->marks.front().ev) == hipSuccess) {
        c->accumulated_samples = c->marks.front().cum_samples;
        c->mark_free.push_back(c->marks.front().ev);
        c->marks.pop_front();
    }
}

static hipEvent_t get_event(pt_ctx *c, size_t i)
{
    while (c->ev_pool.size() <= i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[i];
}

struct Timer {
    pt_ctx *c; int kind; bool on; hipStream_t stream; hipEvent_t a = nullptr, b = nullptr;
    Timer(pt_ctx *c_, int kind_, hipStream_t s_) : c(c_), kind(kind_), on(c_->profiling), stream(s_)
    {
        if (!on) return;
        size_t i = c->timed.size() * 2;
        a = get_event(c, i); b = get_event(c, i + 1);
        if (!a || !b) { on = false; return; }
        (void)hipEventRecord(a, stream);
    }
    ~Timer()
    {
        if (!on) return;
        (void)hipEventRecord(b, stream);
        c->timed.push_back({kind, a, b});
    }
};

// Multiplier of the order in which the persistent workgroups visit the n segments of a queue (pt_kernels.hip ChunkWalk):
// ~ n / golden ratio, the stride whose multiples mod n are the most evenly spread, made coprime to n so that k -> k * mul
// mod n is a permutation.  (Queue order, the A/B of round 3: rank times of an 8-way partition +-7 % instead of +-2.5 %.)
static int seg_perm(int n)
{
    if (n < 16 || n >= 65536) return 1;   // the kernels form k * mul in 32 bits
    long long m = llround((double)n * 0.6180339887498949);
    auto gcd = [](long long a, long long b) { while (b) { const long long t = a % b; a = b; b = t; } return a; };
    while (gcd(m, n) != 1) m++;
    return (int)(m % n);
}

static int run_batch(pt_ctx *c, const DBatch &b)
{
    const DScene &S = c->S;
    const int li = c->next_lane;
    c->next_lane = (c->next_lane + 1) % c->n_lanes;
    Lane &ln = c->lanes[li];
    DStreams st = ln.st;
    st.fb = c->st.fb;   // the (possibly caller-owned) framebuffer is shared
    hipStream_t sm = ln.stream;
    // PATHTRACE_HIP_TRACE_LAUNCH=1 (debugging a device fault): every launch is followed by a stream synchronisation and one
    // line on stderr, so the last line before an abort names the kernel, the bounce and the queue geometry it ran with
    static const bool trace = getenv("PATHTRACE_HIP_TRACE_LAUNCH") != nullptr;
    SpecJob *spec = (c->spec && spec_poll(c->spec) == 1) ? c->spec : nullptr;   // one decision per batch
    auto traced = [&](const char *what, int bounce, const DBatch &q) -> int {
        if (!trace) return 0;
        const hipError_t e = hipStreamSynchronize(sm);
        static auto last = std::chrono::steady_clock::now();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pt launch] lane %d %s bounce %d: n_seg %d x %d -> %d x %d, perm %d/%d, npix %d ns %d s0 %d tiles %d: %s, %.1f us since the previous line\n", li, what, bounce,
                q.n_seg, q.seg_cap, q.n_seg_out, q.seg_cap_out, q.perm, q.perm_out, q.npix, q.ns, q.s0, q.n_tiles, hipGetErrorString(e),
                std::chrono::duration<double, std::micro>(now - last).count());
        last = now;
        if (e != hipSuccess) { set_err("%s failed at bounce %d: %s", what, bounce, hipGetErrorString(e)); return -1; }
        return 0;
    };
    HIP_TRY(hipMemsetAsync(st.qmax, 0, sizeof(int32_t) * (size_t)launch_qmax_words(), sm));   // this batch's live-count maxima (chunk_limit)
    // camera rays are formed by bounce 0's own kernels (pt_kernels.hip PT_FUSE_GENERATE); k_generate only runs when no bounce does
    if (!launch_fuses_generate() || S.max_bounces == 0) {
        { Timer t(c, PT_K_GENERATE, sm); launch_generate(S, st, b, sm); }
        if (traced("k_generate", -1, b)) return -1;
    }
    int qi = 0;
    DBatch bb = b;
    for (int bounce = 0; bounce < S.max_bounces; bounce++) {
        // segments merge pairwise from one bounce to the next (pt_device.h DBatch) until one is left
        if (bb.n_seg > 1 && bb.n_seg > PT_MERGE_MIN_SEGMENTS) { bb.n_seg_out = (bb.n_seg + 1) / 2; bb.seg_cap_out = bb.seg_cap * 2; }
        else { bb.n_seg_out = bb.n_seg; bb.seg_cap_out = bb.seg_cap; }
        bb.perm = seg_perm(bb.n_seg); bb.perm_out = seg_perm(bb.n_seg_out);
        { Timer t(c, PT_K_EXTEND, sm); launch_extend(S, st, bb, qi, bounce, sm, spec); }
        if (traced("k_extend", bounce, bb)) return -1;
        { Timer t(c, PT_K_SHADE, sm); launch_shade(S, st, bb, qi, bounce, sm); }
        if (traced("k_shade", bounce, bb)) return -1;
        if (c->tally) launch_tally(S, st, bb, qi, bounce, c->tally + (b.tiles - c->d_tiles), sm);
        { Timer t(c, PT_K_CONNECT, sm); launch_connect(S, st, bb, bounce, sm, spec); }
        if (traced("k_connect", bounce, bb)) return -1;
        bb.n_seg = bb.n_seg_out; bb.seg_cap = bb.seg_cap_out;
        qi ^= 1;
    }
    // framebuffer[j][i] += col must happen in sample order per pixel (float addition is not associative): every
    // accumulate waits for the previous batch's accumulate, whichever lane that ran on
    if (c->last_lane >= 0 && c->last_lane != li) HIP_TRY(hipStreamWaitEvent(sm, c->lanes[c->last_lane].acc_done, 0));
    { Timer t(c, PT_K_ACCUMULATE, sm); launch_accumulate(S, st, b, sm); }
    if (traced("k_accumulate", -1, b)) return -1;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->host_ctr, st.counters, sizeof(DCounters) * PT_COUNTER_BANKS, hipMemcpyDeviceToHost, sm));
    HIP_TRY(hipEventRecord(ln.acc_done, sm));
    {   // batch mark for pt_snapshot_framebuffer; finished marks are retired here so the list stays short
        retire_marks(c);
        hipEvent_t ev = nullptr;
        if (!c->mark_free.empty()) { ev = c->mark_free.back(); c->mark_free.pop_back(); }
        else HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, sm));
        c->enqueued_samples += (uint64_t)b.n_paths;
        c->marks.push_back({ev, c->enqueued_samples});
    }
    c->last_lane = li;
    c->last_batch = b;
    c->last_batch_lane = li;
    c->have_last = true;
    return 0;
}

// One batch group = consecutive bands whose pixel total fits the path slots; rendered for every sample chunk.
static int render_group(pt_ctx *c, const std::vector<DTile> &bands, const std::vector<int> &band_h, size_t g0, size_t g1,
                        size_t tile_off, int32_t spp_begin, int32_t spp_end)
{
    int64_t npix = 0;
    for (size_t i = g0; i < g1; i++) npix += (int64_t)bands[i].w * band_h[i];
    // what the kernels index with: every batch-local pixel and every path slot of a group stays inside the lane's P slots
    // and the group's entries of the tile table (k_generate / batch_pixel / bounce0_k1 take these on trust)
    if (npix < 1 || npix > c->P || tile_off + (g1 - g0) + 1 > c->h_tiles.size()) {
        set_err("pt_render_tiles_async: internal error: batch group of %lld pixels, %zu bands at table offset %zu (P = %lld, table %zu)",
                (long long)npix, g1 - g0, tile_off, (long long)c->P, c->h_tiles.size());
        return -1;
    }
    // the launch plan (plan_batches): equal batches, as few as the slots allow, at least two, a multiple of the lanes
    // (PATHTRACE_HIP_PLAN=caller, a measurement knob: the caller's calls ARE the batches, cut only where the slots force it --
    // the rule up to ABI v5 -- so that tools/plan_probe.py can time plans other than the library's)
    static const bool caller_plan = env_token("PATHTRACE_HIP_PLAN", "caller");
    int n_batches = 0;
    int ns_plan = plan_batches(npix, spp_end - spp_begin, c->P, c->n_lanes, &n_batches);
    if (caller_plan) { ns_plan = (int)std::max<int64_t>(1, c->P / npix); n_batches = (spp_end - spp_begin + ns_plan - 1) / ns_plan; }
    if (g0 == 0) {
        c->plan.pixels = npix; c->plan.samples = spp_end - spp_begin; c->plan.spp_per_batch = ns_plan;
        c->plan.batches = n_batches; c->plan.paths_per_batch = npix * ns_plan;
    }
    for (int s = spp_begin; s < spp_end;) {
        const int ns = std::min(ns_plan, spp_end - s);
        DBatch b{};
        b.x0 = bands[g0].x0; b.y0 = bands[g0].y0; b.w = bands[g0].w;
        b.n_tiles = (int)(g1 - g0);
        b.tiles = c->d_tiles + tile_off;
        b.npix = (int)npix;
        b.s0 = s; b.ns = ns;
        // Workgroups take the SAME chunk of every segment at the same time (chunk-major order): the addresses in flight are
        // {g * seg_cap * 16 B}.  When a sample's pixels are a multiple of 2^16 -- e.g. 16 tiles of 128 x 128 -- the live
        // segments (the pixels over the scene) repeat with a power-of-two period from sample to sample and their addresses
        // fall on a fraction of the HBM channels: measured 45.5 against 28.3 ms for 2^18 pixels x 54 spp.  One more chunk
        // per segment breaks the period.  (PATHTRACE_HIP_PLAN=seg=N fixes the size for measurements and tests.)
        const bool seg_forced = env_token("PATHTRACE_HIP_PLAN", "seg");
        b.seg_cap = (npix % 65536 == 0 && !seg_forced) ? c->seg_cap + 256 : c->seg_cap;
        b.n_paths = npix * ns;
        b.n_seg = (int)((b.n_paths + b.seg_cap - 1) / b.seg_cap);
        b.n_seg_out = b.n_seg; b.seg_cap_out = b.seg_cap;
        b.perm = b.perm_out = 1;
        b.P = c->P_phys;
        if ((int64_t)b.n_seg * b.seg_cap > c->P_phys || b.n_seg > c->n_seg_max) { set_err("pt_render_tiles_async: internal error: %d segments of %d slots exceed the streams", b.n_seg, b.seg_cap); return -1; }
        // the chunk sort of k_shade costs two workgroup barriers per chunk; measured it pays where a hit is
        // expensive and uneven -- textures, or more than two lights (per-lane light records, sphere lights) -- and not on
        // the one- and two-light Cornell scenes.  PATHTRACE_HIP_SHADE=sort / nosort force it for measurements and tests.
        b.sort_shade = (c->S.textured || c->S.n_lights > 2) ? 1 : 0;
        if (env_token("PATHTRACE_HIP_SHADE", "sort")) b.sort_shade = 1;
        if (env_token("PATHTRACE_HIP_SHADE", "nosort")) b.sort_shade = 0;
        b.stage_shadow = (c->S.light_samples >= 1 && c->S.light_samples <= PT_STAGE_MAX_SAMPLES && !env_token("PATHTRACE_HIP_SHADE", "nostage")) ? 1 : 0;
        if (run_batch(c, b)) return -1;
        s += ns;
    }
    return 0;
}

static int render_tiles(pt_ctx *c, int32_t n_rects, const int32_t *rects, int32_t spp_begin, int32_t spp_end, bool tally)
{
    if (!c || !rects || n_rects < 1) { set_err("pt_render_tiles_async: bad argument"); return -1; }
    if (spp_begin < 0 || spp_end <= spp_begin) { set_err("pt_render_tiles_async: bad sample range [%d,%d)", spp_begin, spp_end); return -1; }
    HIP_TRY(hipSetDevice(c->device));
    const auto t_entry = std::chrono::steady_clock::now();
    {   // an auto-sized context gets / grows its streams for this call (include/pathtrace_hip.h "the launch plan")
        int64_t total = 0;
        for (int r = 0; r < n_rects; r++) total += (int64_t)std::max(rects[4 * r + 2] - rects[4 * r], 0) * std::max(rects[4 * r + 3] - rects[4 * r + 1], 0);
        if (ensure_streams(c, std::max<int64_t>(total, 1), spp_end - spp_begin)) return -1;
    }
    // rects -> bands of at most P pixels each
    std::vector<DTile> bands;
    std::vector<int> band_h;
    c->band_rect.clear();
    c->band_table.clear();
    for (int r = 0; r < n_rects; r++) {
        const int x0 = rects[4 * r], y0 = rects[4 * r + 1], x1 = rects[4 * r + 2], y1 = rects[4 * r + 3];
        if (x0 < 0 || y0 < 0 || x1 > c->cfg.width || y1 > c->cfg.height || x0 >= x1 || y0 >= y1) {
            set_err("pt_render_tiles_async: rect %d = [%d,%d)x[%d,%d) is empty or outside the %dx%d film", r, x0, x1, y0, y1,
                    c->cfg.width, c->cfg.height);
            return -1;
        }
        const int w = x1 - x0;
        if (w > c->P) { set_err("pt_render_tiles_async: rect width %d exceeds max_paths_in_flight", w); return -1; }
        const int rows_fit = (int)std::max<int64_t>(1, c->P / w);
        for (int yb = y0; yb < y1; yb += rows_fit) {
            const int hb = std::min(rows_fit, y1 - yb);
            bands.push_back(DTile{x0, yb, w, 0});
            band_h.push_back(hb);
            c->band_rect.push_back(r);
        }
    }
    // greedy groups of consecutive bands with <= P pixels; tile table = per group [bands..., sentinel]
    std::vector<DTile> table;
    struct Group { size_t g0, g1, off; };
    std::vector<Group> groups;
    for (size_t i = 0; i < bands.size();) {
        int64_t npix = 0;
        size_t j = i;
        const size_t off = table.size();
        while (j < bands.size() && npix + (int64_t)bands[j].w * band_h[j] <= c->P) {
            DTile t = bands[j];
            t.pix0 = (int)npix;
            c->band_table.push_back(table.size());
            table.push_back(t);
            npix += (int64_t)bands[j].w * band_h[j];
            j++;
        }
        table.push_back(DTile{0, 0, 1, (int)npix});
        groups.push_back({i, j, off});
        i = j;
    }
    // upload the table unless it is the one already on the device (the usual case: one call per sample chunk)
    const bool same = table.size() == c->h_tiles.size() &&
                      (table.empty() || memcmp(table.data(), c->h_tiles.data(), table.size() * sizeof(DTile)) == 0);
    if (!same) {
        for (int l = 0; l < c->n_lanes_alloc; l++) HIP_TRY(hipStreamSynchronize(c->lanes[l].stream));   // old table may be in use
        if (table.size() > c->d_tiles_cap) {
            if (dev_alloc(c, &c->d_tiles, table.size() * 2)) return -1;
            c->d_tiles_cap = table.size() * 2;
        }
        HIP_TRY(hipMemcpy(c->d_tiles, table.data(), table.size() * sizeof(DTile), hipMemcpyHostToDevice));
        c->h_tiles = table;
    }
    c->tally = nullptr;
    if (tally) {   // pt_measure_tile_costs: one zeroed cost word per table entry, added to by k_tally after every k_extend
        for (int l = 0; l < c->n_lanes_alloc; l++) HIP_TRY(hipStreamSynchronize(c->lanes[l].stream));
        if (table.size() > c->tally_cap) {
            if (dev_alloc(c, &c->tally_buf, table.size() * 2)) return -1;
            c->tally_cap = table.size() * 2;
        }
        c->tally = c->tally_buf;
        HIP_TRY(hipMemset(c->tally, 0, table.size() * sizeof(unsigned long long)));
    }
    for (const Group &g : groups)
        if (render_group(c, bands, band_h, g.g0, g.g1, g.off, spp_begin, spp_end)) { c->tally = nullptr; return -1; }
    // the last accumulate transitively waited for every earlier accumulate, i.e. for every earlier batch
    HIP_TRY(hipEventRecord(c->done_ev, c->lanes[c->last_lane].stream));
    {   // the call's clock: a host function behind the last batch stamps the time the device finished (pt_render_seconds, pt_wait_for)
        {
            std::lock_guard<std::mutex> g(c->clock->mu);
            c->clock->t_begin = t_entry;
            c->clock->issued++;
        }
        HIP_TRY(hipLaunchHostFunc(c->lanes[c->last_lane].stream, [](void *p) {
            pt_ctx::Clock *k = (pt_ctx::Clock *)p;
            std::lock_guard<std::mutex> g(k->mu);
            k->t_end = std::chrono::steady_clock::now();
            k->landed++;
            k->cv.notify_all();
        }, c->clock));
    }
    return 0;
}

extern "C" double pt_render_seconds(pt_ctx *c)
{
    if (!c || !c->clock) return -1.0;
    std::lock_guard<std::mutex> g(c->clock->mu);
    if (c->clock->issued == 0 || c->clock->landed != c->clock->issued) return -1.0;
    return std::chrono::duration<double>(c->clock->t_end - c->clock->t_begin).count();
}

// pt_multi.cpp: when the last render call of this context landed (false while it is running or before any call)
bool pt_clock_end(pt_ctx *c, std::chrono::steady_clock::time_point *t_end)
{
    if (!c || !c->clock) return false;
    std::lock_guard<std::mutex> g(c->clock->mu);
    if (c->clock->issued == 0 || c->clock->landed != c->clock->issued) return false;
    *t_end = c->clock->t_end;
    return true;
}

extern "C" int pt_wait_for(pt_ctx *c, int32_t timeout_ms)
{
    if (!c || !c->clock) { set_err("pt_wait_for: null ctx"); return -1; }
    std::unique_lock<std::mutex> g(c->clock->mu);
    const bool idle = c->clock->cv.wait_for(g, std::chrono::milliseconds(std::max(timeout_ms, 0)), [&] { return c->clock->landed == c->clock->issued; });
    return idle ? 1 : 0;
}

extern "C" int pt_render_tiles_async(pt_ctx *c, int32_t n_rects, const int32_t *rects, int32_t spp_begin, int32_t spp_end)
{
    return render_tiles(c, n_rects, rects, spp_begin, spp_end, false);
}

extern "C" int pt_wait(pt_ctx *c);
extern "C" int pt_clear_framebuffer(pt_ctx *c);
// Warm-up before a timed render (include/pathtrace_hip.h): one-sample passes over the rects, every lane, at least min_ms.
extern "C" int pt_prime(pt_ctx *c, int32_t n_rects, const int32_t *rects, int32_t min_ms)
{
    if (!c || n_rects < 0 || (n_rects > 0 && !rects)) { set_err("pt_prime: bad argument"); return -1; }
    const int32_t whole[4] = {0, 0, c->cfg.width, c->cfg.height};
    if (n_rects == 0) { n_rects = 1; rects = whole; }
    const pt_plan keep = c->plan;
    const auto t0 = std::chrono::steady_clock::now();
    do {
        for (int l = 0; l < c->n_lanes; l++)
            if (render_tiles(c, n_rects, rects, 0, 1, false)) return -1;
        if (pt_wait(c)) return -1;
    } while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < (double)min_ms);
    {   // the plan a caller reads back is the one of its own calls, not of the warm-up's
        pt_plan now = c->plan;
        c->plan = keep;
        c->plan.path_slots = now.path_slots; c->plan.stream_bytes = now.stream_bytes; c->plan.hbm_free_bytes = now.hbm_free_bytes; c->plan.grown = now.grown;
    }
    return pt_clear_framebuffer(c);
}
// The planner of a tile-partitioned render: the reference's ray count (pt_counters::rays: extension rays +
// light_samples shadow rays per hit, integrator.h:192, 246-247 -- what a tile's time follows, pt_kernels.hip k_tally) for `spp`
// samples per pixel of every rect, counted per rect
// in ONE pass over all of them -- the rects are rendered together as ordinary wavefront batches and k_tally attributes
// every bounce's rays to the rect the path's pixel lies in.  Leaves framebuffer and counters cleared.  Deterministic: the
// RNG is keyed by pixel and sample.
extern "C" int pt_measure_tile_costs(pt_ctx *c, int32_t n_rects, const int32_t *rects, int32_t spp, uint64_t *rays_out)
{
    if (!c || !rects || n_rects < 1 || spp < 1 || !rays_out) { set_err("pt_measure_tile_costs: bad argument"); return -1; }
    if (pt_clear_framebuffer(c)) return -1;
    int rc = render_tiles(c, n_rects, rects, 0, spp, true);
    unsigned long long *dev = c->tally;
    if (!rc) rc = pt_wait(c);
    c->tally = nullptr;
    if (rc) return -1;
    std::vector<unsigned long long> host(c->h_tiles.size(), 0);
    HIP_TRY(hipMemcpy(host.data(), dev, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int r = 0; r < n_rects; r++) rays_out[r] = 0;
    for (size_t k = 0; k < c->band_rect.size(); k++) rays_out[c->band_rect[k]] += host[c->band_table[k]];
    return pt_clear_framebuffer(c);
}

extern "C" int pt_render_async(pt_ctx *c, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t spp_begin, int32_t spp_end)
{
    if (!c) { set_err("pt_render_async: null ctx"); return -1; }
    const int32_t r[4] = {x0, y0, x1, y1};
    return pt_render_tiles_async(c, 1, r, spp_begin, spp_end);
}

extern "C" int pt_poll(pt_ctx *c, uint64_t *samples_done, uint64_t *rays_done)
{
    if (!c) { set_err("pt_poll: null ctx"); return -1; }
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = hipEventQuery(c->done_ev);
    const DCounters &hs = sum_counters(c);
    if (samples_done) *samples_done = hs.camera_samples;
    if (rays_done) *rays_done = hs.rays;
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    set_err("pt_poll: %s", hipGetErrorString(e));
    return -1;
}

static int collect_times(pt_ctx *c)
{
    if (!c->profiling || c->timed.empty()) return 0;
    pt_kernel_times kt{};
    for (auto &t : c->timed) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
        kt.launches[t.kind]++;
        kt.ms[t.kind] += ms;
    }
    const DCounters n = sum_counters(c), &o = c->ctr_at_profile_start;
    kt.units[PT_K_GENERATE] = n.camera_samples - o.camera_samples;
    kt.units[PT_K_EXTEND] = n.ext_rays - o.ext_rays;
    kt.units[PT_K_SHADE] = n.ext_rays - o.ext_rays;
    kt.units[PT_K_CONNECT] = n.shadow_rays - o.shadow_rays;
    kt.units[PT_K_ACCUMULATE] = n.camera_samples - o.camera_samples;
    // accumulate across render calls since pt_set_profiling(1)
    for (int k = 0; k < PT_N_KERNELS; k++) {
        c->ktimes.launches[k] += kt.launches[k];
        c->ktimes.ms[k] += kt.ms[k];
        c->ktimes.units[k] += kt.units[k];
    }
    c->ctr_at_profile_start = n;
    c->timed.clear();
    return 0;
}

extern "C" int pt_wait(pt_ctx *c)
{
    if (!c) { set_err("pt_wait: null ctx"); return -1; }
    HIP_TRY(hipSetDevice(c->device));   // every entry point below starts with pt_wait: the context's device is current after it
    for (int l = 0; l < c->n_lanes_alloc; l++) HIP_TRY(hipStreamSynchronize(c->lanes[l].stream));
    return collect_times(c);
}

extern "C" int pt_read_framebuffer(pt_ctx *c, float *rgb_sum)
{
    if (!c || !rgb_sum) { set_err("pt_read_framebuffer: null argument"); return -1; }
    if (pt_wait(c)) return -1;
    const size_t n = (size_t)c->cfg.width * c->cfg.height;
    std::vector<float4> tmp(n);
    HIP_TRY(hipMemcpy(tmp.data(), c->st.fb, n * sizeof(float4), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        rgb_sum[3 * i + 0] = tmp[i].x;
        rgb_sum[3 * i + 1] = tmp[i].y;
        rgb_sum[3 * i + 2] = tmp[i].z;
    }
    return 0;
}

extern "C" int pt_clear_framebuffer(pt_ctx *c)
{
    if (!c) { set_err("pt_clear_framebuffer: null ctx"); return -1; }
    if (pt_wait(c)) return -1;
    HIP_TRY(hipMemsetAsync(c->st.fb, 0, sizeof(float4) * (size_t)c->cfg.width * c->cfg.height, c->stream));
    HIP_TRY(hipMemsetAsync(c->st.counters, 0, sizeof(DCounters) * PT_COUNTER_BANKS, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->last_lane = -1;
    memset(c->host_ctr, 0, sizeof(DCounters) * PT_COUNTER_BANKS);
    c->ctr_at_profile_start = DCounters{};
    retire_marks(c);
    c->enqueued_samples = c->accumulated_samples = 0;
    return 0;
}

// Progressive preview: the reference's sync_progress (renderer.h:605-620) reads the framebuffer while its worker
// threads are still adding to it.  Here a copy engine reads the live framebuffer on a stream of its own while the
// render streams keep running; `samples_accumulated` is taken BEFORE the copy starts, so every pixel holds at least
// the batches counted (a pixel may already include the next batch -- the same looseness as the reference's).
extern "C" int pt_snapshot_framebuffer(pt_ctx *c, float *rgb_sum, uint64_t *samples_accumulated)
{
    if (!c || !rgb_sum) { set_err("pt_snapshot_framebuffer: null argument"); return -1; }
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->cfg.width * c->cfg.height;
    if (!c->snap_stream) HIP_TRY(hipStreamCreateWithFlags(&c->snap_stream, hipStreamNonBlocking));
    if (!c->snap_host) HIP_TRY(hipHostMalloc((void **)&c->snap_host, n * sizeof(float4)));
    retire_marks(c);
    const uint64_t done = c->accumulated_samples;
    HIP_TRY(hipMemcpyAsync(c->snap_host, c->st.fb, n * sizeof(float4), hipMemcpyDeviceToHost, c->snap_stream));
    HIP_TRY(hipStreamSynchronize(c->snap_stream));
    for (size_t i = 0; i < n; i++) {
        rgb_sum[3 * i + 0] = c->snap_host[i].x;
        rgb_sum[3 * i + 1] = c->snap_host[i].y;
        rgb_sum[3 * i + 2] = c->snap_host[i].z;
    }
    if (samples_accumulated) *samples_accumulated = done;
    return 0;
}

extern "C" int pt_get_counters(pt_ctx *c, pt_counters *out)
{
    if (!c || !out) { set_err("pt_get_counters: null argument"); return -1; }
    if (pt_wait(c)) return -1;
    const DCounters &d = sum_counters(c);
    out->camera_samples = d.camera_samples;
    out->rays = d.rays; out->extension_rays = d.ext_rays; out->extension_hits = d.ext_hits; out->shadow_rays = d.shadow_rays;
    out->term_miss = d.term_miss; out->term_rr = d.term_rr; out->term_emitter = d.term_emitter;
    out->term_pdf = d.term_pdf; out->term_bounce_limit = d.term_bounce_limit;
    out->shadow_rays_traced = d.shadow_rays - d.shadow_untraced;
    out->rays_traced = d.rays - d.shadow_untraced;
    return 0;
}

extern "C" int pt_trace_rays(pt_ctx *c, int64_t n, int32_t nr, const float *origins, const float *dirs, uint32_t k0, uint32_t k1,
                             uint32_t vol_dim, float *t_out, int32_t *id_out)
{
    if (!c || n < 1 || (nr != 1 && nr != 2 && nr != 4) || !origins || !dirs || !t_out || !id_out) { set_err("pt_trace_rays: bad argument"); return -1; }
    if (pt_wait(c)) return -1;
    float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    int *d_i = nullptr;
    const size_t nray = (size_t)n * nr;
    int rc = -1;
    do {
        if (hipMalloc((void **)&d_o, (size_t)n * 12) != hipSuccess || hipMalloc((void **)&d_d, nray * 12) != hipSuccess ||
            hipMalloc((void **)&d_t, nray * 4) != hipSuccess || hipMalloc((void **)&d_i, nray * 4) != hipSuccess) { set_err("pt_trace_rays: hipMalloc failed"); break; }
        if (hipMemcpy(d_o, origins, (size_t)n * 12, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_d, dirs, nray * 12, hipMemcpyHostToDevice) != hipSuccess) { set_err("pt_trace_rays: upload failed"); break; }
        launch_trace(c->S, c->lanes[0].st, n, nr, d_o, d_d, k0, k1, vol_dim, d_t, d_i, c->stream, (c->spec && spec_poll(c->spec) == 1) ? c->spec : nullptr);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) { set_err("pt_trace_rays: kernel failed"); break; }
        if (hipMemcpy(t_out, d_t, nray * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(id_out, d_i, nray * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err("pt_trace_rays: download failed"); break; }
        rc = 0;
    } while (0);
    (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_t); (void)hipFree(d_i);
    return rc;
}

extern "C" void *pt_device_framebuffer(pt_ctx *c) { return c ? (void *)c->st.fb : nullptr; }
extern "C" int pt_set_device_framebuffer(pt_ctx *c, void *device_rgba, size_t bytes)
{
    if (!c) { set_err("pt_set_device_framebuffer: null ctx"); return -1; }
    if (pt_wait(c)) return -1;
    if (!device_rgba) { c->st.fb = c->fb_own; c->fb_external = false; return 0; }
    if (bytes < sizeof(float4) * (size_t)c->cfg.width * c->cfg.height) {
        set_err("pt_set_device_framebuffer: buffer too small (%zu bytes)", bytes);
        return -1;
    }
    c->st.fb = (float4 *)device_rgba;
    c->fb_external = true;
    return 0;
}
extern "C" void *pt_get_stream(pt_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int pt_set_stream(pt_ctx *c, void *s)
{
    if (!c) { set_err("pt_set_stream: null ctx"); return -1; }
    if (pt_wait(c)) return -1;
    // a caller-owned stream means the caller orders the work: run everything on it, one lane; handing the stream back
    // (NULL) restores the context's own lanes.  pt_wait above left every lane idle, so no batch order is pending.
    c->stream = s ? (hipStream_t)s : c->own_stream;
    c->lanes[0].stream = c->stream;
    c->n_lanes = s ? 1 : c->n_lanes_alloc;
    c->next_lane = 0;
    c->last_lane = -1;
    return 0;
}
// Batches rotate over `n` of the context's lanes from now on (1 .. the lanes it owns; PATHTRACE_HIP_LANES at pt_create).
// n = 1 serialises the kernels of consecutive batches: what a per-kernel measurement needs (bench.py's roofline pass).
extern "C" int pt_set_lanes(pt_ctx *c, int32_t n)
{
    if (!c) { set_err("pt_set_lanes: null ctx"); return -1; }
    if (pt_wait(c)) return -1;
    if (n < 1 || n > c->n_lanes_alloc) { set_err("pt_set_lanes: %d outside 1..%d", n, c->n_lanes_alloc); return -1; }
    if (c->stream != c->own_stream && n != 1) { set_err("pt_set_lanes: a caller-owned stream runs one lane"); return -1; }
    c->n_lanes = n;
    c->next_lane = 0;
    return c->n_lanes_alloc;
}
// The per-scene build of the traversal sweep (pt_spec.cpp).  Status: 1 = the context launches the scene's own k_extend /
// k_connect / k_trace, 0 = still building (generic kernels meanwhile), -1 = not available (PATHTRACE_HIP_SPEC=off, no
// hiprtc, a scene the build does not serve, a failed build: pt_last_error has the reason).  pt_spec_wait blocks until
// the build has ended.  The image does not depend on it: both forms perform the same arithmetic.
extern "C" int pt_spec_status(pt_ctx *c)
{
    if (!c) { set_err("pt_spec_status: null ctx"); return -1; }
    if (!c->spec) { set_err("no per-scene build for this context"); return -1; }
    HIP_TRY(hipSetDevice(c->device));
    const int st = spec_poll(c->spec);
    if (st < 0) set_err("per-scene build: %s", spec_log(c->spec));
    return st;
}
extern "C" int pt_spec_wait(pt_ctx *c)
{
    if (!c) { set_err("pt_spec_wait: null ctx"); return -1; }
    if (!c->spec) { set_err("no per-scene build for this context"); return -1; }
    HIP_TRY(hipSetDevice(c->device));
    const int st = spec_wait(c->spec);
    if (st < 0) set_err("per-scene build: %s", spec_log(c->spec));
    return st;
}
// Who compiled the module this context launches: one line of JSON (pt_spec.cpp spec_info) -- "built_by": "helper" (the compile
// helper beside the library, this toolchain's hiprtc) or "in-process" (whatever libhiprtc the host process resolves), the path
// of that libhiprtc, the producer string of the code object and whether it is the compiler that built this library.  Returns
// the text length; buf receives it when cap > length.
extern "C" int pt_spec_info(pt_ctx *c, char *buf, size_t cap)
{
    if (!c) { set_err("pt_spec_info: null ctx"); return -1; }
    const std::string t = c->spec ? spec_info(c->spec) : std::string("{\"status\": -1, \"built_by\": null, \"note\": \"no per-scene build for this context\"}");
    if (buf && cap > t.size()) memcpy(buf, t.c_str(), t.size() + 1);
    return (int)t.size();
}
// Host-only check of the per-scene build (no device): compiles the module for gfx950 and returns the size of its code
// object, < 0 on failure (pt_last_error has the compiler's log).
static long spec_check(const pt_scene_desc *scene, int32_t light_samples, std::string *info)
{
    std::string table, log;
    if (spec_header_text(scene, table) <= 0) return -2;
    HostProgram hp;
    if (build_program(scene, hp)) return -2;
    bool textured = scene->background_texture >= 0;
    for (int i = 0; i < scene->n_materials; i++) {
        const pt_material &m = scene->materials[i];
        textured |= m.texture >= 0 && (m.type == PT_MAT_LAMBERTIAN || m.type == PT_MAT_DIFFUSE_LIGHT || m.type == PT_MAT_ISOTROPIC);
    }
    const long n = spec_build_check(table, hp.geom_all != 0, textured, (light_samples % 2 == 0) ? 2 : 1, log, info);
    if (n < 0) set_err("per-scene build: %s", log.c_str());
    return n;
}
extern "C" long pt_spec_build_check(const pt_scene_desc *scene, int32_t light_samples) { return spec_check(scene, light_samples, nullptr); }
// The same check, answering with pt_spec_info's line for the module it built (host only): which compiler a context of this
// scene would get in this process.  Returns the text length (buf receives it when cap > length), < 0 when the build failed.
extern "C" int pt_spec_build_info(const pt_scene_desc *scene, int32_t light_samples, char *buf, size_t cap)
{
    std::string info;
    if (spec_check(scene, light_samples, &info) < 0) return -1;
    if (buf && cap > info.size()) memcpy(buf, info.c_str(), info.size() + 1);
    return (int)info.size();
}
extern "C" int pt_set_profiling(pt_ctx *c, int enabled)
{
    if (!c) { set_err("pt_set_profiling: null ctx"); return -1; }
    if (pt_wait(c)) return -1;
    c->profiling = enabled != 0;
    c->ktimes = pt_kernel_times{};
    c->timed.clear();
    c->ctr_at_profile_start = sum_counters(c);
    return 0;
}
extern "C" int pt_get_kernel_times(pt_ctx *c, pt_kernel_times *out)
{
    if (!c || !out) { set_err("pt_get_kernel_times: null argument"); return -1; }
    if (pt_wait(c)) return -1;
    *out = c->ktimes;
    return 0;
}
extern "C" int pt_read_last_batch_radiance(pt_ctx *c, float *rgba, size_t max_records, size_t *n_records)
{
    if (!c || !rgba || !n_records) { set_err("pt_read_last_batch_radiance: null argument"); return -1; }
    if (pt_wait(c)) return -1;
    if (!c->have_last) { *n_records = 0; return 0; }
    size_t n = std::min<size_t>((size_t)c->last_batch.n_paths, max_records);
    HIP_TRY(hipMemcpy(rgba, c->lanes[c->last_batch_lane].st.radiance, n * sizeof(float4), hipMemcpyDeviceToHost));
    *n_records = n;
    return 0;
}
