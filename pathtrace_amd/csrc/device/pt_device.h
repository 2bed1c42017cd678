// Device-side scene tables and wavefront stream descriptors shared by the host code that fills them
// (pt_context.cpp) and the gfx950 kernels (pt_kernels.hip).  Plain structs; HIP only for float4/uint2.
#pragma once
#ifdef __HIPCC_RTC__
// hiprtc (the per-scene build, pt_spec.cpp) keeps the fixed-width integers in a namespace of its own
typedef signed int int32_t;
typedef unsigned int uint32_t;
typedef signed long long int64_t;
typedef unsigned long long uint64_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace ptd {

// ---- scene tables (a few hundred bytes for the Cornell scenes; read with wave-uniform indices during
// traversal, so the compiler keeps them in SGPRs via s_load; staged to LDS for per-lane lookups) -------
struct DRect {           // rect (reference primitive.h:120-183) in its XZ-canonical frame
    float x0, z0, x1, z1, y;
    int32_t plane;       // plane_enum XY/XZ/YZ
    float ny;            // 2*normal-1 = +1 / -1 (primitive.h:212)
    int32_t mat;
};
struct DPrim {           // 272 bytes
    int32_t type, mat;
    int32_t boundary;    // volume: prim index
    int32_t phase_mat;   // volume
    float density;       // volume
    float cx, cy, cz, radius;  // sphere
    int32_t pad[3];
    int32_t hit_mat[8];  // rec.mat_ptr by face: rect/sphere [0] = mat, box [0..5] = mat, volume [0] = phase_mat
    DRect r[6];          // rect: r[0];  box: the six sides in primitive.h:232-240 order
};
struct DInst {           // 112 bytes
    float inv[12];       // world -> local (transform.inverse())
    float fwd[12];       // local -> world
    int32_t prim;
    int32_t vol_ordinal; // ordinal among volume instances (stream RNG dimension slot), -1 otherwise
    int32_t ident;       // 1 if the linear parts of inv and fwd are exactly the identity (pure translation)
    int32_t pad[1];
};
struct DMat {            // 48 bytes
    int32_t type;
    float r, g, b;
    float alpha, power;
    int32_t two_sided;
    int32_t tex;         // albedo / emit texture (DScene::tex index) when it is not a constant one, else -1
    float att[3];        // lambertian / metal with a constant albedo: albedo / float(M_PI), the attenuation scatter() returns
                         // (material.h:46, 94), divided once on the host with the same IEEE float division
    int32_t pad;
};
// Per hit id (instance * 8 + face): everything of the hit record that does not depend on where the face was hit, computed
// once on the host with the float operations the kernels would perform per hit (pt_context.cpp build_faces; both sides
// are compiled without FMA contraction, IEEE division and square root): the face's material, its local normal, and for
// both outcomes of rect::hit's facing test (primitive.h:214-222) the world normal  transform.apply_normal(n)
// (transform3.h:60-63), that normal normalised once more (every consumer -- cos_i integrator.h:201, cosine_pdf pdf.h:18-29,
// cos_l integrator.h:236 -- calls unit_vector on it) and the onb built from it (helpers.h:127-136).
// Eleven float4 per id:  [0] = bits(mat | ptype << 24 | shading class << 28), local normal xyz;  [1..3] normal kept:  n.xyz nu.x | nu.yz u.xy |
// u.z v.xyz;  [4..6] normal flipped, same layout;  [7] unused.  [8..10] the face's DMat (the 48 bytes of
// mats[mat]), so that everything a hit id leads to is one round trip.  A sphere's normal depends on the hit point: ptype says so.
#define PT_FACE_F4 11
// texture.h / image.h.  Read with per-lane indices (which child a checker picks depends on the hit point).
struct DTex {            // 48 bytes
    int32_t type;        // PT_TEX_*
    int32_t even, odd;   // checker children
    float scale;         // checker, perlin
    float r, g, b, a;    // constant
    int32_t width, height, texel0;   // image: texels[texel0 + y*width + x]
    int32_t uses_uv;     // the tree below reaches an image texture (hit_record u, v are needed)
};
// Traversal program: the pointer BVH (reference bvh.h:31-69) flattened into a linear op list that every
// lane of a wave sweeps in lock step.  eval(node) = ENTER box-test (miss: result MISS, jump to `a`) ;
// eval(left) ; [push result: folded into the next op as push_slot] ; eval(right) ; COMBINE slot.
// COMBINE keeps the reference's rule "both hit: left iff left.t < right.t" (bvh.h:40-47), so exact-t ties and
// NaN t resolve identically.  Every op carries the data it needs INLINE (node box, or the instance's inverse
// affine + its primitive's parameters): one op = one 128-byte scalar load with an address that depends only on
// the program counter, so loads are never chained (op -> instance -> primitive).
enum {
    OP_ENTER = 0,        // f[0..5] = node bbox; a = pc to continue at when the box is missed
    OP_COMBINE = 1,      // slot
    OP_LEAF_RECT_XY = 2, // f[0..11] = instance inverse affine, f[12..16] = x0 z0 x1 z1 y   (primitive.h:120-124)
    OP_LEAF_RECT_XZ = 3,
    OP_LEAF_RECT_YZ = 4,
    OP_LEAF_BOX = 5,     // f[0..11] inverse, f[12..14] = p0, f[15..17] = p1              (primitive.h:229-242)
    OP_LEAF_SPHERE = 6,  // f[0..11] inverse, f[12..14] = center, f[15] = radius
    OP_LEAF_VOLBOX = 7,  // constant_medium with a box boundary: as OP_LEAF_BOX + f[18] = density, f[19] = bits(vol_ord)
    OP_LEAF_VOLSPHERE = 8, // constant_medium with a sphere boundary: as OP_LEAF_SPHERE + f[18] = density, f[19] = bits(vol_ord)
    OP_LEAF_NONE = 9,    // a leaf that never reports a hit: constant_medium whose boundary is a rect (its second boundary hit
                         // beyond t1 + 0.0001 cannot exist, or both are NaN and the medium test fails: volume.h:33-46, 70-75)
    OP_LEAF_VOLVOL = 10, // constant_medium whose boundary is a constant_medium (volume.h:10 takes any hittable): f[12..17] = the INNER
                         // medium's boundary (box p0 p1, or sphere centre + radius), f[18] = outer density, f[19] = bits(first draw slot),
                         // f[20] = inner density, f[21] = bits(inner boundary: 1 box, 2 sphere).  General sweep only (the scene is not tame)
};
struct DOp {             // 128 bytes = two 64-byte halves, 128-byte aligned in the device array
    // first half: everything an op needs FIRST (header + node box or instance matrix)
    int32_t kind;
    int32_t a;           // ENTER: skip target;  LEAF: instance index (hit id = a*8 + face)
    int32_t slot;        // COMBINE: short-stack slot;  LEAF: bits 0-3 shape of the inverse's linear part -- 0 general, 1 exactly
                         // the identity, 2 / 3 / 4 the x / y / z axis is mapped to itself (zero row and column off the
                         // diagonal); bit 4: the leaf's data are outside the precondition of the unscaled division
                         // bit 5: a rect that walls the scene in (pt_context.cpp plane_walls_scene: shadow sweeps prove it unreachable),
                         // bits 8-13: the same for the six sides of a box, in box::hit's order;
                         // bit 6: a box / medium-on-a-box leaf with its enlarged world box in g[10..15] (wave-level cull)
    int32_t push_slot;   // >= 0: store the current partial result into this slot BEFORE executing the op
    float f[12];
    // second half: primitive parameters, needed only after the ray has been transformed
    float g[16];         // g[i] = "f[12 + i]" of the table above
};
#define PT_FLAT_MAX_INSTANCES 24
#define PT_WALK_MIN_INSTANCES 2048   // measured (tools/walk_bench.py): the lock-step sweep beats the walk up to at least 300 instances
#define PT_STAGE_MAX_SAMPLES 4   // 24 bytes x 256 lanes per sample: 24 KB of LDS per workgroup at 4, six workgroups per CU
#define PT_MAX_STACK 8   // short-stack slots per lane and ray held in LDS (tree height <= 8)

struct DCamera {
    float origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3];
    float lens_radius;
};

struct DScene {
    const DInst *insts;
    const DPrim *prims;
    const DMat *mats;
    const DOp *ops;
    const int32_t *lights;
    const float4 *faces;         // [n_insts*8][PT_FACE_F4] by hit id, see above
    const float4 *emit;          // [n_insts*8] by hit id: emitted radiance of that face's material (xyz); w = 0: that is
                                 // all; w bit 0: one-sided diffuse_light (facing test of material.h:214-216), bit 1:
                                 // textured emit (value/alpha at the hit point) -- both need the hit record
    const DTex *tex;             // texture table, texels as float4 (byte / 255.0 like image.h:58-62)
    const float4 *texels;
    const float4 *ranvec;        // perlin::ranvec[256] (xyz), perlin perm_x / perm_y / perm_z [3][256]
    const int32_t *perm;
    int32_t bg_tex;              // World::background texture, -1 = the constant bg[] below
    int32_t geom_all;            // the program has sphere or constant_medium leaves: launch the <GA = true> traversal
    int32_t textured;            // some material or the background uses the texture table: launch the <TEX = true> kernels
    int32_t n_insts, n_prims, n_mats, n_ops, n_lights, n_vol;
    int32_t stack_depth;         // short-stack slots the program uses
    int32_t ops_fast_off, n_ops_fast;   // the fast program (no COMBINE ops) sits at ops + ops_fast_off
    // Small scenes (PT_FLAT_MAX_INSTANCES): the fast program keeps only the ROOT's ENTER op -- every leaf is tested, which
    // is what the node boxes of a handful of room-sized instances amount to anyway -- and the winner's PARENT box (it
    // implies every ancestor's, pt_context.cpp) is tested afterwards, per lane, from this table: chains[instance][2] =
    // (min xyz, max xyz), an all-space box for children of the root.  n_chain = 0: the fast program carries all its ENTER
    // ops (pt_kernels.hip world_hit_fast).
    const float4 *chains;
    int32_t n_chain, pad_chain;
    // Scenes of very many instances (> PT_WALK_MIN_INSTANCES): the per-lane WALK (pt_kernels.hip world_hit_walk) instead of
    // the sweeps.  wnodes[2 n], wnodes[2 n + 1] = bvh node n: (box min xyz, bits(left)), (box max xyz, bits(right)); a child
    // >= 0 is a node, < 0 is ~(index into ops of the leaf's DOp in the fast program).
    const float4 *wnodes;
    int32_t walk, pad_walk;
    int32_t tame;                // the fast sweep / walk may run (ordered finite node boxes; not switched off): leaves whose
                                 // data allow it then use the unscaled exact division of pt_fdiv.h (DOp::slot bit 4 says no)
    DCamera cam;
    float bg[3];
    // config
    int32_t width, height;
    int32_t max_bounces, light_samples, russian_roulette, only_direct;
    float normal_offset;
    uint32_t seed_k0, seed_k1;   // mix(seed), mix(seed + 0x632BE5AB): folded into the per-path keys
};

// ---- wavefront streams in HBM -----------------------------------------------------------------------
// A batch owns P path slots cut into segments.  Slot s of segment g lives at g*seg_cap + s.  Queues are
// *segmented*: a workgroup takes one 256-lane chunk of one segment and appends its survivors to the output
// segment with ONE atomicAdd on that segment's counter (a few workgroups per counter: no hot word).
struct DQueue {          // path queue: 64 B per path
    float4 *r0;          // origin.xyz, bits(slot)
    float4 *r1;          // direction.xyz, last_bsdf_pdf
    float4 *s0;          // beta.xyz, bits(k0)
    float4 *s1;          // attenuation.xyz, bits(k1)
    int32_t *count;      // [n_seg] live entries per segment
};
struct DShadowQueue {    // per surviving hit: 16 B + light_samples * 24 B
    float4 *p0;          // hit point p.xyz, bits(slot | pending flag)
    uint2 *key;          // k0, k1 (only read when the scene has volumes)
    float4 *d;           // [k][P]: light direction.xyz, coef.x
    float2 *e;           // [k][P]: coef.y, coef.z
    int32_t *count;      // [n_seg]
};
struct DCounters {       // one bank; padded to 128 B so that banks never share a cache line
    unsigned long long camera_samples;
    unsigned long long rays, ext_rays, ext_hits, shadow_rays;
    unsigned long long term_miss, term_rr, term_emitter, term_pdf, term_bounce_limit;
    unsigned long long shadow_untraced;   // shadow rays counted in `rays` / `shadow_rays` like the reference counts them but never
                                          // traced: the light samples of hits that got no shadow record (k_shade, staged samples)
    unsigned long long pad[5];
};
#define PT_N_COUNTERS 11                  // leading words of DCounters that are counters
// Every workgroup adds its counts with a handful of atomics; on ONE set of words the ~4000 workgroups of a launch
// serialise at the memory-side atomic rate (about 90 adds per microsecond per word), which was a 50-100 us floor per
// launch.  The counters are therefore banked by workgroup index and summed on the host.
#define PT_COUNTER_BANKS 256
struct DTile { int32_t x0, y0, w, pix0; };   // pix0 = first batch-local pixel index of this tile
struct DBatch {
    // the batch covers n_tiles pixel rects (image tiles, reference queue.h:121-127); batch-local pixel pl lives
    // in the tile t with tiles[t].pix0 <= pl < tiles[t+1].pix0, row-major inside the tile.  n_tiles == 1 uses
    // (x0, y0, w) directly.
    int32_t x0, y0, w;
    int32_t n_tiles;
    const DTile *tiles;          // device array [n_tiles + 1] (last entry: pix0 = npix), only read when n_tiles > 1
    int32_t npix;                // pixels in the batch
    int32_t s0, ns;              // first sample index, samples in the batch
    // Segmentation of the P slots.  Bounce b reads queues cut into n_seg segments of seg_cap slots and writes its
    // survivors into n_seg_out segments of seg_cap_out = 2*seg_cap slots (segment g -> g >> 1; the two source
    // regions are adjacent, so the merged segment is exactly their union and can never overflow).  The number of
    // live paths roughly halves per bounce, so segments -- and with them the 256-lane chunks the workgroups take --
    // stay full instead of thinning out.
    int32_t n_seg, seg_cap;
    int32_t n_seg_out, seg_cap_out;
    int32_t perm, perm_out;      // multipliers of the segment visiting order for n_seg and n_seg_out (pt_kernels.hip ChunkWalk; 1 = queue order)
    int32_t sort_shade;          // k_shade orders each chunk by shading class before shading it (PATHTRACE_HIP_NO_SORT=1: off)
    int32_t stage_shadow;        // k_shade stages a hit's light samples in LDS and gives hits whose samples cannot contribute no
                                 // shadow record (light_samples <= PT_STAGE_MAX_SAMPLES; PATHTRACE_HIP_NO_STAGE=1: off)
    int64_t P;                   // allocated slots (plane stride of the shadow queue)
    int64_t n_paths;             // npix*ns  (<= P)
};
struct DStreams {
    DQueue q[2];
    DShadowQueue sq;
    float2 *hit;                 // [P]: t, bits(id)   id = -1 miss, else instance*8 + face
    float4 *radiance;            // [P]: per camera sample radiance sum (rgb, unused)
    float4 *pending;             // [P]: second emitter addition (integrator.h:319), applied by connect
    float4 *fb;                  // [height*width] rgba framebuffer SUM
    int32_t *qmax;               // [64 bounces][2 queues][64 banks] per batch, zeroed when it starts: the largest live count of any output
                                 // segment of bounce b's path queue / shadow queue, max over the banks (pt_kernels.hip chunk_limit)
    DCounters *counters;         // [PT_COUNTER_BANKS]
    float2 *gstack;              // general sweep's short stack for trees deeper than PT_MAX_STACK: [slot][ray <= 4][gstack_stride
    int32_t gstack_stride;       // threads], NULL while the LDS stack suffices (pt_kernels.hip stack_of)
    int32_t pad_;
};

}  // namespace ptd
