// The reference-side binding of libpathtrace_hip.so: `HipWavefront : Renderer` (renderer.h:114-150) -- the class a
// maintainer of the reference adds next to Naive / Progressive / Tiled.  It is written against the
// REFERENCE's own types (World, bvh_node, instance, rect / box / sphere / constant_medium, the materials and textures,
// camera, Config, Renderer) and compiles only next to them: include it after the reference's renderer.h and volume.h /
// image.h (tools/integration/plugin_driver.cpp does; oracle/Makefile `plugin` builds that into oracle/_ref/, build
// container only).  Nothing here is used by the product library or by the GPU tests.
//
// Two parts:
//   HipFlatScene  walks the reference's object graph into the flat POD scene of include/pathtrace_hip.h;
//   HipWavefront  forwards start_render / sync_progress / is_done / finalize to the C ABI.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "pathtrace_hip.h"

// What build_scene (scene_parser.h:241-595) has at hand when it returns and World does not keep: the objects in the
// order it constructed them.  `instances` is the list handed to bvh_node BEFORE its constructor sorted it (file order =
// hit_record::primitive ids); a constant_medium's own isotropic phase function (volume.h:14-17) is listed among the
// materials where the parser met the volume.
struct HipSceneLists {
    std::vector<texture *> textures;
    std::vector<material *> materials;
    std::vector<hittable *> primitives;
    std::vector<hittable *> instances;
};

struct HipFlatScene {
    std::vector<pt_texture> textures;
    std::vector<uint8_t> texels;
    std::vector<pt_material> materials;
    std::vector<pt_primitive> primitives;
    std::vector<pt_instance> instances;
    std::vector<pt_bvh_node> nodes;
    std::vector<int32_t> lights;
    std::vector<float> perlin_ranvec;
    std::vector<int32_t> perlin_perm;
    pt_scene_desc desc{};

    template <class T, class U>
    static int index_of(const std::vector<T *> &v, const U *p)
    {
        for (size_t i = 0; i < v.size(); i++)
            if ((const void *)v[i] == (const void *)p) return (int)i;
        return -1;
    }
    static void put(float *d, const vec3 &v) { d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; }

    // a texture* as a material / the background refers to it: a constant_texture is folded into (colour, alpha), anything
    // else is an index into the texture table
    struct TexRef { vec3 color; float alpha; int index; };
    TexRef tex_ref(const HipSceneLists &L, texture *t) const
    {
        if (auto *c = dynamic_cast<constant_texture *>(t)) return TexRef{c->color, c->a, -1};
        const int i = index_of(L.textures, t);
        if (i < 0) throw std::runtime_error("HipFlatScene: a texture that build_scene did not list");
        return TexRef{vec3(0, 0, 0), 1.0f, i};
    }

    void flatten_textures(const HipSceneLists &L)
    {
        for (texture *t : L.textures) {
            pt_texture o{};
            o.alpha = 1.0f; o.even = -1; o.odd = -1; o.scale = 1.0f;
            if (auto *c = dynamic_cast<constant_texture *>(t)) {
                o.type = PT_TEX_CONSTANT; put(o.color, c->color); o.alpha = c->a;
            } else if (auto *k = dynamic_cast<checker_texture *>(t)) {
                o.type = PT_TEX_CHECKER; o.even = index_of(L.textures, k->even); o.odd = index_of(L.textures, k->odd); o.scale = k->scale;
                if (o.even < 0 || o.odd < 0) throw std::runtime_error("HipFlatScene: checker child not listed");
            } else if (auto *n = dynamic_cast<noise_texture *>(t)) {
                o.type = PT_TEX_PERLIN; o.scale = n->scale;
            } else if (auto *im = dynamic_cast<image_texture *>(t)) {
                // image.h:52-69 stored byte / 255.0 per channel; the bytes lodepng decoded are round(value * 255)
                o.type = PT_TEX_IMAGE; o.width = im->width; o.height = im->height; o.texel_offset = (int64_t)texels.size();
                for (int y = 0; y < im->height; y++)
                    for (int x = 0; x < im->width; x++) {
                        for (int ch = 0; ch < 3; ch++) texels.push_back((uint8_t)(int)(im->data[y][x][ch] * 255.0 + 0.5));
                        texels.push_back((uint8_t)(int)(im->alpha_mask[y][x] * 255.0 + 0.5));
                    }
            } else throw std::runtime_error("HipFlatScene: texture type without a HIP counterpart");
            textures.push_back(o);
        }
        // the process-wide Perlin tables (texture.h:176-183)
        perlin_ranvec.resize(768);
        perlin_perm.resize(768);
        for (int i = 0; i < 256; i++) {
            for (int c = 0; c < 3; c++) perlin_ranvec[3 * i + c] = perlin::ranvec[i][c];
            perlin_perm[i] = perlin::perm_x[i]; perlin_perm[256 + i] = perlin::perm_y[i]; perlin_perm[512 + i] = perlin::perm_z[i];
        }
    }

    void flatten_materials(const HipSceneLists &L)
    {
        for (material *m : L.materials) {
            pt_material o{};
            o.alpha = 1.0f; o.power = 1.0f; o.two_sided = 1; o.fuzz = 0.0f; o.ior = 1.45f; o.texture = -1;
            if (auto *l = dynamic_cast<lambertian *>(m)) {
                const TexRef r = tex_ref(L, l->albedo);
                o.type = PT_MAT_LAMBERTIAN; put(o.color, r.color); o.alpha = r.alpha; o.texture = r.index;
            } else if (auto *me = dynamic_cast<metal *>(m)) {
                o.type = PT_MAT_METAL; put(o.color, me->albedo); o.fuzz = me->fuzz;
            } else if (auto *d = dynamic_cast<dielectric *>(m)) {
                o.type = PT_MAT_DIELECTRIC; put(o.color, vec3(1, 1, 1)); o.ior = d->ref_idx;
            } else if (auto *e = dynamic_cast<diffuse_light *>(m)) {
                const TexRef r = tex_ref(L, e->emit);
                o.type = PT_MAT_DIFFUSE_LIGHT; put(o.color, r.color); o.alpha = r.alpha; o.texture = r.index;
                o.power = e->power; o.two_sided = e->two_sided;
            } else if (auto *i = dynamic_cast<isotropic *>(m)) {
                const TexRef r = tex_ref(L, i->albedo);
                o.type = PT_MAT_ISOTROPIC; put(o.color, r.color); o.alpha = r.alpha; o.texture = r.index;
            } else throw std::runtime_error("HipFlatScene: material type without a HIP counterpart");
            materials.push_back(o);
        }
    }

    void flatten_primitives(const HipSceneLists &L)
    {
        for (hittable *h : L.primitives) {
            pt_primitive o{};
            o.boundary = -1; o.phase_material = -1;
            if (auto *r = dynamic_cast<rect *>(h)) {                       // primitive.h:177-183
                o.type = PT_PRIM_RECT; o.material = index_of(L.materials, r->mp);
                o.rect[0] = r->x0; o.rect[1] = r->z0; o.rect[2] = r->x1; o.rect[3] = r->z1; o.rect[4] = r->y;
                o.plane = (int)r->type; o.flipped = !r->normal;
            } else if (auto *b = dynamic_cast<box *>(h)) {                  // primitive.h:254-255; its six rects share one material
                o.type = PT_PRIM_BOX; put(o.p0, b->p0); put(o.p1, b->p1);
                o.material = index_of(L.materials, ((rect *)((hittable_list *)b->group)->list[0])->mp);
            } else if (auto *s = dynamic_cast<sphere *>(h)) {               // primitive.h:59-61
                o.type = PT_PRIM_SPHERE; put(o.center, s->center); o.radius = s->radius; o.material = index_of(L.materials, s->mat_ptr);
            } else if (auto *cm = dynamic_cast<constant_medium *>(h)) {     // volume.h:24-26
                o.type = PT_PRIM_VOLUME; o.boundary = index_of(L.primitives, cm->boundary); o.density = cm->density;
                o.phase_material = index_of(L.materials, cm->phase_function);
                o.material = primitives[o.boundary].material;              // scene_parser.h:231: the boundary's material
            } else throw std::runtime_error("HipFlatScene: primitive type without a HIP counterpart");
            if (o.material < 0) throw std::runtime_error("HipFlatScene: a material that build_scene did not list");
            primitives.push_back(o);
        }
    }

    int flatten_bvh(const HipSceneLists &L, hittable *h)
    {   // preorder; a leaf is an instance: ~index in file order
        const int li = index_of(L.instances, h);
        if (li >= 0) return ~li;
        bvh_node *n = (bvh_node *)h;
        const int me = (int)nodes.size();
        nodes.push_back(pt_bvh_node{});
        for (int c = 0; c < 3; c++) { nodes[me].bbox[c] = n->box._min[c]; nodes[me].bbox[3 + c] = n->box._max[c]; }
        const int l = flatten_bvh(L, n->left);
        const int r = flatten_bvh(L, n->right);
        nodes[me].left = l; nodes[me].right = r;
        return me;
    }

    void build(World *world, const camera &cam, const HipSceneLists &L)
    {
        flatten_textures(L);
        flatten_materials(L);
        flatten_primitives(L);
        for (hittable *h : L.instances) {
            instance *in = (instance *)h;
            pt_instance o{};
            o.primitive = index_of(L.primitives, in->ptr);
            if (o.primitive < 0) throw std::runtime_error("HipFlatScene: a primitive that build_scene did not list");
            const Eigen::Matrix4f f = in->transform._transform.matrix(), r = in->transform.inverse()._transform.matrix();
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 4; j++) { o.fwd[4 * i + j] = f(i, j); o.inv[4 * i + j] = r(i, j); }
            for (int c = 0; c < 3; c++) { o.bbox[c] = in->bbox._min[c]; o.bbox[3 + c] = in->bbox._max[c]; }
            instances.push_back(o);
        }
        flatten_bvh(L, world->ptr);
        for (hittable *l : world->lights) lights.push_back(index_of(L.instances, l));   // world.h:39
        desc = pt_scene_desc{};
        desc.n_materials = (int)materials.size(); desc.materials = materials.data();
        desc.n_primitives = (int)primitives.size(); desc.primitives = primitives.data();
        desc.n_instances = (int)instances.size(); desc.instances = instances.data();
        desc.n_nodes = (int)nodes.size(); desc.nodes = nodes.data();
        desc.n_lights = (int)lights.size(); desc.lights = lights.data();
        put(desc.camera.origin, cam.origin); put(desc.camera.lower_left_corner, cam.lower_left_corner);
        put(desc.camera.horizontal, cam.horizontal); put(desc.camera.vertical, cam.vertical);
        put(desc.camera.u, cam.u); put(desc.camera.v, cam.v); put(desc.camera.w, cam.w);
        desc.camera.lens_radius = cam.lens_radius;
        const TexRef bg = tex_ref(L, world->background);                     // world.h:27-30
        put(desc.background, bg.color);
        desc.background_texture = bg.index;
        desc.n_textures = (int)textures.size(); desc.textures = textures.data();
        desc.texel_bytes = (int64_t)texels.size(); desc.texels = texels.data();
        desc.perlin_ranvec = perlin_ranvec.data(); desc.perlin_perm = perlin_perm.data();
    }
};

// Which GPUs render: the reference's own worker count, config.threads (config.h:117; Tiled::start_render spawns that many
// worker threads over one spiral tile queue, renderer.h:553-603).  threads > 1 = that many devices, capped by the devices
// present, through pt_multi_* (one context per device, tiles of block_width x block_height in NaiveSpiral order owned by
// cost-balanced devices, every device driven by a host thread of its own, one exchange of the owned tiles at the end);
// threads <= 1 or a one-GPU machine = one context.  PATHTRACE_HIP_DEVICES="0,1,.." overrides the list ("all" = every visible
// device; an ordinal may repeat: "0,0" rehearses the multi-device path on a one-GPU box).
class HipWavefront : public Renderer
{
public:
    HipWavefront(Integrator *integrator, camera cam, Config config, World *world, const HipSceneLists &lists)
        : Renderer{integrator, cam, config}
    {
        flat.build(world, cam, lists);
        pt_config pc{};
        pc.width = film.width; pc.height = film.height;
        pc.max_bounces = config.max_bounces; pc.light_samples = config.light_samples;
        pc.russian_roulette = config.russian_roulette; pc.only_direct_illumination = config.only_direct_illumination;
        pc.normal_offset = config.normal_offset; pc.seed = 0; pc.device = -1;
        pc.max_paths_in_flight = 0;                                           // the library's launch plan sizes the streams (ABI v6)
        std::vector<int32_t> devices;
        if (const char *e = getenv("PATHTRACE_HIP_DEVICES")) {
            if (!strcmp(e, "all")) for (int d = 0; d < pt_device_count(); d++) devices.push_back(d);
            else for (const char *p = e; *p;) {
                char *end = nullptr;
                const long v = strtol(p, &end, 10);
                if (end == p) break;
                devices.push_back((int32_t)v);
                p = (*end == ',') ? end + 1 : end;
            }
        } else if (config.threads > 1) {
            const int n = std::min<int>((int)config.threads, pt_device_count());
            for (int d = 0; d < n && n > 1; d++) devices.push_back(d);
        }
        // set-up, like Renderer::Renderer's framebuffer allocation (renderer.h:121-133): the wavefront streams for this job
        // (pt_reserve: the launch plan from film x samples and the free HBM) and the per-scene build of the traversal
        // kernels, so that start_render -> finalize times rendering alone
        if (devices.size() > 1) {
            multi = pt_multi_create(&flat.desc, &pc, (int32_t)devices.size(), devices.data(), std::max(config.block_width, 1), std::max(config.block_height, 1));
            ASSERT(multi != nullptr, pt_last_error());
            ASSERT(pt_multi_reserve(multi, std::max(config.samples, 1), 50) == 0, pt_last_error());   // + per-scene build + 50 ms warm-up per device
        } else {
            if (devices.size() == 1) pc.device = devices[0];
            ctx = pt_create(&flat.desc, &pc);
            ASSERT(ctx != nullptr, pt_last_error());                          // the reference's error style (types.h:5-14)
            ASSERT(pt_reserve(ctx, (int64_t)film.width * film.height, std::max(config.samples, 1)) == 0, pt_last_error());
            (void)pt_spec_wait(ctx);                                          // -1: the generic kernels render
            ASSERT(pt_prime(ctx, 0, nullptr, 50) == 0, pt_last_error());      // a fresh process's first render is ~5 % slower otherwise
        }
        staging.resize((size_t)film.width * film.height * 3);
        completed = false;
    }
    void preprocess() {}
    void start_render(std::chrono::high_resolution_clock::time_point)
    {
        render_start_time = std::chrono::high_resolution_clock::now();
        // whole film, all samples: enqueued asynchronously, returns at once (replaces the thread spawn renderer.h:595)
        const int rc = multi ? pt_multi_render_async(multi, 0, config.samples) : pt_render_async(ctx, 0, 0, film.width, film.height, 0, config.samples);
        ASSERT(rc == 0, pt_last_error());
    }
    void next_pixel_and_ray(int, ray &, int, int) {}
    void to_framebuffer()
    {
        for (int j = 0; j < film.height; j++)
            for (int i = 0; i < film.width; i++) {
                const float *p = &staging[3 * ((size_t)j * film.width + i)];
                framebuffer[j][i] = vec3(p[0], p[1], p[2]);                    // row 0 = bottom row = framebuffer[0]
            }
    }
    void sync_progress()
    {
        uint64_t samples = 0, rays = 0;
        const int done = multi ? pt_multi_poll(multi, &samples, &rays) : pt_poll(ctx, &samples, &rays);   // replaces summing samples_done[] (renderer.h:607-612)
        ASSERT(done >= 0, pt_last_error());
        print_out_progress((long)samples, (long)config.samples * film.total_pixels - (long)samples, render_start_time);
        if (!done) {                                                          // preview from the LIVE framebuffer (renderer.h:614-618)
            uint64_t acc = 0;
            const int rc = multi ? pt_multi_snapshot_framebuffer(multi, staging.data(), &acc) : pt_snapshot_framebuffer(ctx, staging.data(), &acc);
            ASSERT(rc == 0, pt_last_error());
            to_framebuffer();
            float avg, mx, tot;
            const int div = 1 + (int)(acc / ((uint64_t)film.width * film.height));
            calculate_luminance(framebuffer, film.width, film.height, div, film.width * film.height, mx, tot, avg);
            output_to_file(output, framebuffer, film.width, film.height, div, mx, film.exposure, film.gamma);
        }
        completed = done == 1;
    }
    bool is_done() { return completed; }
    void compute(int) {}                                                      // no CPU worker threads
    void finalize()
    {
        ASSERT((multi ? pt_multi_wait(multi) : pt_wait(ctx)) == 0, pt_last_error());
        // renderer.h:696-706.  The reference's elapsed time is finalize's clock minus start_render's, which includes up to one
        // 0.5 s sleep of main.cpp:158-163's loop; the library stamps the moment the device finished its last batch
        // (pt_render_seconds) and that is what the rates below use; the process's own wall figure is printed beside it.
        const std::chrono::duration<double> wall = std::chrono::high_resolution_clock::now() - render_start_time;
        double dt = multi ? pt_multi_render_seconds(multi) : pt_render_seconds(ctx);
        if (!(dt > 0)) dt = wall.count();
        ASSERT((multi ? pt_multi_read_framebuffer(multi, staging.data()) : pt_read_framebuffer(ctx, staging.data())) == 0, pt_last_error());
        to_framebuffer();
        pt_counters c;
        ASSERT((multi ? pt_multi_get_counters(multi, &c) : pt_get_counters(ctx, &c)) == 0, pt_last_error());   // c.rays = total_bounces of renderer.h:696-706
        rays_traced = c.rays;
        const int workers = multi ? pt_multi_device_count(multi) : 1;
        std::cout << "time taken to compute " << dt << std::endl;
        std::cout << "(start_render to finalize on this process's clock, polling included: " << wall.count() << ")" << std::endl;
        const float rate1 = film.total_pixels * config.samples / dt, rate2 = c.rays / dt;
        std::cout << "computed " << film.total_pixels * config.samples << " camera rays in " << dt << "s, at " << rate1 << " rays per second, or " << rate1 / workers << "per device" << std::endl;
        std::cout << "computed " << c.rays << " rays, at " << rate2 << " rays per second, or " << rate2 / workers << " per device" << std::endl;
        std::cout << "traced " << c.rays_traced << " rays, at " << (float)(c.rays_traced / dt) << " rays per second" << std::endl;
        if (ctx) {
            pt_plan pl;
            if (pt_get_plan(ctx, &pl) == 0)
                std::cout << "launch plan: " << pl.batches << " batches of " << pl.spp_per_batch << " spp on " << pl.lanes << " lanes, " << pl.path_slots
                          << " path slots, " << pl.stream_bytes / 1e9 << " GB of streams" << std::endl;
            // ABI v5: who compiled the traversal kernels this render ran -- a foreign compiler (a host process that carries
            // another ROCm under the same sonames) builds correct, measurably slower kernels: say so next to the statistics
            char info[1024];
            const int n = pt_spec_info(ctx, info, sizeof info);
            if (n > 0 && n < (int)sizeof info) std::cout << "per-scene kernels: " << info << std::endl;
        }
        float max_luminance, avg_luminance, total_luminance;                  // unchanged film output, renderer.h:719-727
        calculate_luminance(framebuffer, film.width, film.height, config.samples, film.width * film.height, max_luminance,
                            total_luminance, avg_luminance);
        std::cout << "avg lum " << avg_luminance << std::endl;
        std::cout << "max lum " << max_luminance << std::endl;
        output_to_file(output, framebuffer, film.width, film.height, config.samples, max_luminance, film.exposure, film.gamma);
        pt_destroy(ctx);
        pt_multi_destroy(multi);
        ctx = nullptr;
        multi = nullptr;
    }
    pt_ctx *ctx = nullptr;
    pt_multi *multi = nullptr;
    HipFlatScene flat;
    std::vector<float> staging;
    uint64_t rays_traced = 0;
};
