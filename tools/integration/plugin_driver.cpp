// TEST INFRASTRUCTURE (build container only, output oracle/_ref/plugin_driver, never shipped): compiles the
// reference-side plugin tools/integration/hip_wavefront.h against the REAL reference headers and links it with
// libpathtrace_hip.so -- the build a maintainer of the reference would do -- and dumps the flat scene
// the plugin derives from the reference's own World so that tests/test_integration_plugin.py can compare it with
// pth_scene_from_file's.
//
//   plugin_driver params.txt flatten W H out.txt      canonical text dump of the plugin's pt_scene_desc
//   plugin_driver params.txt render  <cfg...> out.f32 [threads] the whole Renderer protocol through HipWavefront (needs a GPU);
//                                                      threads = config.threads = GPUs that take part (default 1)
//
// The scene comes from oracle/ref_build.h (the reference's constructors driven by oracle/scene_params.py's text), because
// the reference's own scene_parser.h includes the un-vendored lodepng header and does not compile here.
#include "bvh.h"
#include "camera.h"
#include "helpers.h"
#include "hittable_list.h"
#include "material.h"
#include "image.h"
#include "pdf.h"
#include "primitive.h"
#include "random.h"
#include "texture.h"
#include "volume.h"
#include "world.h"
#include "types.h"
#include "integrator.h"
#include "renderer.h"
#include "config.h"

#include <thread>

#include "ref_build.h"
#include "hip_wavefront.h"

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static void dump_floats(FILE *f, const char *tag, const float *v, int n)
{
    fprintf(f, "%s", tag);
    for (int i = 0; i < n; i++) fprintf(f, " %08x", bits(v[i]));
    fprintf(f, "\n");
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: plugin_driver params.txt flatten|render ...\n"); return 2; }
    const std::string mode = argv[2];
    Built b = build(argv[1]);
    HipSceneLists lists;
    lists.textures = b.textures;
    lists.materials = b.materials;
    lists.primitives = b.prims;
    lists.instances = b.list;
    if (mode == "flatten") {
        const int W = atoi(argv[3]), H = atoi(argv[4]);
        camera cam(vec3(b.cam[0], b.cam[1], b.cam[2]), vec3(b.cam[3], b.cam[4], b.cam[5]), vec3(0, 1, 0), b.cam[6], float(W) / float(H),
                   b.cam[7], b.cam[8], 0.0, 1.0);                            // main.cpp:86-104
        HipFlatScene fs;
        fs.build(b.world, cam, lists);
        const pt_scene_desc &d = fs.desc;
        FILE *f = fopen(argv[5], "w");
        fprintf(f, "counts %d %d %d %d %d %d\n", d.n_materials, d.n_primitives, d.n_instances, d.n_nodes, d.n_lights, d.n_textures);
        for (int i = 0; i < d.n_textures; i++) {
            const pt_texture &t = d.textures[i];
            fprintf(f, "texture %d %08x %08x %08x %08x %d %d %08x %d %d %lld\n", t.type, bits(t.color[0]), bits(t.color[1]), bits(t.color[2]),
                    bits(t.alpha), t.even, t.odd, bits(t.scale), t.width, t.height, (long long)t.texel_offset);
        }
        fprintf(f, "texels %lld", (long long)d.texel_bytes);
        unsigned long long h = 1469598103934665603ull;
        for (long long i = 0; i < d.texel_bytes; i++) { h ^= d.texels[i]; h *= 1099511628211ull; }
        fprintf(f, " %016llx\n", h);
        for (int i = 0; i < d.n_materials; i++) {
            const pt_material &m = d.materials[i];
            fprintf(f, "material %d %08x %08x %08x %08x %08x %d %08x %08x %d\n", m.type, bits(m.color[0]), bits(m.color[1]), bits(m.color[2]),
                    bits(m.alpha), bits(m.power), m.two_sided, bits(m.fuzz), bits(m.ior), m.texture);
        }
        for (int i = 0; i < d.n_primitives; i++) {
            const pt_primitive &p = d.primitives[i];
            fprintf(f, "prim %d %d", p.type, p.material);
            for (int k = 0; k < 5; k++) fprintf(f, " %08x", bits(p.rect[k]));
            fprintf(f, " %d %d", p.plane, p.flipped);
            for (int k = 0; k < 3; k++) fprintf(f, " %08x", bits(p.p0[k]));
            for (int k = 0; k < 3; k++) fprintf(f, " %08x", bits(p.p1[k]));
            for (int k = 0; k < 3; k++) fprintf(f, " %08x", bits(p.center[k]));
            fprintf(f, " %08x %d %08x %d\n", bits(p.radius), p.boundary, bits(p.density), p.phase_material);
        }
        for (int i = 0; i < d.n_instances; i++) {
            const pt_instance &in = d.instances[i];
            fprintf(f, "instance %d", in.primitive);
            for (int k = 0; k < 12; k++) fprintf(f, " %08x", bits(in.fwd[k]));
            for (int k = 0; k < 12; k++) fprintf(f, " %08x", bits(in.inv[k]));
            for (int k = 0; k < 6; k++) fprintf(f, " %08x", bits(in.bbox[k]));
            fprintf(f, "\n");
        }
        for (int i = 0; i < d.n_nodes; i++) {
            const pt_bvh_node &n = d.nodes[i];
            fprintf(f, "node");
            for (int k = 0; k < 6; k++) fprintf(f, " %08x", bits(n.bbox[k]));
            fprintf(f, " %d %d\n", n.left, n.right);
        }
        fprintf(f, "lights");
        for (int i = 0; i < d.n_lights; i++) fprintf(f, " %d", d.lights[i]);
        fprintf(f, "\n");
        dump_floats(f, "camera", d.camera.origin, 22);
        dump_floats(f, "background", d.background, 3);
        fprintf(f, "background_texture %d\n", d.background_texture);
        if (d.n_textures) {
            dump_floats(f, "perlin_ranvec", d.perlin_ranvec, 768);
            fprintf(f, "perlin_perm");
            for (int i = 0; i < 768; i++) fprintf(f, " %d", d.perlin_perm[i]);
            fprintf(f, "\n");
        }
        fclose(f);
        return 0;
    }
    if (mode == "render") {   // <cfg...> = W H spp max_bounces light_samples rr normal_offset only_direct bw bh ; then out.f32
        char **a = argv + 3;
        Config c;
        c.film.width = atoi(a[0]); c.film.height = atoi(a[1]); c.film.total_pixels = (long)c.film.width * c.film.height;
        c.film.exposure = 2.2f; c.film.gamma = 0.0f;
        c.samples = atoi(a[2]); c.max_bounces = atoi(a[3]); c.light_samples = atoi(a[4]); c.russian_roulette = atoi(a[5]) != 0;
        c.normal_offset = (float)strtod(a[6], nullptr); c.only_direct_illumination = atoi(a[7]) != 0;
        c.block_width = atoi(a[8]); c.block_height = atoi(a[9]);
        c.ppm_output_path = "/dev/null"; c.png_output_path = ""; c.traced_paths_output_path = "/dev/null";
        c.traced_paths_2d_output_path = "/dev/null"; c.scene_path = ""; c.should_trace_paths = false; c.avg_number_of_paths = 100;
        c.trace_probability = 0.0; c.render_type = TILED; c.integrator_type = INEEPT; c.threads = argc > 14 ? (uint16_t)atoi(argv[14]) : 1;
        camera cam(vec3(b.cam[0], b.cam[1], b.cam[2]), vec3(b.cam[3], b.cam[4], b.cam[5]), vec3(0, 1, 0), b.cam[6],
                   float(c.film.width) / float(c.film.height), b.cam[7], b.cam[8], 0.0, 1.0);
        b.world->config = c;                                                   // main.cpp:139
        Integrator *integrator = new NEEIterative(c.max_bounces, b.world);     // main.cpp:30-55 (only its type selects the kernels)
        Renderer *r = new HipWavefront(integrator, cam, c, b.world, lists);    // main.cpp:57-84 + the new case
        r->start_render(std::chrono::high_resolution_clock::now());           // main.cpp:156-167
        while (!r->is_done()) {                                               // main.cpp:158-163 (it sleeps 0.5 s; the preview is the same work)
            r->sync_progress();
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
        r->finalize();
        FILE *f = fopen(a[10], "wb");
        for (int j = 0; j < c.film.height; j++)
            for (int i = 0; i < c.film.width; i++) fwrite(r->framebuffer[j][i].e, sizeof(float), 3, f);
        fclose(f);
        return 0;
    }
    fprintf(stderr, "unknown mode %s\n", mode.c_str());
    return 2;
}
