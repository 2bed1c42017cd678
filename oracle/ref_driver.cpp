// TEST INFRASTRUCTURE -- driver around the *real* reference headers (never shipped, never
// linked into the product).  Built by oracle/Makefile into oracle/_ref/ref_driver from the
// reference sources where they lie (-I/root/reference); no reference source is copied.
//
// Why a driver instead of the reference's main.cpp: main.cpp includes scene_parser.h, which
// includes the un-vendored thirdparty/lodepng/lodepng.h (empty submodule), so the full
// program is unbuildable here without a stand-in header.  The hot path itself (renderer.h,
// integrator.h, world.h, bvh.h, primitive.h, volume.h, material.h, pdf.h, camera.h,
// transform3.h + vendored Eigen) compiles as it lies.  This file therefore replaces only
// the JSON front end: it reads the "constructor parameter" text that
// oracle/scene_params.py derives from a scene JSON and calls the reference's own
// constructors in the order scene_parser.h:241-595 does.
//
// Modes (argv[2]):
//   rng N out.f64                     first N random_double() values after static init
//   tables out.txt                    matrices / bboxes / BVH topology / camera / lights
//   render  <cfg...> out.f32 [out.ppm] Tiled + NEEIterative through start_render/sync/finalize; the optional PPM is
//                                     the file Renderer::output writes (output_to_file renderer.h:24-55)
//   samples <cfg...> N out.f32        first N camera samples in Tiled::compute order:
//                                     u v  ray(7)  col(3)  rays(1)  = 13 floats / sample
//   hits    <cfg...> N out.f32        world->hit of the first N camera rays:
//                                     hit t p(3) n(3) inst  = 9 floats / sample
//   <cfg...> = W H spp max_bounces light_samples rr(0/1) normal_offset only_direct(0/1) bw bh
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

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "ref_build.h"

static Config make_config(char **a, const char *ppm_path)
{
    Config c;
    c.film.width = atoi(a[0]);
    c.film.height = atoi(a[1]);
    c.film.total_pixels = (long)c.film.width * c.film.height;
    c.film.exposure = 2.2f; // config.h:24-25 swap, irrelevant for the float dump
    c.film.gamma = 0.0f;
    c.samples = atoi(a[2]);
    c.max_bounces = atoi(a[3]);
    c.light_samples = atoi(a[4]);
    c.russian_roulette = atoi(a[5]) != 0;
    c.normal_offset = (float)strtod(a[6], nullptr);
    c.only_direct_illumination = atoi(a[7]) != 0;
    c.block_width = atoi(a[8]);
    c.block_height = atoi(a[9]);
    c.ppm_output_path = ppm_path;
    c.png_output_path = "";
    c.traced_paths_output_path = "/dev/null";
    c.traced_paths_2d_output_path = "/dev/null";
    c.scene_path = "";
    c.should_trace_paths = false;
    c.avg_number_of_paths = 100;
    c.trace_probability = 0.0;
    c.render_type = TILED;
    c.integrator_type = INEEPT;
    c.threads = 1;
    return c;
}

static camera make_camera(const Built &b, const Config &c)
{
    // main.cpp:86-104 + :148
    return camera(vec3(b.cam[0], b.cam[1], b.cam[2]), vec3(b.cam[3], b.cam[4], b.cam[5]), vec3(0, 1, 0), b.cam[6],
                  float(c.film.width) / float(c.film.height), b.cam[7], b.cam[8], 0.0, 1.0);
}

static int index_of(const std::vector<hittable *> &v, const hittable *p)
{
    for (size_t i = 0; i < v.size(); i++)
        if (v[i] == p)
            return (int)i;
    return -1;
}

static void dump_bvh(FILE *f, const Built &b, const hittable *h, int depth)
{
    int li = index_of(b.list, h);
    if (li >= 0)
    {
        fprintf(f, "%*sleaf %d\n", depth * 2, "", li);
        return;
    }
    const bvh_node *n = (const bvh_node *)h;
    fprintf(f, "%*snode %a %a %a %a %a %a\n", depth * 2, "", n->box._min[0], n->box._min[1], n->box._min[2],
            n->box._max[0], n->box._max[1], n->box._max[2]);
    dump_bvh(f, b, n->left, depth + 1);
    dump_bvh(f, b, n->right, depth + 1);
}

int main(int argc, char **argv)
{
    if (argc < 3)
    {
        fprintf(stderr, "usage: ref_driver params.txt mode ...\n");
        return 2;
    }
    std::string mode = argv[2];
    if (mode == "rng")
    {
        int n = atoi(argv[3]);
        std::vector<double> v(n);
        for (int i = 0; i < n; i++)
            v[i] = random_double();
        FILE *f = fopen(argv[4], "wb");
        fwrite(v.data(), sizeof(double), n, f);
        fclose(f);
        return 0;
    }
    if (mode == "perlin")
    {   // the static tables of texture.h:180-183 as the reference's own static initialisers left them
        FILE *f = fopen(argv[3], "wb");
        for (int i = 0; i < 256; i++)
        {
            float v[3] = {perlin::ranvec[i][0], perlin::ranvec[i][1], perlin::ranvec[i][2]};
            fwrite(v, sizeof(float), 3, f);
        }
        int *perms[3] = {perlin::perm_x, perlin::perm_y, perlin::perm_z};
        for (int k = 0; k < 3; k++)
            for (int i = 0; i < 256; i++)
            {
                float v = (float)perms[k][i];
                fwrite(&v, sizeof(float), 1, f);
            }
        fclose(f);
        return 0;
    }
    Built b = build(argv[1]);
    if (mode == "texeval")
    {   // texeval points.f32 n out.f32: every texture's value(u,v,p) and alpha(u,v,p) at n points (u v px py pz)
        int n = atoi(argv[4]);
        std::vector<float> pts((size_t)n * 5);
        FILE *fi = fopen(argv[3], "rb");
        if (!fi || fread(pts.data(), sizeof(float), pts.size(), fi) != pts.size())
        {
            fprintf(stderr, "cannot read %s\n", argv[3]);
            return 2;
        }
        fclose(fi);
        FILE *f = fopen(argv[5], "wb");
        for (texture *t : b.textures)
            for (int i = 0; i < n; i++)
            {
                const float *q = &pts[(size_t)i * 5];
                vec3 p(q[2], q[3], q[4]);
                vec3 c = t->value(q[0], q[1], p);
                float out[4] = {c[0], c[1], c[2], t->alpha(q[0], q[1], p)};
                fwrite(out, sizeof(float), 4, f);
            }
        fclose(f);
        return 0;
    }
    if (mode == "tables")
    {
        FILE *f = fopen(argv[3], "w");
        fprintf(f, "next_random %a\n", random_double());
        for (size_t i = 0; i < b.list.size(); i++)
        {
            instance *in = (instance *)b.list[i];
            Eigen::Matrix4f m = in->transform._transform.matrix();
            Eigen::Matrix4f mi = in->transform.inverse()._transform.matrix();
            fprintf(f, "instance %zu\n fwd", i);
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 4; c++)
                    fprintf(f, " %a", m(r, c));
            fprintf(f, "\n inv");
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 4; c++)
                    fprintf(f, " %a", mi(r, c));
            fprintf(f, "\n bbox %a %a %a %a %a %a\n", in->bbox._min[0], in->bbox._min[1], in->bbox._min[2],
                    in->bbox._max[0], in->bbox._max[1], in->bbox._max[2]);
        }
        fprintf(f, "lights");
        for (auto *l : b.lights)
            fprintf(f, " %d", index_of(b.list, l));
        fprintf(f, "\nbvh\n");
        dump_bvh(f, b, b.world->ptr, 0);
        if (argc >= 6)
        {
            Config c;
            c.film.width = atoi(argv[4]);
            c.film.height = atoi(argv[5]);
            camera cam = make_camera(b, c);
            fprintf(f, "camera %d %d\n", c.film.width, c.film.height);
            const vec3 *vs[] = {&cam.origin, &cam.lower_left_corner, &cam.horizontal, &cam.vertical, &cam.u, &cam.v, &cam.w};
            for (auto *v : vs)
                fprintf(f, " %a %a %a\n", (*v)[0], (*v)[1], (*v)[2]);
            fprintf(f, " %a %a %a\n", cam.lens_radius, cam.time0, cam.time1);
        }
        fclose(f);
        return 0;
    }
    if (argc < 14)
    {
        fprintf(stderr, "missing cfg\n");
        return 2;
    }
    Config config = make_config(argv + 3, (mode == "render" && argc >= 15) ? argv[14] : "/dev/null");
    b.world->config = config; // main.cpp:139
    camera cam = make_camera(b, config);
    if (mode == "render")
    {
        auto t0 = std::chrono::high_resolution_clock::now();
        Integrator *integ = new NEEIterative(config.max_bounces, b.world); // main.cpp:47
        Tiled *r = new Tiled(integ, cam, config);                          // main.cpp:76
        r->start_render(t0);
        while (!r->is_done())
        { // main.cpp:158-163
            r->sync_progress();
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
        r->finalize();
        long rays = 0;
        for (int t = 0; t < config.threads; t++)
            rays += r->bounce_counts[t];
        FILE *f = fopen(argv[13], "wb");
        for (int j = 0; j < config.film.height; j++)
            for (int i = 0; i < config.film.width; i++)
                fwrite(r->framebuffer[j][i].e, sizeof(float), 3, f);
        fclose(f);
        printf("\nREF_RAYS %ld\n", rays);
        return 0;
    }
    if (mode == "samples" || mode == "hits")
    {
        int N = atoi(argv[13]);
        FILE *f = fopen(argv[14], "wb");
        Integrator *integ = new NEEIterative(config.max_bounces, b.world);
        NaiveSpiral spiral(config.film.width, config.film.height, config.block_width, config.block_height);
        int done = 0;
        // same nesting as renderer.h:632-642
        while (!spiral.is_empty() && done < N)
        {
            auto rc = spiral.next();
            for (int s = 0; s < config.samples && done < N; s++)
                for (int j = rc.second.second - 1; j >= rc.first.second && done < N; j--)
                    for (int i = rc.first.first; i < rc.second.first && done < N; i++)
                    {
                        float u = float(i + random_double()) / float(config.film.width);
                        float v = float(j + random_double()) / float(config.film.height);
                        ray r = cam.get_ray(u, v);
                        bool tr = random_double() < config.trace_probability;
                        (void)tr;
                        float rec[13] = {u, v, r.A[0], r.A[1], r.A[2], r.B[0], r.B[1], r.B[2], r._time};
                        if (mode == "samples")
                        {
                            long count = 0;
                            vec3 col = de_nan(integ->color(r, 0, &count, nullptr));
                            rec[9] = col[0];
                            rec[10] = col[1];
                            rec[11] = col[2];
                            rec[12] = (float)count;
                            fwrite(rec, sizeof(float), 13, f);
                        }
                        else
                        {
                            hit_record h;
                            bool hit = b.world->hit(r, 0.001, MAXFLOAT, h);
                            float o[9] = {hit ? 1.f : 0.f, 0, 0, 0, 0, 0, 0, 0, -1};
                            if (hit)
                            {
                                o[1] = h.t;
                                for (int k = 0; k < 3; k++)
                                {
                                    o[2 + k] = h.p[k];
                                    o[5 + k] = h.normal[k];
                                }
                                o[8] = (float)index_of(b.list, h.primitive);
                            }
                            fwrite(o, sizeof(float), 9, f);
                        }
                        done++;
                    }
        }
        fclose(f);
        return 0;
    }
    fprintf(stderr, "unknown mode %s\n", mode.c_str());
    return 2;
}
