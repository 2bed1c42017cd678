// TEST INFRASTRUCTURE -- shared by oracle/ref_driver.cpp and tools/integration/plugin_driver.cpp (both build only in
// the build container, into oracle/_ref/, from the reference sources where they lie; never shipped).  Reads the
// "constructor parameter" text oracle/scene_params.py derives from a scene JSON and calls the REFERENCE's own
// constructors in the order scene_parser.h:241-595 does.  Include after the reference headers.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

static float hx(std::istringstream &ss)
{
    std::string tok;
    ss >> tok;
    return (float)strtod(tok.c_str(), nullptr);
}

struct Built
{
    std::vector<texture *> textures;
    int background_texture = -1;
    std::vector<material *> materials;
    std::vector<hittable *> prims;
    std::vector<hittable *> list; // BVH input order (file order)
    std::vector<hittable *> list_sorted;
    std::vector<hittable *> lights;
    float cam[9];
    vec3 background = vec3(0.8, 0.2, 0.8);
    World *world = nullptr;
};

static Built build(const char *path)
{
    Built b;
    std::ifstream in(path);
    if (!in)
    {
        fprintf(stderr, "cannot open %s\n", path);
        exit(2);
    }
    std::string line;
    while (std::getline(in, line))
    {
        std::istringstream ss(line);
        std::string kind;
        ss >> kind;
        if (kind == "camera")
        {
            for (int i = 0; i < 9; i++)
                b.cam[i] = hx(ss);
        }
        else if (kind == "background")
        {
            float r = hx(ss), g = hx(ss), bl = hx(ss);
            b.background = vec3(r, g, bl);
        }
        else if (kind == "texture")
        {
            std::string t;
            ss >> t;
            if (t == "constant")
            {
                float r = hx(ss), g = hx(ss), bl = hx(ss), alpha = hx(ss);
                b.textures.push_back(new constant_texture(vec3(r, g, bl), alpha));
            }
            else if (t == "checker")
            {
                int even, odd;
                ss >> even >> odd;
                float scale = hx(ss);
                b.textures.push_back(new checker_texture(b.textures[even], b.textures[odd], scale));
            }
            else if (t == "perlin")
            {
                b.textures.push_back(new noise_texture(hx(ss)));
            }
            else if (t == "image")
            {
                int w, h;
                std::string hex;
                ss >> w >> h >> hex;
                std::vector<unsigned char> px(hex.size() / 2);
                for (size_t i = 0; i < px.size(); i++)
                    px[i] = (unsigned char)strtol(hex.substr(2 * i, 2).c_str(), nullptr, 16);
                b.textures.push_back(from_4byte_vector(px, w, h)); // what decode_into_texture does after lodepng
            }
        }
        else if (kind == "background_texture")
        {
            ss >> b.background_texture;
        }
        else if (kind == "material")
        {
            int type;
            ss >> type;
            float r = hx(ss), g = hx(ss), bl = hx(ss), alpha = hx(ss), power = hx(ss);
            int two_sided;
            ss >> two_sided;
            float fuzz = hx(ss), ior = hx(ss);
            int tex = -1;
            ss >> tex;
            vec3 col(r, g, bl);
            material *m = nullptr;
            if (tex >= 0 && type == 0)
            {
                b.materials.push_back(new lambertian(b.textures[tex]));
                continue;
            }
            if (tex >= 0 && type == 3)
            {
                b.materials.push_back(new diffuse_light(b.textures[tex], power, two_sided != 0));
                continue;
            }
            switch (type)
            {
            case 0:
                m = alpha == 1.0f ? new lambertian(col) : new lambertian(new constant_texture(col, alpha));
                break;
            case 1:
                m = new metal(col, fuzz);
                break;
            case 2:
                m = new dielectric(ior);
                break;
            case 3:
                m = alpha == 1.0f ? new diffuse_light(col, power, two_sided != 0)
                                  : new diffuse_light(new constant_texture(col, alpha), power, two_sided != 0);
                break;
            case 4:
                m = new isotropic(col); // placeholder slot; constant_medium makes its own
                break;
            }
            b.materials.push_back(m);
        }
        else if (kind == "prim")
        {
            std::string t;
            int mat;
            ss >> t >> mat;
            if (t == "rect")
            {
                float x0 = hx(ss), z0 = hx(ss), x1 = hx(ss), z1 = hx(ss), y = hx(ss);
                int plane, flipped;
                ss >> plane >> flipped;
                b.prims.push_back(new rect(x0, z0, x1, z1, y, b.materials[mat], (plane_enum)plane, flipped != 0));
            }
            else if (t == "box")
            {
                float v[6];
                for (int i = 0; i < 6; i++)
                    v[i] = hx(ss);
                b.prims.push_back(new box(vec3(v[0], v[1], v[2]), vec3(v[3], v[4], v[5]), b.materials[mat]));
            }
            else if (t == "sphere")
            {
                float cx = hx(ss), cy = hx(ss), cz = hx(ss), r = hx(ss);
                b.prims.push_back(new sphere(vec3(cx, cy, cz), r, b.materials[mat]));
            }
            else if (t == "volume")
            {
                int boundary, phase;
                ss >> boundary;
                float density = hx(ss);
                ss >> phase;
                material *pm = b.materials[phase];
                vec3 color = ((isotropic *)pm)->albedo->value(0, 0, vec3(0, 0, 0));
                constant_medium *cm = new constant_medium(b.prims[boundary], density, color);
                b.prims.push_back(cm);
                b.materials[phase] = cm->phase_function; // the isotropic the medium made for itself (volume.h:14-17)
            }
        }
        else if (kind == "instance")
        {
            int prim;
            ss >> prim;
            float v[9];
            for (int i = 0; i < 9; i++)
                v[i] = hx(ss);
            int is_light;
            ss >> is_light;
            transform3 xf(vec3(v[0], v[1], v[2]), vec3(v[3], v[4], v[5]), vec3(v[6], v[7], v[8]));
            hittable *inst = new instance(b.prims[prim], xf);
            b.list.push_back(inst);
            if (is_light)
                b.lights.push_back(inst);
        }
    }
    b.list_sorted = b.list; // bvh_node's qsort permutes the array it is given
    bvh_node *root = new bvh_node(b.list_sorted.data(), (int)b.list_sorted.size(), 0.0f, 0.0f);
    b.world = new World(root, b.background_texture >= 0 ? b.textures[b.background_texture] : (texture *)new constant_texture(b.background), b.lights);
    return b;
}

