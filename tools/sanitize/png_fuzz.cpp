#include <string>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <fstream>
#include <random>
#include "pathtrace_hip.h"
static std::string g_err;
void pth_set_error(const std::string &m) { g_err = m; }
extern "C" const char *pt_last_error(void) { return g_err.c_str(); }
#include "device_stubs.h"
int main(int argc, char **argv)
{
    std::mt19937 rng(12345);
    int ok = 0, rej = 0;
    for (int i = 1; i < argc; i++) {
        std::ifstream f(argv[i], std::ios::binary);
        std::vector<char> orig((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        for (int k = 0; k < 1500; k++) {
            std::vector<char> m = orig;
            int nmut = 1 + rng() % 4;
            for (int j = 0; j < nmut; j++) {
                int kind = rng() % 3;
                if (kind == 0) m[rng() % m.size()] ^= (char)(1 << (rng() % 8));
                else if (kind == 1) m[rng() % m.size()] = (char)rng();
                else m.resize(1 + rng() % m.size());
            }
            std::ofstream o("/tmp/pt_png_fuzz_mut.png", std::ios::binary);
            o.write(m.data(), m.size());
            o.close();
            std::string js = "{\"camera\":{\"look_from\":[0,0,-5],\"look_at\":[0,0,0]},\"world\":{\"color\":[0,0,0]},\"textures\":[{\"id\":\"t\",\"type\":\"png\",\"data\":{\"path\":\"/tmp/pt_png_fuzz_mut.png\"}}],\"materials\":[{\"id\":\"m\",\"type\":\"lambertian\",\"data\":{\"texture\":\"t\"}}],\"primitives\":[],\"instances\":[{\"type\":\"direct\",\"primitive\":{\"type\":\"rect\",\"material\":{\"id\":\"m\"},\"size\":[1,1]}}]}";
            pth_scene *s = pth_scene_from_json(js.c_str(), 8, 8);
            if (s) { ok++; pth_scene_free(s); } else rej++;
        }
    }
    printf("accepted %d rejected %d\n", ok, rej);
    return 0;
}
