#include <string>
#include <cstdio>
#include <cstdint>
#include "pathtrace_hip.h"
static std::string g_err;
void pth_set_error(const std::string &m) { g_err = m; }
extern "C" const char *pt_last_error(void) { return g_err.c_str(); }
#include "device_stubs.h"
int main(int argc, char **argv)
{
    int bad = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() > 4 && a.substr(a.size() - 4) == ".cfg") {
            pth_config c;
            int rc = pth_config_from_file(a.c_str(), &c);
            printf("%s cfg rc=%d\n", a.c_str(), rc);
            continue;
        }
        pth_scene *s = pth_scene_from_file(a.c_str(), 64, 48);
        if (!s) { printf("%s: rejected: %s\n", a.c_str(), pt_last_error()); bad++; continue; }
        const pt_scene_desc *d = pth_scene_desc(s);
        printf("%s: %d inst %d nodes %d tex\n", a.c_str(), d->n_instances, d->n_nodes, d->n_textures);
        pth_scene_free(s);
    }
    int32_t rects[4 * 1024];
    printf("tiles %d\n", pth_spiral_tiles(3840, 2160, 128, 128, rects, 1024));
    return 0;
}
