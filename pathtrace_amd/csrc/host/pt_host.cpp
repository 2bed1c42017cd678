// Host front end: the reference's input formats and everything main() does before and after the render
// (reference main.cpp:108-168), re-stated for the HIP renderer.  Exposed through the pth_* C ABI of
// include/pathtrace_hip.h.  Compile with -ffp-contract=off: the transforms, bounding boxes and camera built
// here must be bit-identical to the reference's (they are compared against oracle/_ref fixtures in tests/).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/pathtrace_hip.h"
#include "json_min.h"

using pth::Json;

static thread_local std::string g_herr;
extern "C" const char *pt_last_error(void);
// pt_last_error() lives in pt_context.cpp; host errors are reported through the same channel
void pth_set_error(const std::string &m);

namespace {

struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(float t, V3 v) { return v3(t * v.x, t * v.y, t * v.z); }
inline V3 operator/(V3 v, float t) { return v3(v.x / t, v.y / t, v.z / t); }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float length(V3 v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
inline V3 unit(V3 v) { return v / length(v); }

V3 json_vec3(const Json &j) { return v3(j.at(0).as_float(), j.at(1).as_float(), j.at(2).as_float()); }   // scene_parser.h:34-37
const V3 MAUVE = {0.8f, 0.2f, 0.8f};                                                                   // scene_parser.h:16

// ---- process-wide generator of the reference (random.h:9-15): mt19937(5489) + generate_canonical<double,53>
struct Mt19937 {
    uint32_t mt[624];
    int idx;
    explicit Mt19937(uint32_t seed = 5489u)
    {
        mt[0] = seed;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    uint32_t next()
    {
        if (idx >= 624) {
            for (int i = 0; i < 624; i++) {
                uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    double random_double()
    {
        double lo = (double)next(), hi = (double)next();
        double r = (lo + hi * 4294967296.0) / 18446744073709551616.0;
        if (r >= 1.0) r = std::nextafter(1.0, 0.0);
        return r;
    }
};
// texture.h:180-183: the static initialisers consume 256*3 + 3*255 = 1533 values before main() (SceneBuilder::build)

// ---- transform3 (transform3.h:19-68) through the Eigen 3.2.10 code paths it instantiates
struct Quat { float x, y, z, w; };
Quat angle_axis(float angle, float ax, float ay, float az)
{
    float ha = 0.5f * angle;
    float s = std::sin(ha);
    return Quat{s * ax, s * ay, s * az, std::cos(ha)};
}
Quat qmul(Quat a, Quat b)
{   // Eigen/src/Geometry/arch/Geometry_SSE.h:19-41, lane by lane
    Quat r;
    r.x = (a.x * b.w - a.z * b.y) + (a.y * b.z + a.w * b.x);
    r.y = (a.y * b.w - a.x * b.z) + (a.z * b.x + a.w * b.y);
    r.z = (a.z * b.w - a.y * b.x) + (a.x * b.y + a.w * b.z);
    r.w = (a.w * b.w - a.x * b.x) + (-(a.z * b.z) + -(a.y * b.y));
    return r;
}
void compose(const V3 &scale, const V3 &rotate, const V3 &translate, float m[12])
{   // t_translate * (AAx * AAy * AAz) * t_scale, rotations in units of pi (transform3.h:21-24)
    Quat q = qmul(qmul(angle_axis((float)(rotate.x * M_PI), 1, 0, 0), angle_axis((float)(rotate.y * M_PI), 0, 1, 0)),
                  angle_axis((float)(rotate.z * M_PI), 0, 0, 1));
    float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;   // Quaternion.h:525-557
    float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    float R[3][3] = {{1.0f - (tyy + tzz), txy - twz, txz + twy},
                     {txy + twz, 1.0f - (txx + tzz), tyz - twx},
                     {txz - twy, tyz + twx, 1.0f - (txx + tyy)}};
    const float s[3] = {scale.x, scale.y, scale.z}, t[3] = {translate.x, translate.y, translate.z};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) m[4 * i + j] = R[i][j] * s[j];
        m[4 * i + 3] = 0.0f + t[i];
    }
}
void invert(const float f[12], float r[12])
{   // Transform::inverse(Affine) Transform.h:1158-1184; 3x3 by cofactors LU/Inverse.h:116-159
    auto M = [&](int i, int j) { return f[4 * i + j]; };
    auto cof = [&](int i, int j) {
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
    };
    float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    float det = c0 * M(0, 0) + (c1 * M(1, 0) + c2 * M(2, 0));
    float invdet = 1.0f / det;
    r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
    r[4] = cof(0, 1) * invdet; r[5] = cof(1, 1) * invdet; r[6] = cof(2, 1) * invdet;
    r[8] = cof(0, 2) * invdet; r[9] = cof(1, 2) * invdet; r[10] = cof(2, 2) * invdet;
    for (int i = 0; i < 3; i++)
        r[4 * i + 3] = ((-r[4 * i + 0]) * f[3] + (-r[4 * i + 1]) * f[7]) + (-r[4 * i + 2]) * f[11];
}
V3 xf_point(const float m[12], V3 p)
{   // transform3::operator* (transform3.h:65-68)
    return v3(m[3] + ((m[0] * p.x + m[1] * p.y) + m[2] * p.z), m[7] + ((m[4] * p.x + m[5] * p.y) + m[6] * p.z),
              m[11] + ((m[8] * p.x + m[9] * p.y) + m[10] * p.z));
}

struct Box { V3 mn, mx; };
inline float ffmin(float a, float b) { return a < b ? a : b; }
inline float ffmax(float a, float b) { return a > b ? a : b; }
Box surround(Box a, Box b)
{   // aabb.h:55-64
    return Box{v3(ffmin(a.mn.x, b.mn.x), ffmin(a.mn.y, b.mn.y), ffmin(a.mn.z, b.mn.z)),
               v3(ffmax(a.mx.x, b.mx.x), ffmax(a.mx.y, b.mx.y), ffmax(a.mx.z, b.mx.z))};
}
V3 shuffle(V3 v, int plane)
{   // primitive.h:104-121
    if (plane == PT_PLANE_XY) return v3(v.x, v.z, v.y);
    if (plane == PT_PLANE_YZ) return v3(v.y, v.x, v.z);
    return v;
}
Box rect_bbox(float x0, float z0, float x1, float z1, float y, int plane)
{   // primitive.h:140-149
    return Box{shuffle(v3(x0, (float)(y - 0.001), z0), plane), shuffle(v3(x1, (float)(y + 0.001), z1), plane)};
}

// ---- PNG -> RGBA8, i.e. what lodepng::decode(image, w, h, path) hands to from_4byte_vector (scene_parser.h:39-55).
// lodepng is an un-vendored submodule of the reference (thirdparty/lodepng, .gitmodules); this is a plain reader of
// the published formats (RFC 1950/1951 inflate, PNG filters) for non-interlaced 8-bit images of all five colour types.
struct BitReader {
    const uint8_t *p; size_t n, pos = 0; uint32_t acc = 0; int cnt = 0;
    BitReader(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    uint32_t bits(int k)
    {
        while (cnt < k) {
            if (pos >= n) throw pth::JsonError("PNG: truncated deflate stream");
            acc |= (uint32_t)p[pos++] << cnt; cnt += 8;
        }
        uint32_t v = acc & ((1u << k) - 1u);
        acc >>= k; cnt -= k;
        return v;
    }
    void align() { acc = 0; cnt = 0; }
};
struct Huffman {
    uint16_t count[16] = {0}, symbol[320] = {0};
    void build(const uint8_t *len, int n)
    {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16]; offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(BitReader &br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= (int)br.bits(1);
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        throw pth::JsonError("PNG: bad Huffman code");
    }
};
std::vector<uint8_t> inflate_zlib(const std::vector<uint8_t> &z)
{
    if (z.size() < 6 || (z[0] & 15) != 8) throw pth::JsonError("PNG: not a zlib stream");
    BitReader br(z.data() + 2, z.size() - 2);
    std::vector<uint8_t> out;
    static const int lbase[] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const int lext[] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    static const int dbase[] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    static const int dext[] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    for (bool last = false; !last;) {
        last = br.bits(1);
        const int type = (int)br.bits(2);
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) throw pth::JsonError("PNG: truncated stored block");
            size_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > br.n) throw pth::JsonError("PNG: truncated stored block");
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
            continue;
        }
        if (type == 3) throw pth::JsonError("PNG: bad deflate block type");
        Huffman lit, dist;
        uint8_t lens[320];
        if (type == 1) {
            for (int i = 0; i < 288; i++) lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
            lit.build(lens, 288);
            for (int i = 0; i < 30; i++) lens[i] = 5;
            dist.build(lens, 30);
        } else {
            const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            static const int order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
            Huffman lc;
            lc.build(cl, 19);
            int i = 0;
            while (i < nlen + ndist) {
                int sym = lc.decode(br);
                if (sym < 16) lens[i++] = (uint8_t)sym;
                else {
                    int rep, val = 0;
                    if (sym == 16) { if (!i) throw pth::JsonError("PNG: bad code lengths"); val = lens[i - 1]; rep = 3 + (int)br.bits(2); }
                    else if (sym == 17) rep = 3 + (int)br.bits(3);
                    else rep = 11 + (int)br.bits(7);
                    if (i + rep > nlen + ndist) throw pth::JsonError("PNG: bad code lengths");
                    while (rep--) lens[i++] = (uint8_t)val;
                }
            }
            lit.build(lens, nlen);
            dist.build(lens + nlen, ndist);
        }
        for (;;) {
            int sym = lit.decode(br);
            if (sym < 256) out.push_back((uint8_t)sym);
            else if (sym == 256) break;
            else {
                sym -= 257;
                if (sym >= 29) throw pth::JsonError("PNG: bad length symbol");
                int len = lbase[sym] + (int)br.bits(lext[sym]);
                int ds = dist.decode(br);
                if (ds >= 30) throw pth::JsonError("PNG: bad distance symbol");
                size_t d = (size_t)dbase[ds] + br.bits(dext[ds]);
                if (d > out.size()) throw pth::JsonError("PNG: distance too far back");
                for (int k = 0; k < len; k++) out.push_back(out[out.size() - d]);
            }
        }
    }
    {   // zlib trailer: Adler-32 of the inflated bytes (RFC 1950), big-endian after the deflate stream
        br.align();
        if (br.pos + 4 > br.n) throw pth::JsonError("PNG: zlib trailer missing");
        uint32_t a = 1, bsum = 0;
        for (uint8_t v : out) { a = (a + v) % 65521u; bsum = (bsum + a) % 65521u; }
        const uint8_t *t = br.p + br.pos;
        const uint32_t want = ((uint32_t)t[0] << 24) | (t[1] << 16) | (t[2] << 8) | t[3];
        if (((bsum << 16) | a) != want) throw pth::JsonError("PNG: Adler-32 mismatch in the image data");
    }
    return out;
}
std::string read_file(const std::string &path);
std::vector<uint8_t> decode_png(const std::string &path, int &width, int &height)
{
    const std::string raw = read_file(path);
    const uint8_t *b = (const uint8_t *)raw.data();
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (raw.size() < 8 || memcmp(b, sig, 8)) throw pth::JsonError("not a PNG file: " + path);
    auto be32 = [&](size_t o) { return ((uint32_t)b[o] << 24) | (b[o + 1] << 16) | (b[o + 2] << 8) | b[o + 3]; };
    std::vector<uint8_t> idat, plte, trns;
    int depth = 0, ctype = 0, interlace = 0;
    bool have_hdr = false;
    auto crc32 = [](const uint8_t *p, size_t n) {   // PNG chunk CRC (ISO 3309); lodepng checks it by default
        uint32_t c = 0xffffffffu;
        for (size_t i = 0; i < n; i++) {
            c ^= p[i];
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
        }
        return c ^ 0xffffffffu;
    };
    bool ended = false;
    for (size_t pos = 8; pos + 12 <= raw.size();) {
        const uint32_t n = be32(pos);
        const std::string typ(raw, pos + 4, 4);
        if (pos + 12 + (size_t)n > raw.size()) throw pth::JsonError("PNG: truncated chunk in " + path);
        if (crc32(b + pos + 4, 4 + (size_t)n) != be32(pos + 8 + n)) throw pth::JsonError("PNG: chunk CRC mismatch in " + path);
        const uint8_t *body = b + pos + 8;
        if (typ == "IHDR" && n >= 13) {
            width = (int)be32(pos + 8); height = (int)be32(pos + 12);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            have_hdr = true;
        } else if (typ == "PLTE") plte.assign(body, body + n);
        else if (typ == "tRNS") trns.assign(body, body + n);
        else if (typ == "IDAT") idat.insert(idat.end(), body, body + n);
        else if (typ == "IEND") { ended = true; break; }
        pos += 12 + n;
    }
    if (!have_hdr || width < 1 || height < 1) throw pth::JsonError("PNG: no IHDR in " + path);
    if (!ended) throw pth::JsonError("PNG: no IEND chunk in " + path);
    if (depth != 8 || interlace != 0 || !(ctype == 0 || ctype == 2 || ctype == 3 || ctype == 4 || ctype == 6))
        throw pth::JsonError("PNG: only non-interlaced 8-bit images are supported (" + path + ")");
    const int ch = ctype == 0 ? 1 : (ctype == 2 ? 3 : (ctype == 3 ? 1 : (ctype == 4 ? 2 : 4)));
    const size_t stride = (size_t)width * ch;
    std::vector<uint8_t> data = inflate_zlib(idat);
    if (data.size() < (stride + 1) * (size_t)height) throw pth::JsonError("PNG: image data too short in " + path);
    std::vector<uint8_t> px(stride * height), prev(stride, 0);
    for (int y = 0; y < height; y++) {
        const uint8_t ft = data[(stride + 1) * y];
        const uint8_t *line = &data[(stride + 1) * y + 1];
        uint8_t *cur = &px[stride * y];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, bb = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int pr = 0;
            if (ft == 1) pr = a;
            else if (ft == 2) pr = bb;
            else if (ft == 3) pr = (a + bb) >> 1;
            else if (ft == 4) {
                const int pa = std::abs(bb - c), pb = std::abs(a - c), pc = std::abs(a + bb - 2 * c);
                pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c);
            } else if (ft != 0) throw pth::JsonError("PNG: bad filter type in " + path);
            cur[i] = (uint8_t)(line[i] + pr);
        }
        memcpy(prev.data(), cur, stride);
    }
    std::vector<uint8_t> out((size_t)width * height * 4);
    for (size_t i = 0; i < (size_t)width * height; i++) {
        uint8_t r, g, bl, a = 255;
        const uint8_t *q = &px[i * ch];
        if (ctype == 0) { r = g = bl = q[0]; if (trns.size() >= 2 && q[0] == trns[1]) a = 0; }
        else if (ctype == 4) { r = g = bl = q[0]; a = q[1]; }
        else if (ctype == 2) { r = q[0]; g = q[1]; bl = q[2]; if (trns.size() >= 6 && r == trns[1] && g == trns[3] && bl == trns[5]) a = 0; }
        else if (ctype == 6) { r = q[0]; g = q[1]; bl = q[2]; a = q[3]; }
        else {
            if ((size_t)q[0] * 3 + 2 >= plte.size()) throw pth::JsonError("PNG: palette index out of range in " + path);
            r = plte[q[0] * 3]; g = plte[q[0] * 3 + 1]; bl = plte[q[0] * 3 + 2];
            if (q[0] < trns.size()) a = trns[q[0]];
        }
        out[4 * i] = r; out[4 * i + 1] = g; out[4 * i + 2] = bl; out[4 * i + 3] = a;
    }
    return out;
}

struct WrappedMaterial { int index; std::string type; };   // scene_parser.h:57-69
struct WrappedPrim { int index; WrappedMaterial mat; };    // scene_parser.h:73-89

}  // namespace

struct pth_scene {
    std::vector<pt_material> materials;
    std::vector<pt_primitive> primitives;
    std::vector<pt_instance> instances;
    std::vector<pt_bvh_node> nodes;
    std::vector<int32_t> lights;
    std::vector<pt_texture> textures;
    std::vector<uint8_t> texels;
    std::vector<float> perlin_ranvec;    // 256 x 3
    std::vector<int32_t> perlin_perm;    // 3 x 256
    pt_scene_desc desc{};
};

namespace {

struct SceneBuilder {
    pth_scene &sc;
    std::map<std::string, int> textures;   // id -> index into sc.textures (std::map::emplace: the first id wins)
    int error_tex = -1;
    std::string base_dir = ".";            // relative PNG paths are resolved against it (the reference: its working directory)
    std::map<std::string, WrappedMaterial> materials;
    std::map<std::string, WrappedPrim> prims;
    int last_id = 0;
    int error_mat = -1;
    explicit SceneBuilder(pth_scene &s) : sc(s) {}

    std::string new_id() { return std::to_string(last_id++); }   // scene_parser.h:26-32
    int add_material(int type, V3 color, float alpha = 1.0f, float power = 1.0f, bool two_sided = true, float fuzz = 0.0f, float ior = 1.45f)
    {
        pt_material m{};
        m.type = type; m.color[0] = color.x; m.color[1] = color.y; m.color[2] = color.z;
        m.alpha = alpha; m.power = power; m.two_sided = two_sided; m.fuzz = fuzz; m.ior = ior;
        m.texture = -1;
        sc.materials.push_back(m);
        return (int)sc.materials.size() - 1;
    }
    WrappedMaterial error_material()
    {   // scene_parser.h:20-24, 92-96
        if (error_mat < 0) error_mat = add_material(PT_MAT_LAMBERTIAN, MAUVE);
        return WrappedMaterial{error_mat, "lambertian"};
    }

    int add_texture(int type, V3 color = v3(0, 0, 0), float alpha = 1.0f, int even = -1, int odd = -1, float scale = 1.0f)
    {
        pt_texture t{};
        t.type = type; t.color[0] = color.x; t.color[1] = color.y; t.color[2] = color.z; t.alpha = alpha;
        t.even = even; t.odd = odd; t.scale = scale;
        sc.textures.push_back(t);
        return (int)sc.textures.size() - 1;
    }
    int error_texture()
    {   // scene_parser.h:98-102: one shared mauve constant_texture
        if (error_tex < 0) error_tex = add_texture(PT_TEX_CONSTANT, MAUVE);
        return error_tex;
    }
    int texture_by_id(const std::string &id)
    {   // the reference's textures[id] default-constructs a null pointer for an unknown id and crashes later
        auto it = textures.find(id);
        if (it == textures.end()) throw pth::JsonError("unknown texture '" + id + "'");
        return it->second;
    }
    void parse_textures(const Json &scene)
    {   // scene_parser.h:263-330
        for (const Json &el : scene["textures"].arr) {
            if (el.value_bool("skip", false)) continue;
            if (!el.contains("id")) throw pth::JsonError("texture without id");
            std::string id = el["id"].as_string();
            if (!el.contains("data")) { textures.emplace(id, error_texture()); continue; }
            const Json &d = el["data"];
            std::string type = el["type"].as_string();
            if (type != "checker" && type != "perlin" && type != "png") type = "constant";   // scene.h:71-79: map operator[] -> enum 0
            int idx;
            if (type == "constant") {
                idx = add_texture(PT_TEX_CONSTANT, json_vec3(d["color"]), (float)d.value("alpha", 1.0));
            } else if (type == "checker") {
                auto child = [&](const Json &c) {
                    if (c.contains("texture")) return texture_by_id(c["texture"].as_string());
                    return add_texture(PT_TEX_CONSTANT, json_vec3(c["color"]));
                };
                int odd = child(d["odd"]);      // scene_parser.h:292-309: odd, then even
                int even = child(d["even"]);
                idx = add_texture(PT_TEX_CHECKER, v3(0, 0, 0), 1.0f, even, odd, d["scale"].as_float());
            } else if (type == "perlin") {
                idx = add_texture(PT_TEX_PERLIN, v3(0, 0, 0), 1.0f, -1, -1, (float)d.value("scale", 1.0));
            } else {
                std::string path = d["path"].as_string();
                if (!path.empty() && path[0] != '/') path = base_dir + "/" + path;
                int w = 0, h = 0;
                std::vector<uint8_t> rgba = decode_png(path, w, h);   // lodepng::decode(image, w, h, path): RGBA8
                idx = add_texture(PT_TEX_IMAGE);
                sc.textures[idx].width = w; sc.textures[idx].height = h;
                sc.textures[idx].texel_offset = (int64_t)sc.texels.size();
                sc.texels.insert(sc.texels.end(), rgba.begin(), rgba.end());
            }
            textures.emplace(id, idx);
        }
    }
    // material / background reference to a texture id: a constant texture is folded into (colour, alpha), anything
    // else is carried as an index
    struct TexRef { V3 color; float alpha; int index; };
    TexRef texture_ref(const std::string &id)
    {
        int idx = texture_by_id(id);
        const pt_texture &t = sc.textures[idx];
        if (t.type == PT_TEX_CONSTANT) return TexRef{v3(t.color[0], t.color[1], t.color[2]), t.alpha, -1};
        return TexRef{v3(0, 0, 0), 1.0f, idx};
    }
    void parse_materials(const Json &scene)
    {   // scene_parser.h:332-447
        for (const Json &el : scene["materials"].arr) {
            if (el.value_bool("skip", false)) continue;
            if (!el.contains("id")) throw pth::JsonError("material without id");
            std::string id = el["id"].as_string();
            if (!el.contains("data")) { materials.emplace(id, error_material()); continue; }
            const Json &d = el["data"];
            std::string type = el["type"].as_string();
            static const char *known[] = {"lambertian", "metal", "dielectric", "isotropic", "diffuse_light"};
            if (std::find(known, known + 5, type) == known + 5) type = "lambertian";   // map operator[] -> enum 0
            if (type == "lambertian") {
                if (d.contains("color")) materials.emplace(id, WrappedMaterial{add_material(PT_MAT_LAMBERTIAN, json_vec3(d["color"])), "lambertian"});
                else if (d.contains("texture")) {
                    TexRef t = texture_ref(d["texture"].as_string());
                    int mi = add_material(PT_MAT_LAMBERTIAN, t.color, t.alpha);
                    sc.materials[mi].texture = t.index;
                    materials.emplace(id, WrappedMaterial{mi, "lambertian"});
                } else materials.emplace(id, error_material());
            } else if (type == "metal") {
                V3 c = d.contains("color") ? json_vec3(d["color"]) : v3(1, 1, 1);
                float f = (float)d.value("roughness", 0.0);
                materials.emplace(id, WrappedMaterial{add_material(PT_MAT_METAL, c, 1.0f, 1.0f, true, f < 1 ? f : 1.0f), "metal"});
            } else if (type == "dielectric") {
                float ri = d.contains("ior") ? d["ior"].as_float() : 1.450f;
                materials.emplace(id, WrappedMaterial{add_material(PT_MAT_DIELECTRIC, v3(1, 1, 1), 1.0f, 1.0f, true, 0.0f, ri), "dielectric"});
            } else if (type == "diffuse_light") {
                float power = d.contains("power") ? d["power"].as_float() : 1.0f;
                bool two_sided = d.value_bool("two_sided", true);
                V3 c;
                float a = 1.0f;
                int ti = -1;
                if (d.contains("texture")) { TexRef t = texture_ref(d["texture"].as_string()); c = t.color; a = t.alpha; ti = t.index; }
                else c = d.contains("color") ? json_vec3(d["color"]) : v3(1, 1, 1);
                int mi = add_material(PT_MAT_DIFFUSE_LIGHT, c, a, power, two_sided);
                sc.materials[mi].texture = ti;
                materials.emplace(id, WrappedMaterial{mi, "diffuse_light"});
            }
            // "isotropic": no case in the reference's switch (scene_parser.h:444-445) -> silently dropped
        }
    }
    WrappedPrim parse_prim(const Json &el)
    {   // parse_prim_or_instance scene_parser.h:104-239
        WrappedMaterial mat;
        if (el.contains("material") && el["material"].contains("id")) {
            auto it = materials.find(el["material"]["id"].as_string());
            if (it == materials.end()) throw pth::JsonError("primitive refers to unknown material '" + el["material"]["id"].as_string() + "'");
            mat = it->second;
        } else mat = error_material();
        const std::string &type = el["type"].as_string();
        pt_primitive p{};
        p.material = mat.index;
        p.boundary = -1; p.phase_material = -1;
        if (type == "sphere") {
            p.type = PT_PRIM_SPHERE;
            p.radius = el.contains("radius") ? el["radius"].as_float() : 1.0f;
            V3 o = el.contains("origin") ? json_vec3(el["origin"]) : v3(0, 0, 0);
            p.center[0] = o.x; p.center[1] = o.y; p.center[2] = o.z;
        } else if (type == "rect") {
            p.type = PT_PRIM_RECT;
            p.plane = PT_PLANE_XZ;
            p.flipped = el.value_bool("flip", false);
            if (el.contains("align")) {
                const std::string &a = el["align"].as_string();
                p.plane = a == "xz" ? PT_PLANE_XZ : (a == "yz" ? PT_PLANE_YZ : PT_PLANE_XY);   // unknown string -> enum 0
            }
            if (el.contains("a0") && el.contains("b0") && el.contains("a1") && el.contains("b1")) {
                p.rect[0] = el["a0"].as_float(); p.rect[1] = el["b0"].as_float();
                p.rect[2] = el["a1"].as_float(); p.rect[3] = el["b1"].as_float();
                p.rect[4] = el["c"].as_float();
            } else {
                float a = 1.0f, b = 1.0f;
                if (el.contains("size")) { a = el["size"].at(0).as_float(); b = el["size"].at(1).as_float(); }
                // rect(x, z, ...) : rect(-x / 2.0, -z / 2.0, x / 2.0, z / 2.0, 0.0, ...)  primitive.h:126-130
                p.rect[0] = (float)(-a / 2.0); p.rect[1] = (float)(-b / 2.0);
                p.rect[2] = (float)(a / 2.0); p.rect[3] = (float)(b / 2.0); p.rect[4] = 0.0f;
            }
        } else if (type == "box") {
            p.type = PT_PRIM_BOX;
            if (el.contains("p0") && el.contains("p1")) {
                V3 a = json_vec3(el["p0"]), b = json_vec3(el["p1"]);
                p.p0[0] = a.x; p.p0[1] = a.y; p.p0[2] = a.z; p.p1[0] = b.x; p.p1[1] = b.y; p.p1[2] = b.z;
            } else {
                V3 s = el.contains("size") ? json_vec3(el["size"]) : v3(1, 1, 1);
                // box(w,h,d) : box(vec3(-w/2,-h/2,-d/2), vec3(w/2,h/2,d/2))  primitive.h:230
                p.p0[0] = -s.x / 2; p.p0[1] = -s.y / 2; p.p0[2] = -s.z / 2;
                p.p1[0] = s.x / 2; p.p1[1] = s.y / 2; p.p1[2] = s.z / 2;
            }
        } else if (type == "volume") {
            p.type = PT_PRIM_VOLUME;
            auto it = prims.find(el["primitive"].as_string());
            if (it == prims.end()) throw pth::JsonError("volume refers to an unknown boundary primitive");
            p.boundary = it->second.index;
            p.density = el["density"].as_float();
            V3 color = el.contains("color") ? json_vec3(el["color"]) : MAUVE;
            p.phase_material = add_material(PT_MAT_ISOTROPIC, color);   // constant_medium ctor volume.h:14-17
            mat = it->second.mat;                                       // scene_parser.h:231
            p.material = mat.index;
        } else {
            throw pth::JsonError("primitive type '" + type + "' is outside the hot-path scope");
        }
        sc.primitives.push_back(p);
        return WrappedPrim{(int)sc.primitives.size() - 1, mat};
    }

    Box prim_bbox(int pi) const
    {
        const pt_primitive &p = sc.primitives[pi];
        switch (p.type) {
        case PT_PRIM_RECT: return rect_bbox(p.rect[0], p.rect[1], p.rect[2], p.rect[3], p.rect[4], p.plane);
        case PT_PRIM_BOX: {   // box::bounding_box -> hittable_list::bounding_box (primitive.h:247-252, hittable_list.h:40-68)
            const float *a = p.p0, *b = p.p1;
            Box bb = rect_bbox(a[0], a[1], b[0], b[1], a[2], PT_PLANE_XY);
            bb = surround(bb, rect_bbox(a[0], a[1], b[0], b[1], b[2], PT_PLANE_XY));
            bb = surround(bb, rect_bbox(a[1], a[2], b[1], b[2], a[0], PT_PLANE_YZ));
            bb = surround(bb, rect_bbox(a[1], a[2], b[1], b[2], b[0], PT_PLANE_YZ));
            bb = surround(bb, rect_bbox(a[0], a[2], b[0], b[2], a[1], PT_PLANE_XZ));
            bb = surround(bb, rect_bbox(a[0], a[2], b[0], b[2], b[1], PT_PLANE_XZ));
            return bb;
        }
        case PT_PRIM_SPHERE: {   // primitive.h:97-102
            V3 c = v3(p.center[0], p.center[1], p.center[2]), r = v3(p.radius, p.radius, p.radius);
            return Box{c - r, c + r};
        }
        default: return prim_bbox(p.boundary);   // volume.h:20-23
        }
    }

    void add_instance(const WrappedPrim &wp, const V3 &scale, const V3 &rotate, const V3 &translate)
    {   // instance ctor primitive.h:266-296
        pt_instance in{};
        in.primitive = wp.index;
        compose(scale, rotate, translate, in.fwd);
        invert(in.fwd, in.inv);
        Box pb = prim_bbox(wp.index);
        V3 mn = v3(FLT_MAX, FLT_MAX, FLT_MAX), mx = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    float x = i * pb.mx.x + (1 - i) * pb.mn.x;
                    float y = j * pb.mx.y + (1 - j) * pb.mn.y;
                    float z = k * pb.mx.z + (1 - k) * pb.mn.z;
                    V3 t = xf_point(in.fwd, v3(x, y, z));
                    if (t.x > mx.x) mx.x = t.x;
                    if (t.x < mn.x) mn.x = t.x;
                    if (t.y > mx.y) mx.y = t.y;
                    if (t.y < mn.y) mn.y = t.y;
                    if (t.z > mx.z) mx.z = t.z;
                    if (t.z < mn.z) mn.z = t.z;
                }
        in.bbox[0] = mn.x; in.bbox[1] = mn.y; in.bbox[2] = mn.z; in.bbox[3] = mx.x; in.bbox[4] = mx.y; in.bbox[5] = mx.z;
        sc.instances.push_back(in);
        if (wp.mat.type == "diffuse_light") sc.lights.push_back((int)sc.instances.size() - 1);   // scene_parser.h:541-549
    }

    // ---- bvh_node ctor (bvh.h:133-175); qsort = glibc merge sort, comparator returns -1/+1 only (bvh.h:71-131)
    Mt19937 rng;
    int cmp(int a, int b, int axis) const
    {
        float l = sc.instances[a].bbox[axis], r = sc.instances[b].bbox[axis];
        return (l - r < 0.0) ? -1 : 1;
    }
    void msort(int *b, int n, int axis, int *tmp) const
    {
        if (n <= 1) return;
        int n1 = n / 2, n2 = n - n1;
        int *b1 = b, *b2 = b + n1;
        msort(b1, n1, axis, tmp);
        msort(b2, n2, axis, tmp);
        int *t = tmp;
        while (n1 > 0 && n2 > 0) {
            if (cmp(*b1, *b2, axis) <= 0) { *t++ = *b1++; n1--; }
            else { *t++ = *b2++; n2--; }
        }
        if (n1 > 0) memcpy(t, b1, (size_t)n1 * sizeof(int));
        memcpy(b, tmp, (size_t)(n - n2) * sizeof(int));
    }
    Box child_box(int c) const
    {
        const float *b = c >= 0 ? sc.nodes[c].bbox : sc.instances[~c].bbox;
        return Box{v3(b[0], b[1], b[2]), v3(b[3], b[4], b[5])};
    }
    int build_node(int *l, int n, int *tmp)
    {
        int me = (int)sc.nodes.size();
        sc.nodes.push_back(pt_bvh_node{});
        int axis = int(3 * rng.random_double());
        msort(l, n, axis == 0 ? 0 : (axis == 1 ? 1 : 2), tmp);
        int left, right;
        if (n == 1) left = right = ~l[0];
        else if (n == 2) { left = ~l[0]; right = ~l[1]; }
        else {
            left = build_node(l, n / 2, tmp);
            right = build_node(l + n / 2, n - n / 2, tmp);
        }
        Box bb = surround(child_box(left), child_box(right));
        pt_bvh_node &nd = sc.nodes[me];
        nd.left = left; nd.right = right;
        nd.bbox[0] = bb.mn.x; nd.bbox[1] = bb.mn.y; nd.bbox[2] = bb.mn.z; nd.bbox[3] = bb.mx.x; nd.bbox[4] = bb.mx.y; nd.bbox[5] = bb.mx.z;
        return me;
    }

    void build(const Json &scene, int width, int height)
    {   // build_scene scene_parser.h:241-595
        if (scene["assets"].is_array())
            for (const Json &el : scene["assets"].arr) {
                if (el.value_bool("skip", false)) continue;
                if (el["type"].as_string() != "object") throw pth::JsonError("asset type must be 'object'");
            }
        parse_textures(scene);
        parse_materials(scene);
        for (const Json &el : scene["primitives"].arr) {
            std::string id = el.contains("id") ? el["id"].as_string() : new_id();
            prims.emplace(id, parse_prim(el));
        }
        std::vector<std::string> inst_prim;
        for (const Json &el : scene["instances"].arr) {
            if (el["type"].as_string() == "ref") { inst_prim.push_back(el["primitive"]["id"].as_string()); continue; }
            std::string id = new_id();
            prims.emplace(id, parse_prim(el["primitive"]));   // built even when the instance is skipped
            inst_prim.push_back(id);
        }
        size_t k = 0;
        for (const Json &el : scene["instances"].arr) {
            const std::string &pid = inst_prim[k++];
            if (el.value_bool("skip", false)) continue;
            V3 scale = v3(1, 1, 1), rotate = v3(0, 0, 0), translate = v3(0, 0, 0);
            if (el.contains("transform")) {
                const Json &t = el["transform"];
                if (t.contains("scale") && t["scale"].is_array()) scale = json_vec3(t["scale"]);
                else { float f = (float)t.value("scale", 1.0); scale = v3(f, f, f); }
                if (t.contains("rotate")) rotate = json_vec3(t["rotate"]);
                if (t.contains("translate")) translate = json_vec3(t["translate"]);
            }
            auto it = prims.find(pid);
            if (it == prims.end()) throw pth::JsonError("instance refers to unknown primitive '" + pid + "'");
            add_instance(it->second, scale, rotate, translate);
        }
        if (sc.instances.empty()) throw pth::JsonError("scene has no instances");
        V3 bg = MAUVE;
        int bg_tex = -1;
        if (scene.contains("world")) {
            const Json &w = scene["world"];
            if (w.contains("texture")) { TexRef t = texture_ref(w["texture"].as_string()); bg = t.color; bg_tex = t.index; }
            else if (w.contains("color")) bg = json_vec3(w["color"]);
        }
        // the reference's generator state when main() reaches new bvh_node(...): seed 5489, then the static
        // initialisers of texture.h:180-183 -- perlin_generate() (256 x 3 draws, :99-110) and three
        // perlin_generate_perm() (255 draws each, :76-97) = PERLIN_STATIC_DRAWS values, which ARE the Perlin tables
        rng = Mt19937(5489u);
        sc.perlin_ranvec.resize(768);
        sc.perlin_perm.resize(768);
        for (int i = 0; i < 256; i++) {
            double xr = 2 * rng.random_double() - 1;
            double yr = 2 * rng.random_double() - 1;
            double zr = 2 * rng.random_double() - 1;
            V3 rv = unit(v3((float)xr, (float)yr, (float)zr));
            sc.perlin_ranvec[3 * i] = rv.x; sc.perlin_ranvec[3 * i + 1] = rv.y; sc.perlin_ranvec[3 * i + 2] = rv.z;
        }
        for (int k = 0; k < 3; k++) {
            int32_t *pm = &sc.perlin_perm[256 * k];
            for (int i = 0; i < 256; i++) pm[i] = i;
            for (int i = 255; i > 0; i--) {
                int target = int(rng.random_double() * (i + 1));
                std::swap(pm[i], pm[target]);
            }
        }
        std::vector<int> order(sc.instances.size()), tmp(sc.instances.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
        build_node(order.data(), (int)order.size(), tmp.data());

        // camera: setup_camera main.cpp:86-104 + camera ctor camera.h:9-36
        const Json &cj = scene["camera"];
        V3 lookfrom = json_vec3(cj["look_from"]), lookat = json_vec3(cj["look_at"]), vup = v3(0, 1, 0);
        float vfov = (float)cj.value("fov", 30.0), aperture = (float)cj.value("aperture", 0.0);
        float focus_dist = (float)cj.value("dist_to_focus", 10.0);
        float aspect = float(width) / float(height);
        pt_camera cam{};
        cam.lens_radius = aperture / 2;
        float theta = (float)(vfov * M_PI / 180);
        float half_height = std::tan(theta / 2);
        float half_width = aspect * half_height;
        V3 w = unit(lookfrom - lookat);
        V3 u = unit(cross(vup, w));
        V3 v = cross(w, u);
        V3 llc = lookfrom - (half_width * focus_dist) * u - (half_height * focus_dist) * v - focus_dist * w;
        V3 hor = (2 * half_width * focus_dist) * u;
        V3 ver = (2 * half_height * focus_dist) * v;
        auto put = [](float *d, V3 a) { d[0] = a.x; d[1] = a.y; d[2] = a.z; };
        put(cam.origin, lookfrom); put(cam.lower_left_corner, llc); put(cam.horizontal, hor); put(cam.vertical, ver);
        put(cam.u, u); put(cam.v, v); put(cam.w, w);

        pt_scene_desc &d = sc.desc;
        d.n_materials = (int)sc.materials.size(); d.materials = sc.materials.data();
        d.n_primitives = (int)sc.primitives.size(); d.primitives = sc.primitives.data();
        d.n_instances = (int)sc.instances.size(); d.instances = sc.instances.data();
        d.n_nodes = (int)sc.nodes.size(); d.nodes = sc.nodes.data();
        d.n_lights = (int)sc.lights.size(); d.lights = sc.lights.data();
        d.camera = cam;
        d.background[0] = bg.x; d.background[1] = bg.y; d.background[2] = bg.z;
        d.n_textures = (int)sc.textures.size(); d.textures = sc.textures.data();
        d.texel_bytes = (int64_t)sc.texels.size(); d.texels = sc.texels.data();
        d.background_texture = bg_tex;
        d.perlin_ranvec = sc.perlin_ranvec.data(); d.perlin_perm = sc.perlin_perm.data();
    }
};

std::string read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw pth::JsonError("cannot open '" + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
void copy_str(char *dst, size_t cap, const std::string &s)
{
    if (s.size() + 1 > cap) throw pth::JsonError("path too long: " + s);
    memcpy(dst, s.c_str(), s.size() + 1);
}

void parse_config(const Json &j, pth_config *c)
{   // Config(json) config.h:98-131 and s_film(json) config.h:19-27
    memset(c, 0, sizeof *c);
    const Json &film = j["film"];
    if (!film.is_object()) throw pth::JsonError("config: \"film\" object is required");
    c->width = film.value_int("width", 400);
    c->height = film.value_int("height", 300);
    c->exposure = (float)film.value("gamma", 2.2);    // sic: config.h:24
    c->gamma = (float)film.value("exposure", 0.0);    // sic: config.h:25
    copy_str(c->ppm_output_path, sizeof c->ppm_output_path, j.value_str("ppm_output_path", "out.ppm"));
    copy_str(c->png_output_path, sizeof c->png_output_path, j.value_str("png_output_path", "out.png"));
    copy_str(c->traced_paths_output_path, sizeof c->traced_paths_output_path, j["traced_paths_output_path"].as_string());
    copy_str(c->traced_paths_2d_output_path, sizeof c->traced_paths_2d_output_path, j["traced_paths_2d_output_path"].as_string());
    copy_str(c->scene_path, sizeof c->scene_path, j.value_str("scene", "scenes/scene.json"));
    c->should_trace_paths = j.value_bool("should_trace_paths", false);
    c->avg_number_of_paths = (float)j.value("avg_number_of_paths", 100.0);
    c->only_direct_illumination = j.value_bool("only_direct_illumination", false);
    c->block_width = j.value_int("block_width", 64);
    c->block_height = j.value_int("block_height", 64);
    static const std::map<std::string, int> render_types = {{"naive", 0}, {"progressive", 1}, {"tiled", 2}, {"hip_wavefront", 3}};
    static const std::map<std::string, int> integrators = {
        {"recursive path tracing", 0}, {"iterative path tracing", 1}, {"branched path tracing", 2},
        {"recursive nee path tracing", 3}, {"iterative nee path tracing", 4}, {"bidirectional path tracing", 5},
        {"stochastic progressive photon mapping", 6}, {"vertex connection merging", 7}, {"metropolis light transport", 8}};
    auto look = [](const std::map<std::string, int> &m, const std::string &k) { auto it = m.find(k); return it == m.end() ? 0 : it->second; };
    c->render_type = look(render_types, j.value_str("render_type", "progressive"));
    c->integrator_type = look(integrators, j.value_str("integrator_type", "recursive path tracing"));
    c->max_bounces = j.value_int("max_bounces", 10);
    c->samples = j.value_int("samples", 20);
    c->threads = (uint16_t)j.value_int("threads", 1);
    c->normal_offset = (float)j.value("normal_offset", 0.0001);
    c->light_samples = j.value_int("light_samples", 1);
    c->russian_roulette = j.value_bool("russian_roulette", true);
    long min_camera_rays = (long)c->samples * ((long)c->width * c->height);
    c->trace_probability = c->should_trace_paths ? (float)(j.value("avg_number_of_paths", 100.0) / min_camera_rays) : 0.0f;
}

template <typename F>
int guarded(F f)
{
    try {
        f();
        return 0;
    } catch (const std::exception &e) {
        pth_set_error(e.what());
        return -1;
    }
}

}  // namespace

extern "C" int pth_config_from_json(const char *text, pth_config *out)
{
    if (!text || !out) { pth_set_error("pth_config_from_json: null argument"); return -1; }
    return guarded([&] { parse_config(Json::parse(text), out); });
}
extern "C" int pth_config_from_file(const char *path, pth_config *out)
{
    if (!path || !out) { pth_set_error("pth_config_from_file: null argument"); return -1; }
    return guarded([&] { parse_config(Json::parse(read_file(path)), out); });
}
static thread_local std::string g_scene_base_dir = ".";
extern "C" pth_scene *pth_scene_from_json(const char *text, int32_t width, int32_t height)
{
    if (!text || width < 1 || height < 1) { pth_set_error("pth_scene_from_json: bad argument"); return nullptr; }
    pth_scene *s = new pth_scene();
    if (guarded([&] { SceneBuilder b(*s); b.base_dir = g_scene_base_dir; b.build(Json::parse(text), width, height); })) { delete s; return nullptr; }
    return s;
}
extern "C" pth_scene *pth_scene_from_file(const char *path, int32_t width, int32_t height)
{
    if (!path) { pth_set_error("pth_scene_from_file: null path"); return nullptr; }
    std::string text;
    if (guarded([&] { text = read_file(path); })) return nullptr;
    // the reference resolves a texture's "path" against its working directory, from which scenes are "scenes/x.json"
    // and images "assets/y.png": relative image paths are taken from the parent of the scene file's directory
    std::string p(path);
    size_t cut = p.find_last_of('/');
    std::string dir = cut == std::string::npos ? "." : p.substr(0, cut);
    g_scene_base_dir = dir + "/..";
    pth_scene *s = pth_scene_from_json(text.c_str(), width, height);
    g_scene_base_dir = ".";
    return s;
}
extern "C" const pt_scene_desc *pth_scene_desc(const pth_scene *s) { return s ? &s->desc : nullptr; }
extern "C" void pth_scene_free(pth_scene *s) { delete s; }

extern "C" int pth_spiral_tiles(int32_t width, int32_t height, int32_t bw, int32_t bh, int32_t *rects, int32_t max_tiles)
{   // NaiveSpiral queue.h:68-127
    if (width < 1 || height < 1 || bw < 1 || bh < 1) { pth_set_error("pth_spiral_tiles: bad argument"); return -1; }
    int tw = (int)std::ceil((float)width / bw), th = (int)std::ceil((float)height / bh);
    int radius = 1, n = 0;
    int x = tw % 2 == 0 ? (tw / 2 - 1) : (tw / 2);
    int y = th % 2 == 0 ? (th / 2 - 1) : (th / 2);
    int furthest = std::max(tw, th);
    int dx = 1, dy = 0, count = radius;
    while (radius <= furthest) {
        if (x >= 0 && y >= 0 && x < tw && y < th) {
            if (rects && n < max_tiles) {
                rects[4 * n + 0] = x * bw; rects[4 * n + 1] = y * bh;
                rects[4 * n + 2] = std::min(x * bw + bw, width); rects[4 * n + 3] = std::min(y * bh + bh, height);
            }
            n++;
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
    return n;
}

// ---- film output: calculate_luminance helpers.h:146-168, tonemap_uncharted tonemap.h:4-24, to_srgb helpers.h:78-93,
// output_to_file renderer.h:24-55
namespace {
const float tA = 0.15f, tB = 0.50f, tC = 0.10f, tD = 0.20f, tE = 0.02f, tF = 0.30f;
inline float uncharted(float x) { return ((x * (tA * x + tC * tB) + tD * tE) / (x * (tA * x + tB) + tD * tF)) - tE / tF; }
inline float clamp01(float x) { return x > 1.0f ? 1.0f : (x < 0.0f ? 0.0f : x); }
inline float to_srgb(float c)
{
    if (c < 0.0031308) return 323 * c / 25;
    return (float)((211 * powf(c, (float)(5.0 / 12)) - 11) / 200);
}
inline float de_nan1(float x) { return x == x ? x : 0.0f; }
}  // namespace

extern "C" int pth_write_ppm(const char *path, const float *fb, int32_t width, int32_t height, int32_t samples, float exposure_field)
{
    if (!path || !fb || width < 1 || height < 1 || samples < 1) { pth_set_error("pth_write_ppm: bad argument"); return -1; }
    float max_lum = -FLT_MAX;
    const float k = (float)(1.0 / float(samples));   // vec3::operator/=(float): k = 1.0 / t
    for (int j = height - 1; j >= 0; j--)
        for (int i = 0; i < width; i++) {
            const float *p = fb + ((size_t)j * width + i) * 3;
            float r = de_nan1(p[0]) * k, g = de_nan1(p[1]) * k, b = de_nan1(p[2]) * k;
            float f = std::sqrt(r * r + g * g + b * b);
            if (f > max_lum) max_lum = f;
        }
    FILE *f = fopen(path, "wb");
    if (!f) { pth_set_error(std::string("pth_write_ppm: cannot open ") + path); return -1; }
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    const float white = uncharted(max_lum);
    const float gain = 16 + exposure_field;   // col *= 16 + exposure (renderer.h:37)
    std::vector<unsigned char> row((size_t)width * 3);
    for (int j = height - 1; j >= 0; j--) {
        for (int i = 0; i < width; i++) {
            const float *p = fb + ((size_t)j * width + i) * 3;
            for (int ch = 0; ch < 3; ch++) {
                float c = p[ch] * k;
                c = c * gain;
                c = clamp01(uncharted(c) / white);
                c = 255 * to_srgb(c);
                row[3 * i + ch] = (unsigned char)int(c);
            }
        }
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return 0;
}

// ---- the Renderer plugin surface (renderer.h:114-150), mirrored for the HIP implementation ------------------
namespace pth {

class Renderer {   // same protocol as the reference's abstract Renderer
public:
    virtual ~Renderer() {}
    virtual void preprocess() = 0;
    virtual void start_render(std::chrono::high_resolution_clock::time_point) = 0;
    virtual void sync_progress() = 0;
    virtual bool is_done() = 0;
    virtual void finalize() = 0;
    std::vector<float> framebuffer;   // vec3 framebuffer[j][i] flattened, row 0 = bottom, SUM of samples
    bool completed = false;
    pth_config config{};
};

// render_type "hip_wavefront": start_render enqueues the whole image on the GPU(s) and returns; sync_progress polls.
// Which devices: config.json's own worker count, `threads` (config.h:117; Tiled::start_render spawns that many workers over
// one tile queue, renderer.h:553-603) -- threads > 1 = that many GPUs, capped by the devices present, through pt_multi (tiles
// of block_width x block_height in NaiveSpiral order, cost-balanced ownership, one exchange of the owned tiles at the end);
// threads <= 1, or one device = ONE context on the current device.  PATHTRACE_HIP_DEVICES="0,1,.." overrides the choice
// ("all" = every visible device; an ordinal may repeat -- "0,0": two contexts on one GPU, the rehearsal of the multi-GPU path
// on a one-GPU box).  The constructor does what is set-up in the reference too (Renderer::Renderer allocates the
// framebuffer, renderer.h:121-133): it sizes the wavefront streams for the job by the library's launch plan (pt_reserve) and
// waits for the per-scene build of the traversal kernels, so that start_render -> finalize times rendering alone.
class HipWavefront : public Renderer {
public:
    // the warm-up a fresh process needs before its one render runs at the rate of a warm one (pt_prime; set-up, like the
    // reference's thread spawn outside its timed region): 50 ms of one-sample passes
    static constexpr int kPrimeMs = 50;
    HipWavefront(const pth_config &cfg, const pt_scene_desc *scene)
    {
        config = cfg;
        pt_config pc{};
        pc.width = cfg.width; pc.height = cfg.height; pc.max_bounces = cfg.max_bounces; pc.light_samples = cfg.light_samples;
        pc.russian_roulette = cfg.russian_roulette; pc.only_direct_illumination = cfg.only_direct_illumination;
        pc.normal_offset = cfg.normal_offset; pc.seed = 0; pc.device = -1;
        pc.max_paths_in_flight = 0;   // the library's launch plan sizes the context (include/pathtrace_hip.h, ABI v6)
        std::vector<int32_t> devices;
        const char *e = getenv("PATHTRACE_HIP_DEVICES");
        if (e && !strcmp(e, "all")) {
            for (int d = 0; d < pt_device_count(); d++) devices.push_back(d);
        } else if (e) {
            for (const char *p = e; *p;) {
                char *end = nullptr;
                long v = strtol(p, &end, 10);
                if (end == p) break;
                devices.push_back((int32_t)v);
                p = (*end == ',') ? end + 1 : end;
            }
        } else if (cfg.threads > 1) {
            const int n = std::min<int>((int)cfg.threads, pt_device_count());
            for (int d = 0; d < n && n > 1; d++) devices.push_back(d);
        }
        if (devices.size() > 1) {
            multi = pt_multi_create(scene, &pc, (int32_t)devices.size(), devices.data(), std::max(cfg.block_width, 1), std::max(cfg.block_height, 1));
            if (!multi) throw JsonError(std::string("pt_multi_create: ") + pt_last_error());
            if (pt_multi_reserve(multi, std::max(cfg.samples, 1), kPrimeMs)) throw JsonError(std::string("pt_multi_reserve: ") + pt_last_error());
        } else {
            if (devices.size() == 1) pc.device = devices[0];
            ctx = pt_create(scene, &pc);
            if (!ctx) throw JsonError(std::string("pt_create: ") + pt_last_error());
            if (pt_reserve(ctx, (int64_t)cfg.width * cfg.height, std::max(cfg.samples, 1))) throw JsonError(std::string("pt_reserve: ") + pt_last_error());
            (void)pt_spec_wait(ctx);   // -1: no per-scene build for this scene / machine, the generic kernels render
            if (pt_prime(ctx, 0, nullptr, kPrimeMs)) throw JsonError(std::string("pt_prime: ") + pt_last_error());
        }
        framebuffer.assign((size_t)cfg.width * cfg.height * 3, 0.0f);
    }
    ~HipWavefront() override { pt_destroy(ctx); pt_multi_destroy(multi); }
    void preprocess() override {}
    void start_render(std::chrono::high_resolution_clock::time_point program_start) override
    {
        render_start = std::chrono::high_resolution_clock::now();
        (void)program_start;
        const int rc = multi ? pt_multi_render_async(multi, 0, config.samples) : pt_render_async(ctx, 0, 0, config.width, config.height, 0, config.samples);
        if (rc) throw JsonError(pt_last_error());
    }
    void sync_progress() override
    {
        uint64_t samples = 0, rays = 0;
        int r = multi ? pt_multi_poll(multi, &samples, &rays) : pt_poll(ctx, &samples, &rays);
        if (r < 0) throw JsonError(pt_last_error());
        long total = (long)config.samples * config.width * config.height;
        double dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - render_start).count();
        printf("samples left %20ld rate %10.0f\r", total - (long)samples, dt > 0 ? samples / dt : 0.0);
        fflush(stdout);
        // preview PPM from the live framebuffer, divisor 1 + done / (W*H) (renderer.h:617-618); like the reference's
        // main loop (one sync_progress per 0.5 s, main.cpp:158-163) at most two previews per second
        if (r == 0 && dt - last_preview >= 0.5) {
            uint64_t acc = 0;
            if (multi ? pt_multi_snapshot_framebuffer(multi, framebuffer.data(), &acc) : pt_snapshot_framebuffer(ctx, framebuffer.data(), &acc))
                throw JsonError(pt_last_error());
            const long npix = (long)config.width * config.height;
            if (pth_write_ppm(config.ppm_output_path, framebuffer.data(), config.width, config.height, (int32_t)(1 + (long)acc / npix), config.exposure))
                throw JsonError(pt_last_error());
            last_preview = dt;
            previews++;
        }
        completed = r == 1;
    }
    bool is_done() override { return completed; }
    // the sleep of main.cpp:162, ending early when the device has finished
    void idle(int ms) { if (multi) (void)pt_multi_wait_for(multi, ms); else (void)pt_wait_for(ctx, ms); }
    void finalize() override
    {
        if (multi ? pt_multi_wait(multi) : pt_wait(ctx)) throw JsonError(pt_last_error());
        // "time taken to compute" (renderer.h:698-700) is finalize's clock minus start_render's in the reference, i.e. it
        // includes up to one sleep of the caller's polling loop; the library stamps the moment the device finished
        // (pt_render_seconds), which is what this prints -- the wall figure of this process stands beside it
        const double wall = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - render_start).count();
        double dt = multi ? pt_multi_render_seconds(multi) : pt_render_seconds(ctx);
        if (!(dt > 0)) dt = wall;
        pt_counters c{};
        if (multi ? pt_multi_get_counters(multi, &c) : pt_get_counters(ctx, &c)) throw JsonError(pt_last_error());
        if (multi ? pt_multi_read_framebuffer(multi, framebuffer.data()) : pt_read_framebuffer(ctx, framebuffer.data())) throw JsonError(pt_last_error());
        printf("\ntime taken to compute %g\n", dt);
        printf("(start_render to finalize on this process's clock, polling included: %g)\n", wall);
        if (multi) {
            printf("rendered on %d devices\n", pt_multi_device_count(multi));
            for (int d = 0; d < pt_multi_device_count(multi); d++) {   // the reference prints one bounce count per worker thread
                pt_counters dc{};
                if (pt_multi_get_device_counters(multi, d, &dc)) throw JsonError(pt_last_error());
                printf("  device slot %d: %llu camera rays, %llu rays (%llu traced)\n", d, (unsigned long long)dc.camera_samples,
                       (unsigned long long)dc.rays, (unsigned long long)dc.rays_traced);
            }
        }
        printf("computed %llu camera rays in %gs, at %g rays per second\n", (unsigned long long)c.camera_samples, dt, c.camera_samples / dt);
        printf("computed %llu rays, at %g rays per second\n", (unsigned long long)c.rays, c.rays / dt);
        printf("traced %llu rays, at %g rays per second\n", (unsigned long long)c.rays_traced, c.rays_traced / dt);
        if (ctx) {
            pt_plan pl{};
            if (!pt_get_plan(ctx, &pl))
                printf("launch plan: %d batches of %d spp (%lld paths each) on %d lanes, %lld path slots, %.1f GB of streams%s\n", pl.batches, pl.spp_per_batch,
                       (long long)pl.paths_per_batch, pl.lanes, (long long)pl.path_slots, pl.stream_bytes / 1e9, pl.auto_sized ? " (sized by the library)" : "");
            // who compiled the traversal kernels this render ran (pt_spec_info): a foreign compiler is correct and slower
            char info[1024];
            const int n = pt_spec_info(ctx, info, sizeof info);
            if (n > 0 && n < (int)sizeof info) printf("per-scene kernels: %s\n", info);
        }
        if (pth_write_ppm(config.ppm_output_path, framebuffer.data(), config.width, config.height, config.samples, config.exposure))
            throw JsonError(pt_last_error());
    }
    pt_ctx *ctx = nullptr;
    pt_multi *multi = nullptr;
    std::chrono::high_resolution_clock::time_point render_start;
    double last_preview = 0.0;
    int previews = 0;
};

}  // namespace pth

extern "C" int pth_main(const char *workdir)
{   // main.cpp:108-168
    return guarded([&] {
        std::string wd = workdir ? workdir : ".";
        pth_config cfg;
        parse_config(Json::parse(read_file(wd + "/config.json")), &cfg);
        if (cfg.integrator_type != 4)
            throw pth::JsonError("only integrator_type \"iterative nee path tracing\" is implemented on the HIP path");
        auto t1 = std::chrono::high_resolution_clock::now();
        std::string sp = cfg.scene_path[0] == '/' ? std::string(cfg.scene_path) : wd + "/" + cfg.scene_path;
        pth_scene *scene = pth_scene_from_file(sp.c_str(), cfg.width, cfg.height);
        if (!scene) throw pth::JsonError(pt_last_error());
        auto t2 = std::chrono::high_resolution_clock::now();
        printf("time taken to build bvh %g\n", std::chrono::duration<double>(t2 - t1).count());
        std::string ppm = cfg.ppm_output_path[0] == '/' ? std::string(cfg.ppm_output_path) : wd + "/" + cfg.ppm_output_path;
        copy_str(cfg.ppm_output_path, sizeof cfg.ppm_output_path, ppm);
        {
            pth::HipWavefront r(cfg, pth_scene_desc(scene));
            r.start_render(t2);
            while (!r.is_done()) {
                r.sync_progress();
                if (!r.is_done()) r.idle(50);   // main.cpp:162 sleeps 0.5 s; this wait ends when the device does
            }
            printf(" done\n");
            r.finalize();
        }
        pth_scene_free(scene);
    });
}
