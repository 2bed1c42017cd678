// Multi-GPU inside the library (include/pathtrace_hip.h pt_multi_*): one pt_ctx per device in ONE process,
// the film partitioned by image tile in NaiveSpiral order (queue.h:68-127), every device renders its tiles with no
// communication, and one exchange at the end sums the framebuffers into the first device.  Tile ownership is disjoint,
// so every pixel receives one non-zero term and the image is the single-device image bit for bit, for any device list.
//
// The reference's analogue is the thread fan-out of Tiled::start_render (renderer.h:553-603): config.threads workers
// pulling tiles from one spiral queue into one shared framebuffer.  Here the "workers" are GPUs, the queue is a static
// cost-balanced ownership map (a GPU batches all its tiles into one wavefront launch, so tiles cannot be pulled one at
// a time), and the shared framebuffer is reassembled by the final sum.
//
// Exchange (default): every peer packs the pixels of the tiles it OWNS into a contiguous buffer on its own device, the
// packed buffers travel to the root concurrently -- one device-to-device copy per peer, each on a stream of its own, i.e.
// over that peer's own xGMI link -- and one add kernel per peer scatters them into the root's sum (ownership is disjoint,
// so the adds touch disjoint pixels and run concurrently).  Bytes moved = the pixels the peers own x 16: at 4K on 8
// devices 7/8 x 133 MB = 116 MB in all instead of 7 whole frames.  It also works when one device is listed twice.
// PATHTRACE_HIP_MULTI=rccl: one RCCL ncclReduce per device in a group instead (librccl.so is loaded on demand, the
// library does not link it).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/pathtrace_hip.h"

void pth_set_error(const std::string &m);
bool pt_clock_end(pt_ctx *c, std::chrono::steady_clock::time_point *t_end);   // pt_context.cpp
namespace ptd {
void launch_add_fb(void *dst_rgba, const void *src_rgba, long long n_pixels, hipStream_t s);
void launch_pack_tiles(void *packed, const void *fb, const void *rects, const void *pix0, int n, int width, int total, hipStream_t s);
void launch_unpack_add_tiles(void *fb, const void *packed, const void *rects, const void *pix0, int n, int width, int total, hipStream_t s);
}

namespace {
void merr(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    pth_set_error(buf);
}
#define MHIP(x)                                                                                  \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) { merr("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } \
    } while (0)

// ---- RCCL through dlopen (the few entry points of the single-process form) ----
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Reduce;
    }
};
enum { kNcclFloat = 7, kNcclSum = 0 };
}  // namespace

struct pt_multi {
    int w = 0, h = 0;
    std::vector<int> dev;
    std::vector<pt_ctx *> ctx;
    std::vector<std::vector<int32_t>> rects;   // per context: its tiles as x0 y0 x1 y1
    std::vector<int32_t> owner;                // per spiral tile
    void *reduced = nullptr;                   // on dev[0]: the summed RGBA framebuffer
    hipStream_t stream = nullptr;              // on dev[0]
    // owned-tile exchange, per context i > 0: its rect list and packed offsets on its own device and on the root, the
    // packed pixels on its device, their landing area on the root, one stream on each side, one event
    struct Peer {
        int n = 0, total = 0;
        void *rects_src = nullptr, *pix0_src = nullptr, *packed_src = nullptr;
        void *rects_dst = nullptr, *pix0_dst = nullptr, *packed_dst = nullptr;
        hipStream_t s_src = nullptr, s_dst = nullptr;
        hipEvent_t packed_ev = nullptr;
    };
    std::vector<Peer> peers;
    uint64_t exchange_bytes = 0;               // moved by the last reduce
    bool use_rccl = false;
    Rccl rccl;
    std::vector<void *> comms;
    std::vector<hipStream_t> rstreams;         // one per device for the RCCL group
    std::vector<float> host_tmp;
    std::chrono::steady_clock::time_point t_render;   // entry of the last pt_multi_render_async
    bool rendered = false;
};

// longest-processing-time greedy (costliest tile first, ties by spiral index, to the least loaded owner, ties to the
// lowest index): the map of pathtrace_amd/distributed.py balanced_owners, integer arithmetic only
static std::vector<int32_t> balanced_owners(const std::vector<uint64_t> &cost, int n)
{
    std::vector<int> order(cost.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    std::vector<uint64_t> load(n, 0);
    std::vector<int32_t> own(cost.size(), 0);
    for (int k : order) {
        int best = 0;
        for (int r = 1; r < n; r++)
            if (load[r] < load[best]) best = r;
        own[k] = best;
        load[best] += cost[k];
    }
    return own;
}

extern "C" void pt_multi_destroy(pt_multi *m)
{
    if (!m) return;
    for (size_t i = 0; i < m->comms.size(); i++)
        if (m->comms[i]) m->rccl.CommDestroy(m->comms[i]);
    for (size_t i = 0; i < m->rstreams.size(); i++)
        if (m->rstreams[i]) { (void)hipSetDevice(m->dev[i]); (void)hipStreamDestroy(m->rstreams[i]); }
    for (pt_ctx *c : m->ctx) pt_destroy(c);
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    for (size_t i = 0; i < m->peers.size(); i++) {
        pt_multi::Peer &p = m->peers[i];
        if (i < m->dev.size()) (void)hipSetDevice(m->dev[i]);
        if (p.rects_src) (void)hipFree(p.rects_src);
        if (p.pix0_src) (void)hipFree(p.pix0_src);
        if (p.packed_src) (void)hipFree(p.packed_src);
        if (p.s_src) (void)hipStreamDestroy(p.s_src);
        if (p.packed_ev) (void)hipEventDestroy(p.packed_ev);
        if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
        if (p.rects_dst) (void)hipFree(p.rects_dst);
        if (p.pix0_dst) (void)hipFree(p.pix0_dst);
        if (p.packed_dst) (void)hipFree(p.packed_dst);
        if (p.s_dst) (void)hipStreamDestroy(p.s_dst);
    }
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    if (m->reduced) (void)hipFree(m->reduced);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

extern "C" pt_multi *pt_multi_create(const pt_scene_desc *scene, const pt_config *config, int32_t n_devices, const int32_t *devices,
                                     int32_t block_w, int32_t block_h)
{
    if (!scene || !config || n_devices < 1 || !devices || block_w < 1 || block_h < 1) { merr("pt_multi_create: bad argument"); return nullptr; }
    pt_multi *m = new pt_multi();
    auto fail = [&]() { std::string e = pt_last_error(); pt_multi_destroy(m); pth_set_error(e); return (pt_multi *)nullptr; };
    m->w = config->width; m->h = config->height;
    for (int i = 0; i < n_devices; i++) {
        pt_config pc = *config;
        pc.device = devices[i];
        pt_ctx *c = pt_create(scene, &pc);
        if (!c) return fail();
        m->ctx.push_back(c);
        m->dev.push_back(devices[i]);
    }
    // tiles in spiral order and their cost: World::hit queries of one sample per pixel plus one unit per camera sample,
    // counted on the first device (deterministic: the RNG is keyed by pixel and sample)
    const int n_tiles = pth_spiral_tiles(m->w, m->h, block_w, block_h, nullptr, 0);
    std::vector<int32_t> tiles((size_t)n_tiles * 4);
    pth_spiral_tiles(m->w, m->h, block_w, block_h, tiles.data(), n_tiles);
    const char *menv = getenv("PATHTRACE_HIP_MULTI");   // comma list: roundrobin (tile k -> device k mod n), rccl (ncclReduce of whole frames)
    if (n_devices == 1 || (menv && strstr(menv, "roundrobin"))) {
        m->owner.resize(n_tiles);
        for (int k = 0; k < n_tiles; k++) m->owner[k] = k % n_devices;
    } else {
        // one pass over the frame on the first device (pt_measure_tile_costs) instead of one render + counter read per tile
        std::vector<uint64_t> cost(n_tiles);
        if (pt_measure_tile_costs(m->ctx[0], n_tiles, tiles.data(), 1, cost.data())) return fail();
        for (int k = 0; k < n_tiles; k++)
            cost[k] += (uint64_t)(tiles[4 * k + 2] - tiles[4 * k]) * (uint64_t)(tiles[4 * k + 3] - tiles[4 * k + 1]);   // + one unit per camera sample
        m->owner = balanced_owners(cost, n_devices);
    }
    m->rects.resize(n_devices);
    for (int k = 0; k < n_tiles; k++)
        m->rects[m->owner[k]].insert(m->rects[m->owner[k]].end(), tiles.begin() + 4 * k, tiles.begin() + 4 * k + 4);
    const size_t bytes = (size_t)m->w * m->h * 16;
    if (hipSetDevice(m->dev[0]) != hipSuccess || hipMalloc(&m->reduced, bytes) != hipSuccess ||
        hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
        merr("pt_multi_create: allocating the reduce buffer on device %d failed", m->dev[0]);
        return fail();
    }
    // peers: let the root reach their memory directly where the topology allows it (hipMemcpyPeerAsync works either way,
    // staged through the host when it does not).  A failure other than "already enabled" is reported, not swallowed.
    for (int i = 1; i < n_devices; i++) {
        if (m->dev[i] == m->dev[0]) continue;
        int can = 0;
        hipError_t e = hipDeviceCanAccessPeer(&can, m->dev[0], m->dev[i]);
        if (e == hipSuccess && can) {
            e = hipDeviceEnablePeerAccess(m->dev[i], 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
        }
        if (e != hipSuccess) {
            merr("pt_multi_create: peer access from device %d to device %d: %s", m->dev[0], m->dev[i], hipGetErrorString(e));
            return fail();
        }
    }
    // the exchange's rect tables and packed buffers (see the head of this file)
    m->peers.resize(n_devices);
    for (int i = 1; i < n_devices; i++) {
        pt_multi::Peer &p = m->peers[i];
        p.n = (int)(m->rects[i].size() / 4);
        if (!p.n) continue;
        std::vector<int32_t> pix0(p.n + 1, 0);
        for (int k = 0; k < p.n; k++) {
            const int32_t *r = &m->rects[i][4 * (size_t)k];
            pix0[k + 1] = pix0[k] + (r[2] - r[0]) * (r[3] - r[1]);
        }
        p.total = pix0[p.n];
        const size_t rb = (size_t)p.n * 16, pb = (size_t)(p.n + 1) * 4, kb = (size_t)p.total * 16;
        bool ok = hipSetDevice(m->dev[i]) == hipSuccess && hipMalloc(&p.rects_src, rb) == hipSuccess && hipMalloc(&p.pix0_src, pb) == hipSuccess &&
                  hipMalloc(&p.packed_src, kb) == hipSuccess && hipMemcpy(p.rects_src, m->rects[i].data(), rb, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(p.pix0_src, pix0.data(), pb, hipMemcpyHostToDevice) == hipSuccess &&
                  hipStreamCreateWithFlags(&p.s_src, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&p.packed_ev, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipSetDevice(m->dev[0]) == hipSuccess && hipMalloc(&p.rects_dst, rb) == hipSuccess && hipMalloc(&p.pix0_dst, pb) == hipSuccess &&
             hipMalloc(&p.packed_dst, kb) == hipSuccess && hipMemcpy(p.rects_dst, m->rects[i].data(), rb, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(p.pix0_dst, pix0.data(), pb, hipMemcpyHostToDevice) == hipSuccess &&
             hipStreamCreateWithFlags(&p.s_dst, hipStreamNonBlocking) == hipSuccess;
        if (!ok) { merr("pt_multi_create: exchange buffers for device %d: %s", m->dev[i], hipGetErrorString(hipGetLastError())); return fail(); }
    }
    if (hipSetDevice(m->dev[0]) != hipSuccess) { merr("pt_multi_create: hipSetDevice(%d)", m->dev[0]); return fail(); }
    if (menv && strstr(menv, "rccl")) {
        bool distinct = true;
        for (int i = 0; i < n_devices; i++)
            for (int j = 0; j < i; j++) distinct = distinct && m->dev[i] != m->dev[j];
        if (!distinct) { merr("pt_multi_create: PATHTRACE_HIP_MULTI=rccl needs distinct devices"); return fail(); }
        if (!m->rccl.load()) { merr("pt_multi_create: librccl.so could not be loaded"); return fail(); }
        m->comms.assign(n_devices, nullptr);
        const int rc = m->rccl.CommInitAll(m->comms.data(), n_devices, m->dev.data());
        if (rc != 0) { merr("pt_multi_create: ncclCommInitAll: %s", m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "error"); return fail(); }
        m->rstreams.assign(n_devices, nullptr);
        for (int i = 0; i < n_devices; i++)
            if (hipSetDevice(m->dev[i]) != hipSuccess || hipStreamCreateWithFlags(&m->rstreams[i], hipStreamNonBlocking) != hipSuccess) {
                merr("pt_multi_create: stream on device %d", m->dev[i]);
                return fail();
            }
        m->use_rccl = true;
    }
    return m;
}

// Every device is driven by a host thread of its own -- the counterpart of Tiled::start_render's worker threads
// (renderer.h:553-603), with the difference that a worker here only ENQUEUES: sizing its context's streams (the launch plan,
// pt_context.cpp) and a few hundred kernel launches per device, which one thread would otherwise issue device after device
// (round 4: ~7 700 launches per device from one thread with the old 8 Mi-path default).  The call returns when every device
// has its work queued; nothing waits for the devices themselves.
template <typename F>
static int on_every_device(pt_multi *m, const char *what, F f)
{
    const size_t n = m->ctx.size();
    std::vector<std::string> err(n);
    std::vector<int> rc(n, 0);
    auto work = [&](size_t i) {
        rc[i] = f(i);
        if (rc[i]) err[i] = pt_last_error();   // the message is thread-local: carried over to the caller's thread below
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < n; i++) th.emplace_back(work, i);
    work(0);
    for (std::thread &t : th) t.join();
    for (size_t i = 0; i < n; i++)
        if (rc[i]) { merr("%s: device slot %zu: %s", what, i, err[i].c_str()); return -1; }
    return 0;
}
static int64_t owned_pixels(const pt_multi *m, size_t i)
{
    int64_t px = 0;
    for (size_t k = 0; k + 3 < m->rects[i].size(); k += 4) px += (int64_t)(m->rects[i][k + 2] - m->rects[i][k]) * (m->rects[i][k + 3] - m->rects[i][k + 1]);
    return px;
}

extern "C" int pt_multi_render_async(pt_multi *m, int32_t spp_begin, int32_t spp_end)
{
    if (!m) { merr("pt_multi_render_async: null"); return -1; }
    m->t_render = std::chrono::steady_clock::now();
    m->rendered = true;
    return on_every_device(m, "pt_multi_render_async", [&](size_t i) {
        const int n = (int)(m->rects[i].size() / 4);
        return n ? pt_render_tiles_async(m->ctx[i], n, m->rects[i].data(), spp_begin, spp_end) : 0;
    });
}

extern "C" int pt_multi_reserve(pt_multi *m, int32_t samples, int32_t prime_ms)
{
    if (!m || samples < 1) { merr("pt_multi_reserve: bad argument"); return -1; }
    return on_every_device(m, "pt_multi_reserve", [&](size_t i) {
        const int64_t px = owned_pixels(m, i);
        if (!px) return 0;
        if (pt_reserve(m->ctx[i], px, samples)) return -1;
        (void)pt_spec_wait(m->ctx[i]);   // every device on its scene's own kernels before the first timed launch (-1: generic kernels)
        return prime_ms > 0 ? pt_prime(m->ctx[i], (int32_t)(m->rects[i].size() / 4), m->rects[i].data(), prime_ms) : 0;
    });
}

extern "C" double pt_multi_render_seconds(pt_multi *m)
{
    if (!m || !m->rendered) return -1.0;
    std::chrono::steady_clock::time_point last = m->t_render;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        if (m->rects[i].empty()) continue;
        std::chrono::steady_clock::time_point t;
        if (!pt_clock_end(m->ctx[i], &t)) return -1.0;
        last = std::max(last, t);
    }
    return std::chrono::duration<double>(last - m->t_render).count();
}

extern "C" int pt_multi_wait_for(pt_multi *m, int32_t timeout_ms)
{
    if (!m) { merr("pt_multi_wait_for: null"); return -1; }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(std::max(timeout_ms, 0));
    for (size_t i = 0; i < m->ctx.size(); i++) {
        if (m->rects[i].empty()) continue;
        const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
        const int r = pt_wait_for(m->ctx[i], (int32_t)std::max<long long>(left, 0));
        if (r <= 0) return r;
    }
    return 1;
}

extern "C" int pt_multi_poll(pt_multi *m, uint64_t *samples_done, uint64_t *rays_done)
{
    if (!m) { merr("pt_multi_poll: null"); return -1; }
    uint64_t s = 0, r = 0;
    int done = 1;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        if (m->rects[i].empty()) continue;
        uint64_t si = 0, ri = 0;
        const int d = pt_poll(m->ctx[i], &si, &ri);
        if (d < 0) return -1;
        done = done && d;
        s += si; r += ri;
    }
    if (samples_done) *samples_done = s;
    if (rays_done) *rays_done = r;
    return done;
}

extern "C" int pt_multi_wait(pt_multi *m)
{
    if (!m) { merr("pt_multi_wait: null"); return -1; }
    for (pt_ctx *c : m->ctx)
        if (pt_wait(c)) return -1;
    return 0;
}

// sum of the per-device framebuffers into m->reduced on the first device; the per-device buffers stay as they are, so a
// render can go on after a read
static int reduce_now(pt_multi *m)
{
    if (pt_multi_wait(m)) return -1;
    const size_t npix = (size_t)m->w * m->h, bytes = npix * 16;
    if (m->use_rccl) {
        if (m->rccl.GroupStart() != 0) { merr("ncclGroupStart failed"); return -1; }
        for (size_t i = 0; i < m->ctx.size(); i++) {
            const int rc = m->rccl.Reduce(pt_device_framebuffer(m->ctx[i]), i == 0 ? m->reduced : nullptr, npix * 4, kNcclFloat, kNcclSum, 0,
                                          m->comms[i], m->rstreams[i]);
            if (rc != 0) { merr("ncclReduce: %s", m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "error"); return -1; }
        }
        if (m->rccl.GroupEnd() != 0) { merr("ncclGroupEnd failed"); return -1; }
        for (size_t i = 0; i < m->ctx.size(); i++) { MHIP(hipSetDevice(m->dev[i])); MHIP(hipStreamSynchronize(m->rstreams[i])); }
        MHIP(hipSetDevice(m->dev[0]));
        return 0;
    }
    // owned tiles only: pack on every peer (its own device and stream) ...
    m->exchange_bytes = 0;
    for (size_t i = 1; i < m->ctx.size(); i++) {
        pt_multi::Peer &p = m->peers[i];
        if (!p.n) continue;
        MHIP(hipSetDevice(m->dev[i]));
        ptd::launch_pack_tiles(p.packed_src, pt_device_framebuffer(m->ctx[i]), p.rects_src, p.pix0_src, p.n, m->w, p.total, p.s_src);
        MHIP(hipGetLastError());
        MHIP(hipEventRecord(p.packed_ev, p.s_src));
    }
    // ... the root's own frame is the start of the sum, and every peer's packed pixels come over on a stream of their own
    // and are added where they belong (disjoint pixels: the adds of different peers do not meet)
    MHIP(hipSetDevice(m->dev[0]));
    MHIP(hipMemcpyAsync(m->reduced, pt_device_framebuffer(m->ctx[0]), bytes, hipMemcpyDeviceToDevice, m->stream));
    MHIP(hipStreamSynchronize(m->stream));
    for (size_t i = 1; i < m->ctx.size(); i++) {
        pt_multi::Peer &p = m->peers[i];
        if (!p.n) continue;
        const size_t kb = (size_t)p.total * 16;
        MHIP(hipStreamWaitEvent(p.s_dst, p.packed_ev, 0));
        if (m->dev[i] == m->dev[0]) MHIP(hipMemcpyAsync(p.packed_dst, p.packed_src, kb, hipMemcpyDeviceToDevice, p.s_dst));
        else MHIP(hipMemcpyPeerAsync(p.packed_dst, m->dev[0], p.packed_src, m->dev[i], kb, p.s_dst));
        ptd::launch_unpack_add_tiles(m->reduced, p.packed_dst, p.rects_dst, p.pix0_dst, p.n, m->w, p.total, p.s_dst);
        MHIP(hipGetLastError());
        m->exchange_bytes += kb;
    }
    for (size_t i = 1; i < m->ctx.size(); i++)
        if (m->peers[i].n) MHIP(hipStreamSynchronize(m->peers[i].s_dst));
    (void)npix;
    return 0;
}

extern "C" int pt_multi_read_framebuffer(pt_multi *m, float *rgb_sum)
{
    if (!m || !rgb_sum) { merr("pt_multi_read_framebuffer: null argument"); return -1; }
    if (reduce_now(m)) return -1;
    const size_t n = (size_t)m->w * m->h;
    m->host_tmp.resize(n * 4);
    MHIP(hipMemcpy(m->host_tmp.data(), m->reduced, n * 16, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) { rgb_sum[3 * i] = m->host_tmp[4 * i]; rgb_sum[3 * i + 1] = m->host_tmp[4 * i + 1]; rgb_sum[3 * i + 2] = m->host_tmp[4 * i + 2]; }
    return 0;
}

extern "C" int pt_multi_snapshot_framebuffer(pt_multi *m, float *rgb_sum, uint64_t *samples_accumulated)
{
    if (!m || !rgb_sum) { merr("pt_multi_snapshot_framebuffer: null argument"); return -1; }
    const size_t n = (size_t)m->w * m->h * 3;
    std::fill(rgb_sum, rgb_sum + n, 0.0f);
    std::vector<float> part(n);
    uint64_t acc = 0;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        if (m->rects[i].empty()) continue;
        uint64_t a = 0;
        if (pt_snapshot_framebuffer(m->ctx[i], part.data(), &a)) return -1;
        acc += a;
        for (size_t k = 0; k < n; k++) rgb_sum[k] += part[k];   // disjoint tiles: one non-zero term per pixel
    }
    if (samples_accumulated) *samples_accumulated = acc;
    return 0;
}

extern "C" int pt_multi_get_counters(pt_multi *m, pt_counters *out)
{
    if (!m || !out) { merr("pt_multi_get_counters: null argument"); return -1; }
    pt_counters s{};
    for (pt_ctx *c : m->ctx) {
        pt_counters k{};
        if (pt_get_counters(c, &k)) return -1;
        uint64_t *d = (uint64_t *)&s;
        const uint64_t *q = (const uint64_t *)&k;
        for (size_t j = 0; j < sizeof(pt_counters) / 8; j++) d[j] += q[j];
    }
    *out = s;
    return 0;
}

extern "C" int pt_multi_clear(pt_multi *m)
{
    if (!m) { merr("pt_multi_clear: null"); return -1; }
    for (pt_ctx *c : m->ctx)
        if (pt_clear_framebuffer(c)) return -1;
    return 0;
}

extern "C" int pt_multi_device_count(pt_multi *m) { return m ? (int)m->ctx.size() : 0; }
extern "C" int pt_multi_get_device_counters(pt_multi *m, int32_t index, pt_counters *out)
{
    if (!m || !out || index < 0 || index >= (int)m->ctx.size()) { merr("pt_multi_get_device_counters: bad argument"); return -1; }
    return pt_get_counters(m->ctx[index], out);
}
extern "C" uint64_t pt_multi_exchange_bytes(pt_multi *m) { return m ? m->exchange_bytes : 0; }
extern "C" int pt_multi_tile_owners(pt_multi *m, int32_t *owners, int32_t max_tiles)
{
    if (!m) return 0;
    const int n = (int)m->owner.size();
    if (owners)
        for (int k = 0; k < n && k < max_tiles; k++) owners[k] = m->owner[k];
    return n;
}
