// gather.cpp -- the one exchange of the multi-GPU path: every rank's candidate records to every rank.
//
// The reference has no counterpart -- it is one process; the exchange stands where ERFilter::er_track reads the strong / weak
// lists of ALL planes (src/ER.cpp:63).  One process per GPU; frames (or planes, or strips of a plane) are dealt out to the
// ranks and only the 48-byte records of their NMS survivors travel:
//
//     all_gather(1 x u32: my count)  ->  all_gather(records padded to the largest count)  ->  drop the padding
//
// Two transports behind one interface:
//   * RCCL (librccl.so, loaded with dlopen the first time a communicator is made -- the library itself does not link it):
//     ncclAllGather on a side stream of the context; the records go from the device array the detect call left them in
//     (no host hop on the sending side) into one device buffer, which comes to the host in one copy.
//   * an in-process group ("local"): N communicators made from one str_er_comm_local_group() exchange through host memory
//     with a barrier -- what the CPU tests (no GPU, world size 2 and 3, one thread per rank) run the same packing code on.
#include "../../include/str_er.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

extern "C" {
// (str_er_api.cpp) the device array and count of the candidates of the context's last detect call, and its device
int str_er_internal_last_cands(str_er_ctx *ctx, const void **d_cands, uint32_t *n, int *device);
}

namespace {

struct LocalGroup {               // shared by the communicators of one in-process group
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<const void *> send;
    int arrived = 0, left = 0;
    uint64_t generation = 0;
};

// ---- RCCL through dlopen (no link-time dependency) ------------------------------------------------------------------
struct Id128 { char b[128]; };     // ncclUniqueId, passed by value
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy the process already has (PyTorch brings its own librccl.so) is the one to use: two RCCLs in one process do not mix
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (r.lib) break;
        }
        if (!r.lib)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.lib) break;
            }
        if (!r.lib) return;
        r.GetUniqueId = reinterpret_cast<int (*)(void *)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<int (*)(void **, int, Id128, int)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, hipStream_t)>(dlsym(r.lib, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<const char *(*)(int)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { dlclose(r.lib); r.lib = nullptr; }
    });
    return r.lib ? &r : nullptr;
}
constexpr int NCCL_CHAR = 0;      // ncclInt8 / ncclChar

} // namespace

struct str_er_comm {
    int rank = 0, world = 1;
    std::string err;
    // local transport
    std::shared_ptr<LocalGroup> group;
    // RCCL transport
    void *nccl = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t *d_count = nullptr, *d_counts = nullptr;      // 1 and `world` words
    uint8_t *d_send = nullptr, *d_recv = nullptr;
    size_t send_cap = 0, recv_cap = 0;                     // bytes
    std::vector<uint8_t> h_recv;
};

struct str_er_comm_group { std::shared_ptr<LocalGroup> g; };

namespace {

thread_local std::string g_comm_create_error;
int cfail(str_er_comm *c, int code, const std::string &msg) { if (c) c->err = msg; else g_comm_create_error = msg; return code; }

// all ranks contribute `bytes` each; recv gets world * bytes, rank-major.  Host memory, in-process.
int local_all_gather(str_er_comm *c, const void *send, void *recv, size_t bytes)
{
    LocalGroup &g = *c->group;
    std::unique_lock<std::mutex> lk(g.mu);
    g.cv.wait(lk, [&] { return g.left == 0; });          // the previous round has been read by everybody
    g.send[(size_t)c->rank] = send;
    const uint64_t gen = g.generation;
    if (++g.arrived == g.world) { g.left = g.world; g.arrived = 0; ++g.generation; g.cv.notify_all(); }
    else g.cv.wait(lk, [&] { return g.generation != gen; });
    for (int r = 0; r < g.world; ++r) std::memcpy(static_cast<uint8_t *>(recv) + (size_t)r * bytes, g.send[(size_t)r], bytes);
    // nobody returns before everybody has read: the send buffers belong to the callers
    const uint64_t done_gen = g.generation;
    if (--g.left == 0) g.cv.notify_all();
    else g.cv.wait(lk, [&] { return g.left == 0 || g.generation != done_gen; });
    return STR_ER_OK;
}

int ensure(str_er_comm *c, uint8_t *&p, size_t &cap, size_t need)
{
    if (need <= cap) return STR_ER_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    need = std::max<size_t>(need + need / 2, 1 << 16);
    if (hipMalloc(reinterpret_cast<void **>(&p), need) != hipSuccess) return cfail(c, STR_ER_ENOMEM, "hipMalloc (gather buffer)");
    cap = need;
    return STR_ER_OK;
}

// drop the padding: recv = world blocks of cap records, counts[r] valid in block r; frame offsets applied per rank
void compact(const uint8_t *recv, const uint32_t *counts, int world, size_t cap, const uint32_t *frame_offsets, str_er_cand *out)
{
    size_t at = 0;
    for (int r = 0; r < world; ++r) {
        const str_er_cand *blk = reinterpret_cast<const str_er_cand *>(recv + (size_t)r * cap * sizeof(str_er_cand));
        for (uint32_t i = 0; i < counts[r]; ++i) {
            out[at] = blk[i];
            if (frame_offsets) out[at].frame += frame_offsets[r];
            ++at;
        }
    }
}

} // namespace

extern "C" {

int str_er_comm_unique_id(void *id128)
{
    if (!id128) return STR_ER_EINVAL;
    Rccl *r = rccl();
    if (!r) return STR_ER_ESTATE;
    return r->GetUniqueId(id128) == 0 ? STR_ER_OK : STR_ER_EHIP;
}

int str_er_comm_create(int32_t device, int32_t rank, int32_t world, const void *id128, str_er_comm **out)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return STR_ER_EINVAL;
    *out = nullptr;
    Rccl *r = rccl();
    if (!r) return STR_ER_ESTATE;                       // librccl.so not found
    if (hipSetDevice(device) != hipSuccess) return STR_ER_EHIP;
    str_er_comm *c = new (std::nothrow) str_er_comm();
    if (!c) return STR_ER_ENOMEM;
    c->rank = rank; c->world = world; c->device = device;
    Id128 id;
    std::memcpy(id.b, id128, 128);
    (void)hipGetLastError();          // (RCCL checks the thread's last HIP error after its launches: a stale one from an earlier, handled failure would fail it)
    const int nrc = r->CommInitRank(&c->nccl, world, id, rank);
    if (nrc != 0) {
        cfail(nullptr, STR_ER_EHIP, std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(nrc) : "error") + " (" + std::to_string(nrc) + ")");
        c->nccl = nullptr;
        str_er_comm_destroy(c);
        return STR_ER_EHIP;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->d_count), 4) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&c->d_counts), 4 * (size_t)world) != hipSuccess) {
        cfail(nullptr, STR_ER_EHIP, "stream / buffer creation for the communicator failed");
        str_er_comm_destroy(c);
        return STR_ER_EHIP;
    }
    *out = c;
    return STR_ER_OK;
}

int str_er_comm_local_group(int32_t world, str_er_comm_group **out)
{
    if (!out || world < 1 || world > 1024) return STR_ER_EINVAL;
    str_er_comm_group *g = new (std::nothrow) str_er_comm_group();
    if (!g) return STR_ER_ENOMEM;
    g->g = std::make_shared<LocalGroup>();
    g->g->world = world;
    g->g->send.assign((size_t)world, nullptr);
    *out = g;
    return STR_ER_OK;
}

void str_er_comm_local_group_free(str_er_comm_group *g) { delete g; }

int str_er_comm_create_local(str_er_comm_group *g, int32_t rank, str_er_comm **out)
{
    if (!g || !out || rank < 0 || rank >= g->g->world) return STR_ER_EINVAL;
    str_er_comm *c = new (std::nothrow) str_er_comm();
    if (!c) return STR_ER_ENOMEM;
    c->rank = rank; c->world = g->g->world; c->group = g->g;
    *out = c;
    return STR_ER_OK;
}

void str_er_comm_destroy(str_er_comm *c)
{
    if (!c) return;
    if (c->nccl) {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (Rccl *r = rccl()) (void)r->CommDestroy(c->nccl);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (void *p : {(void *)c->d_count, (void *)c->d_counts, (void *)c->d_send, (void *)c->d_recv}) if (p) (void)hipFree(p);
    delete c;
}

const char *str_er_comm_last_error(const str_er_comm *c) { return c ? c->err.c_str() : g_comm_create_error.c_str(); }
int32_t str_er_comm_rank(const str_er_comm *c) { return c ? c->rank : -1; }
int32_t str_er_comm_world(const str_er_comm *c) { return c ? c->world : 0; }

// counts[world] (may be null); *all is malloc'ed (free with str_er_gather_free), ordered by rank.
static int gather_impl(str_er_comm *c, const str_er_cand *h_local, const void *d_local, uint32_t n_local, uint32_t frame_offset,
                       str_er_cand **all, int32_t *n_all, int32_t *counts_out)
{
    if (!c || !all || !n_all) return STR_ER_EINVAL;
    *all = nullptr; *n_all = 0;
    const int W = c->world;
    std::vector<uint32_t> counts((size_t)W), offs((size_t)W);
    uint32_t mine[2] = {n_local, frame_offset};
    std::vector<uint32_t> both(2 * (size_t)W);
    if (c->group) {
        const int rc = local_all_gather(c, mine, both.data(), sizeof(mine));
        if (rc != STR_ER_OK) return rc;
    } else {
        Rccl *r = rccl();
        if (hipSetDevice(c->device) != hipSuccess) return cfail(c, STR_ER_EHIP, "hipSetDevice");
        uint8_t *tmp = nullptr;               // 8 bytes per rank
        size_t need = 8 * (size_t)W + 8;
        if (ensure(c, c->d_send, c->send_cap, need) != STR_ER_OK || ensure(c, c->d_recv, c->recv_cap, need) != STR_ER_OK) return STR_ER_ENOMEM;
        tmp = c->d_send;
        if (hipMemcpyAsync(tmp, mine, 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) return cfail(c, STR_ER_EHIP, "count upload");
        if (r->AllGather(tmp, c->d_recv, 8, NCCL_CHAR, c->nccl, c->stream) != 0) return cfail(c, STR_ER_EHIP, "ncclAllGather (counts)");
        if (hipMemcpyAsync(both.data(), c->d_recv, 8 * (size_t)W, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            return cfail(c, STR_ER_EHIP, "count download");
    }
    size_t total = 0, cap = 1;
    for (int r = 0; r < W; ++r) {
        counts[(size_t)r] = both[2 * (size_t)r]; offs[(size_t)r] = both[2 * (size_t)r + 1];
        total += counts[(size_t)r];
        cap = std::max<size_t>(cap, counts[(size_t)r]);
    }
    const size_t blk = cap * sizeof(str_er_cand);
    std::vector<uint8_t> recv_h;
    const uint8_t *recv = nullptr;
    if (c->group) {
        std::vector<uint8_t> send(blk, 0);
        if (n_local) std::memcpy(send.data(), h_local, (size_t)n_local * sizeof(str_er_cand));
        recv_h.resize(blk * (size_t)W);
        const int rc = local_all_gather(c, send.data(), recv_h.data(), blk);
        if (rc != STR_ER_OK) return rc;
        recv = recv_h.data();
    } else {
        Rccl *r = rccl();
        if (ensure(c, c->d_send, c->send_cap, blk) != STR_ER_OK || ensure(c, c->d_recv, c->recv_cap, blk * (size_t)W) != STR_ER_OK) return STR_ER_ENOMEM;
        // the padded send block: device to device when the records are still on the device, else one upload
        if (n_local) {
            const hipError_t e = d_local ? hipMemcpyAsync(c->d_send, d_local, (size_t)n_local * sizeof(str_er_cand), hipMemcpyDeviceToDevice, c->stream)
                                         : hipMemcpyAsync(c->d_send, h_local, (size_t)n_local * sizeof(str_er_cand), hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) return cfail(c, STR_ER_EHIP, "record staging");
        }
        if (r->AllGather(c->d_send, c->d_recv, blk, NCCL_CHAR, c->nccl, c->stream) != 0) return cfail(c, STR_ER_EHIP, "ncclAllGather (records)");
        c->h_recv.resize(blk * (size_t)W);
        if (hipMemcpyAsync(c->h_recv.data(), c->d_recv, blk * (size_t)W, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            return cfail(c, STR_ER_EHIP, "record download");
        recv = c->h_recv.data();
    }
    str_er_cand *out = static_cast<str_er_cand *>(std::malloc(std::max<size_t>(total, 1) * sizeof(str_er_cand)));
    if (!out) return cfail(c, STR_ER_ENOMEM, "result allocation");
    compact(recv, counts.data(), W, cap, offs.data(), out);
    if (counts_out) for (int r = 0; r < W; ++r) counts_out[r] = (int32_t)counts[(size_t)r];
    *all = out;
    *n_all = (int32_t)total;
    return STR_ER_OK;
}

int str_er_gather_cands(str_er_comm *c, const str_er_cand *local, int32_t n_local, uint32_t frame_offset, str_er_cand **all, int32_t *n_all,
                        int32_t *counts)
{
    if (n_local < 0 || (n_local > 0 && !local)) return STR_ER_EINVAL;
    return gather_impl(c, local, nullptr, (uint32_t)n_local, frame_offset, all, n_all, counts);
}

int str_er_gather_last(str_er_comm *c, str_er_ctx *ctx, uint32_t frame_offset, str_er_cand **all, int32_t *n_all, int32_t *counts)
{
    if (!c || !ctx) return STR_ER_EINVAL;
    const void *d = nullptr;
    uint32_t    n = 0;
    int         dev = 0;
    const int   rc = str_er_internal_last_cands(ctx, &d, &n, &dev);
    if (rc != STR_ER_OK) return cfail(c, rc, "the context has no finished detect call");
    if (c->group) return cfail(c, STR_ER_EINVAL, "str_er_gather_last needs an RCCL communicator (device records); use str_er_gather_cands");
    if (dev != c->device) return cfail(c, STR_ER_EINVAL, "context and communicator are on different devices");
    return gather_impl(c, nullptr, d, n, frame_offset, all, n_all, counts);
}

void str_er_gather_free(str_er_cand *p) { std::free(p); }

} // extern "C"
