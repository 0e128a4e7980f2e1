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
    bool failed = false;               // a rank hit an error it could not announce: everybody waiting returns an error
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
    g.cv.wait(lk, [&] { return g.left == 0 || g.failed; });          // the previous round has been read by everybody
    if (g.failed) return cfail(c, STR_ER_ESTATE, "a rank of the group failed: the group is unusable");
    g.send[(size_t)c->rank] = send;
    const uint64_t gen = g.generation;
    if (++g.arrived == g.world) { g.left = g.world; g.arrived = 0; ++g.generation; g.cv.notify_all(); }
    else g.cv.wait(lk, [&] { return g.generation != gen || g.failed; });
    if (g.failed) return cfail(c, STR_ER_ESTATE, "a rank of the group failed: the group is unusable");
    for (int r = 0; r < g.world; ++r) std::memcpy(static_cast<uint8_t *>(recv) + (size_t)r * bytes, g.send[(size_t)r], bytes);
    // nobody returns before everybody has read: the send buffers belong to the callers
    const uint64_t done_gen = g.generation;
    if (--g.left == 0) g.cv.notify_all();
    else g.cv.wait(lk, [&] { return g.left == 0 || g.generation != done_gen; });
    return STR_ER_OK;
}

// a rank that cannot go on after its peers may have entered a collective: wake them with an error instead of leaving them waiting
void abort_group(str_er_comm *c)
{
    if (!c->group) return;
    std::lock_guard<std::mutex> lk(c->group->mu);
    c->group->failed = true;
    c->group->cv.notify_all();
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

} // namespace

extern "C" {

// (nothing is thrown across the C ABI)
#define COMM_GUARD(cm)                                                                                    \
    catch (const std::bad_alloc &) { return cfail_noexcept((cm), STR_ER_ENOMEM, "out of host memory"); }  \
    catch (...) { return cfail_noexcept((cm), STR_ER_EHIP, "internal error (exception)"); }
static int cfail_noexcept(str_er_comm *c, int code, const char *msg)
{
    try { return cfail(c, code, msg); } catch (...) { return code; }
}

int str_er_comm_unique_id(void *id128)
try {
    if (!id128) return STR_ER_EINVAL;
    Rccl *r = rccl();
    if (!r) return STR_ER_ESTATE;
    return r->GetUniqueId(id128) == 0 ? STR_ER_OK : STR_ER_EHIP;
} COMM_GUARD(nullptr)

int str_er_comm_create(int32_t device, int32_t rank, int32_t world, const void *id128, str_er_comm **out)
try {
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
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        cfail(nullptr, STR_ER_EHIP, "stream / buffer creation for the communicator failed");
        str_er_comm_destroy(c);
        return STR_ER_EHIP;
    }
    *out = c;
    return STR_ER_OK;
} COMM_GUARD(nullptr)

int str_er_comm_local_group(int32_t world, str_er_comm_group **out)
try {
    if (!out || world < 1 || world > 1024) return STR_ER_EINVAL;
    str_er_comm_group *g = new (std::nothrow) str_er_comm_group();
    if (!g) return STR_ER_ENOMEM;
    g->g = std::make_shared<LocalGroup>();
    g->g->world = world;
    g->g->send.assign((size_t)world, nullptr);
    *out = g;
    return STR_ER_OK;
} COMM_GUARD(nullptr)

void str_er_comm_local_group_free(str_er_comm_group *g) { delete g; }

int str_er_comm_create_local(str_er_comm_group *g, int32_t rank, str_er_comm **out)
try {
    if (!g || !out || rank < 0 || rank >= g->g->world) return STR_ER_EINVAL;
    str_er_comm *c = new (std::nothrow) str_er_comm();
    if (!c) return STR_ER_ENOMEM;
    c->rank = rank; c->world = g->g->world; c->group = g->g;
    *out = c;
    return STR_ER_OK;
} COMM_GUARD(nullptr)

void str_er_comm_destroy(str_er_comm *c)
{
    if (!c) return;
    if (c->nccl) {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (Rccl *r = rccl()) (void)r->CommDestroy(c->nccl);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (void *p : {(void *)c->d_send, (void *)c->d_recv}) if (p) (void)hipFree(p);
    delete c;
}

const char *str_er_comm_last_error(const str_er_comm *c) { return c ? c->err.c_str() : g_comm_create_error.c_str(); }
int32_t str_er_comm_rank(const str_er_comm *c) { return c ? c->rank : -1; }
int32_t str_er_comm_world(const str_er_comm *c) { return c ? c->world : 0; }

// ---- the exchange itself: a variable-length all-gather ---------------------------------------------------------------------------
// Every rank contributes n bytes (from host memory, or from device memory: d_local) and one 32-bit tag.  Header round first
// (16 bytes per rank: size, tag), then the payload padded to the largest size.  `local_rc` != 0 announces in the header round that this
// rank cannot take part (bad arguments, wrong device, nothing to send): then EVERY rank returns an error and no rank is left
// waiting in the payload round.  A failure after the header round (out of memory, a HIP error) cannot be announced any more: the rank
// marks an in-process group as failed, which wakes its peers with an error; RCCL peers are left to their own time-outs.
struct VarGather {
    std::vector<uint64_t> sizes;
    std::vector<uint32_t> tags;
    size_t cap = 0;                    // bytes per rank block
    const uint8_t *host = nullptr;     // world blocks of cap bytes (want_device = false)
    uint8_t *dev = nullptr;            // the same in the communicator's device buffer (want_device = true)
    std::vector<uint8_t> store;
};

static int all_gather_var(str_er_comm *c, int local_rc, const char *local_msg, const void *h_local, const void *d_local, uint64_t n, uint32_t tag,
                          bool want_device, VarGather &out)
{
    const int W = c->world;
    struct Head { uint64_t n; uint32_t tag, pad; };
    static_assert(sizeof(Head) == 16, "header layout");
    Head mine{local_rc != 0 ? ~0ull : n, tag, 0};
    std::vector<Head> heads((size_t)W);
    Rccl *r = c->group ? nullptr : rccl();
    if (c->group) {
        const int rc = local_all_gather(c, &mine, heads.data(), sizeof(Head));
        if (rc != STR_ER_OK) return rc;
    } else {
        // (failures here precede every collective of this call on this rank only if they also strike the peers: the device and the
        // header buffers were set up when the communicator was made or by the previous call)
        if (hipSetDevice(c->device) != hipSuccess) return cfail(c, STR_ER_EHIP, "hipSetDevice");
        const size_t need = sizeof(Head) * (size_t)W + sizeof(Head);
        if (ensure(c, c->d_send, c->send_cap, need) != STR_ER_OK || ensure(c, c->d_recv, c->recv_cap, need) != STR_ER_OK) return STR_ER_ENOMEM;
        if (hipMemcpyAsync(c->d_send, &mine, sizeof(Head), hipMemcpyHostToDevice, c->stream) != hipSuccess) return cfail(c, STR_ER_EHIP, "header upload");
        if (r->AllGather(c->d_send, c->d_recv, sizeof(Head), NCCL_CHAR, c->nccl, c->stream) != 0) return cfail(c, STR_ER_EHIP, "ncclAllGather (headers)");
        if (hipMemcpyAsync(heads.data(), c->d_recv, sizeof(Head) * (size_t)W, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            return cfail(c, STR_ER_EHIP, "header download");
    }
    out.sizes.resize((size_t)W); out.tags.resize((size_t)W);
    size_t cap = 1;
    int bad = -1;
    for (int k = 0; k < W; ++k) {
        if (heads[(size_t)k].n == ~0ull) { if (bad < 0) bad = k; continue; }
        out.sizes[(size_t)k] = heads[(size_t)k].n; out.tags[(size_t)k] = heads[(size_t)k].tag;
        cap = std::max<size_t>(cap, (size_t)heads[(size_t)k].n);
    }
    if (bad >= 0)     // every rank sees the same headers: every rank returns here
        return cfail(c, local_rc != 0 ? local_rc : STR_ER_ESTATE,
                     local_rc != 0 ? std::string(local_msg) : "rank " + std::to_string(bad) + " could not take part in the exchange; nothing was exchanged");
    cap = (cap + 15) & ~(size_t)15;
    out.cap = cap;
    auto die = [&](int code, const char *msg) { abort_group(c); return cfail(c, code, msg); };
    if (c->group) {
        std::vector<uint8_t> send(cap, 0);
        if (n) {
            if (d_local) { if (hipMemcpy(send.data(), d_local, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return die(STR_ER_EHIP, "payload download"); }
            else std::memcpy(send.data(), h_local, (size_t)n);
        }
        out.store.resize(cap * (size_t)W);
        const int rc = local_all_gather(c, send.data(), out.store.data(), cap);
        if (rc != STR_ER_OK) return rc;
        out.host = out.store.data();
        if (want_device) {          // (a process with a GPU: the ranks of the group are threads with a context each)
            if (ensure(c, c->d_recv, c->recv_cap, cap * (size_t)W) != STR_ER_OK) return die(STR_ER_ENOMEM, "hipMalloc (gather buffer)");
            if (hipMemcpy(c->d_recv, out.store.data(), cap * (size_t)W, hipMemcpyHostToDevice) != hipSuccess) return die(STR_ER_EHIP, "payload upload");
            out.dev = c->d_recv;
        }
        return STR_ER_OK;
    }
    if (ensure(c, c->d_send, c->send_cap, cap) != STR_ER_OK || ensure(c, c->d_recv, c->recv_cap, cap * (size_t)W) != STR_ER_OK) return STR_ER_ENOMEM;
    if (n) {      // the padded send block: device to device when the bytes are still on the device, else one upload
        const hipError_t e = d_local ? hipMemcpyAsync(c->d_send, d_local, (size_t)n, hipMemcpyDeviceToDevice, c->stream)
                                     : hipMemcpyAsync(c->d_send, h_local, (size_t)n, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return cfail(c, STR_ER_EHIP, "payload staging");
    }
    if (r->AllGather(c->d_send, c->d_recv, cap, NCCL_CHAR, c->nccl, c->stream) != 0) return cfail(c, STR_ER_EHIP, "ncclAllGather (payload)");
    if (want_device) {
        if (hipStreamSynchronize(c->stream) != hipSuccess) return cfail(c, STR_ER_EHIP, "ncclAllGather (payload) failed");
        out.dev = c->d_recv;
        return STR_ER_OK;
    }
    c->h_recv.resize(cap * (size_t)W);
    if (hipMemcpyAsync(c->h_recv.data(), c->d_recv, cap * (size_t)W, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return cfail(c, STR_ER_EHIP, "payload download");
    out.host = c->h_recv.data();
    return STR_ER_OK;
}

// counts[world] (may be null); *all is malloc'ed (free with str_er_gather_free), ordered by rank.
static int gather_impl(str_er_comm *c, int local_rc, const char *local_msg, const str_er_cand *h_local, const void *d_local, uint32_t n_local,
                       uint32_t frame_offset, str_er_cand **all, int32_t *n_all, int32_t *counts_out)
{
    if (!c || !all || !n_all) return STR_ER_EINVAL;
    *all = nullptr; *n_all = 0;
    const int W = c->world;
    VarGather g;
    const int rc = all_gather_var(c, local_rc, local_msg, h_local, d_local, (uint64_t)n_local * sizeof(str_er_cand), frame_offset, false, g);
    if (rc != STR_ER_OK) return rc;
    std::vector<uint32_t> counts((size_t)W);
    size_t total = 0;
    for (int k = 0; k < W; ++k) { counts[(size_t)k] = (uint32_t)(g.sizes[(size_t)k] / sizeof(str_er_cand)); total += counts[(size_t)k]; }
    str_er_cand *out = static_cast<str_er_cand *>(std::malloc(std::max<size_t>(total, 1) * sizeof(str_er_cand)));
    if (!out) return cfail(c, STR_ER_ENOMEM, "result allocation");
    size_t at = 0;
    for (int k = 0; k < W; ++k) {
        const str_er_cand *blk = reinterpret_cast<const str_er_cand *>(g.host + (size_t)k * g.cap);
        for (uint32_t i = 0; i < counts[(size_t)k]; ++i) { out[at] = blk[i]; out[at].frame += g.tags[(size_t)k]; ++at; }
    }
    if (counts_out) for (int k = 0; k < W; ++k) counts_out[k] = (int32_t)counts[(size_t)k];
    *all = out;
    *n_all = (int32_t)total;
    return STR_ER_OK;
}

int str_er_gather_cands(str_er_comm *c, const str_er_cand *local, int32_t n_local, uint32_t frame_offset, str_er_cand **all, int32_t *n_all,
                        int32_t *counts)
try {
    if (!c) return STR_ER_EINVAL;
    const bool bad = n_local < 0 || (n_local > 0 && !local);
    return gather_impl(c, bad ? STR_ER_EINVAL : 0, "bad arguments", local, nullptr, bad ? 0u : (uint32_t)n_local, frame_offset, all, n_all, counts);
} COMM_GUARD(c)

int str_er_gather_last(str_er_comm *c, str_er_ctx *ctx, uint32_t frame_offset, str_er_cand **all, int32_t *n_all, int32_t *counts)
try {
    if (!c || !ctx) return STR_ER_EINVAL;
    const void *d = nullptr;
    uint32_t    n = 0;
    int         dev = 0;
    // (what is wrong on this rank only is announced to the others in the header round: they return an error too instead of waiting)
    int         lrc = str_er_internal_last_cands(ctx, &d, &n, &dev);
    const char *msg = "the context has no finished detect call";
    if (lrc == STR_ER_OK && c->group) { lrc = STR_ER_EINVAL; msg = "str_er_gather_last needs an RCCL communicator (device records); use str_er_gather_cands"; }
    if (lrc == STR_ER_OK && dev != c->device) { lrc = STR_ER_EINVAL; msg = "context and communicator are on different devices"; }
    return gather_impl(c, lrc, msg, nullptr, d, lrc == STR_ER_OK ? n : 0u, frame_offset, all, n_all, counts);
} COMM_GUARD(c)

// Variable-length all-gather of bytes (strip blobs).  out_kind HOST: *all is malloc'ed (str_er_comm_free), the contributions
// back to back; out_kind DEVICE: *all points into the communicator's device buffer (valid until its next collective), rank k's bytes at
// starts[k].  starts / sizes: world entries each.
int str_er_comm_allgather_bytes(str_er_comm *c, const void *local, int64_t n_local, int in_kind, int out_kind, void **all, int64_t *starts, int64_t *sizes)
try {
    if (!c || !all || !starts || !sizes) return STR_ER_EINVAL;
    *all = nullptr;
    const bool bad = n_local < 0 || (n_local > 0 && !local) || (in_kind != STR_ER_MEM_HOST && in_kind != STR_ER_MEM_DEVICE) ||
                     (out_kind != STR_ER_MEM_HOST && out_kind != STR_ER_MEM_DEVICE);
    VarGather g;
    const int rc = all_gather_var(c, bad ? STR_ER_EINVAL : 0, "bad arguments", in_kind == STR_ER_MEM_HOST ? local : nullptr,
                                  in_kind == STR_ER_MEM_DEVICE ? local : nullptr, bad ? 0u : (uint64_t)n_local, 0u, out_kind == STR_ER_MEM_DEVICE, g);
    if (rc != STR_ER_OK) return rc;
    const int W = c->world;
    if (out_kind == STR_ER_MEM_DEVICE) {
        for (int k = 0; k < W; ++k) { starts[k] = (int64_t)((size_t)k * g.cap); sizes[k] = (int64_t)g.sizes[(size_t)k]; }
        *all = g.dev;
        return STR_ER_OK;
    }
    size_t total = 0;
    for (int k = 0; k < W; ++k) total += (size_t)g.sizes[(size_t)k];
    uint8_t *out = static_cast<uint8_t *>(std::malloc(std::max<size_t>(total, 1)));
    if (!out) return cfail(c, STR_ER_ENOMEM, "result allocation");
    size_t at = 0;
    for (int k = 0; k < W; ++k) {
        std::memcpy(out + at, g.host + (size_t)k * g.cap, (size_t)g.sizes[(size_t)k]);
        starts[k] = (int64_t)at; sizes[k] = (int64_t)g.sizes[(size_t)k];
        at += (size_t)g.sizes[(size_t)k];
    }
    *all = out;
    return STR_ER_OK;
} COMM_GUARD(c)

void str_er_comm_free(void *p) { std::free(p); }

void str_er_gather_free(str_er_cand *p) { std::free(p); }

} // extern "C"
