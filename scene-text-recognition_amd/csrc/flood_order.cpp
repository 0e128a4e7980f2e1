// flood_order.cpp -- replay of the ORDER of the reference's flood on a host core.
//
// Where two or more child chains compete for a parent in non_maximum_supression (src/ER.cpp:416-462) the reference's winner is
// the child its flood entered last (children are prepended when merged, src/ER.cpp:183-185).  That order is the visiting order
// of a depth-first walk with a fixed neighbour order over the pixel grid -- an inherently sequential computation (ordered
// depth-first search is P-complete): every step depends on the marks the previous step left.  The GPU version of the same walk
// (k_flood_order, er_kernels.hip) spends ~0.8 us per pixel on one lane, a host core ~10-20 ns, and the walk is needed for roughly one
// plane in a thousand (single two-way ties whose two outcomes give different pools), so this is the one piece of the NMS that
// runs where the reference's own flood runs: on the host -- like the greedy line assignment of er_grouping (er_group.cpp).
// Nothing is built here (the component tree, the kept nodes, the pools all come from the GPU); the walk only stamps pixels.
//
// Same start pixel, same edge order (right, bottom, left, top), same LIFO buckets per level, same "priority == highest_level
// means empty" rule as src/ER.cpp:254-345: pixels at the sentinel level are marked but never popped.
#include "flood_order.h"

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace str_er {

void flood_order_host(const uint8_t *pix, int w, int h, int64_t stride, int invert, float qscale, int hi, const uint32_t *watch,
                      uint32_t n_watch, uint32_t *stamp, const uint32_t *group)
{
    const uint32_t n = (uint32_t)w * (uint32_t)h;
    if (n == 0) return;
    const bool all = n_watch == 0xFFFFFFFFu;
    // per pixel: quantised level (src/ER.cpp:250: 8U -> 8U convertTo with scale 1/step = round-half-even of float(p) * float(1/step))
    // in bits 0..8, "watched" in bit 14, "accessible" in bit 15 -- one 16-bit word is all the walk reads per neighbour
    constexpr uint16_t ACC = 0x8000u, WATCH = 0x4000u, LEVEL = 0x01FFu;
    uint16_t lut[256];
    for (int v = 0; v < 256; ++v) lut[v] = (uint16_t)std::lrintf((float)v * qscale);
    // (the two per-pixel arrays are kept per thread: a fresh 12 MB allocation per walk is ~3000 page faults and a memset -- a third of a
    // walk that stops early; both are written before they are read.  Round 4, measured and not adopted: one byte of state per pixel and the level
    // looked up from the plane when the walk first touches a pixel -- no pass over the whole plane in front of a walk that stops early -- is 11 %
    // SLOWER on the boxes' EPYC 9575F (16.0 -> 17.9 ms for a whole 1920 x 1080 plane, tools/walk_bench.cpp): two dependent loads per new pixel
    // instead of one, and the walk is a chain of mispredicted branches and dependent loads, ~8 ns per pixel, whatever the bytes)
    // Kept for the planes a video stream brings again and again, up to one 1920 x 1080 plane (12.4 MB a thread: 0.8 GB for a pool of 64); the scratch
    // of a larger plane (one 4K plane: 50 MB) is handed back when its walk ends, so that up to 64 pool threads and every caller thread do not hold it
    // for the life of the process.
    thread_local std::vector<uint16_t> st_buf;
    thread_local std::vector<uint32_t> link_buf;
    struct Shrink {
        std::vector<uint16_t> &a; std::vector<uint32_t> &b;
        ~Shrink() { if (a.size() > ((size_t)1 << 21) + ((size_t)1 << 16)) { std::vector<uint16_t>().swap(a); std::vector<uint32_t>().swap(b); } }
    } shrink{st_buf, link_buf};
    if (st_buf.size() < n) st_buf.resize(n);
    if (link_buf.size() < n) link_buf.resize(n);
    uint16_t *const st = st_buf.data();
    for (int y = 0; y < h; ++y) {
        const uint8_t *row = pix + (size_t)y * stride;
        uint16_t      *o = st + (size_t)y * w;
        for (int x = 0; x < w; ++x) o[x] = lut[row[x] ^ invert];
    }
    // `remaining` = watched pixels whose stamps are still needed.  With groups (the children competing for one parent) only the
    // LAST one entered matters: once all but one member of a group are stamped that one is known to come later, so a group of k
    // needs k - 1 stamps; members left unstamped get 0xFFFFFFFF ("later than every stamped one").
    uint32_t remaining = 0xFFFFFFFFu;
    std::vector<uint32_t> open_in_group;        // per watch entry: index of its group's counter
    std::vector<uint32_t> group_left;           // per group: members not stamped yet
    if (!all) {
        remaining = 0;
        if (group) {
            open_in_group.resize(n_watch);
            for (uint32_t i = 0; i < n_watch; ++i) {
                uint32_t g = 0;
                while (g < i && group[g] != group[i]) ++g;          // (first entry with the same parent; a handful of entries)
                open_in_group[i] = g;
            }
            group_left.assign(n_watch, 0);
        }
        for (uint32_t i = 0; i < n_watch; ++i)
            if (watch[i] < n && !(st[watch[i]] & WATCH)) {
                st[watch[i]] |= WATCH;
                if (group) { if (group_left[open_in_group[i]]++ > 0) ++remaining; }      // k members -> k - 1 needed
                else ++remaining;
            }
        if (group) for (uint32_t i = 0; i < n_watch; ++i) stamp[i] = 0xFFFFFFFFu;
        if (remaining == 0) return;
    }
    // the 256 LIFO buckets of src/ER.cpp:254-255 as linked lists through one array (a pixel is in at most one bucket at a time):
    // link[p] = next entry << 3 | the edge at which p resumes
    constexpr uint32_t NIL = 0x1FFFFFFFu;
    uint32_t *const link = link_buf.data();
    uint32_t head[257];
    for (uint32_t &v : head) v = NIL;
    uint32_t priority = (uint32_t)hi, counter = 0;
    uint32_t cur = 0, edge = 0, cl = st[0] & LEVEL;
    const uint32_t W = (uint32_t)w, HI = (uint32_t)hi;
    auto mark = [&](uint32_t p, uint16_t s) {
        ++counter;
        st[p] = (uint16_t)(s | ACC);
        if (all) stamp[p] = counter;
        else if (s & WATCH) {
            for (uint32_t j = 0; j < n_watch; ++j)
                if (watch[j] == p) {          // (a handful per plane)
                    stamp[j] = counter;
                    if (!group) --remaining;
                    else if (--group_left[open_in_group[j]] >= 1) --remaining;       // the group's last member needs no stamp
                    break;
                }
        }
    };
    mark(0, st[0]);
    uint32_t x = 0;       // column of `cur`
    while (remaining != 0) {
        bool descended = false;
        for (; edge < 4; ++edge) {
            uint32_t q;
            switch (edge) {
            case 0: if (x + 1 >= W) continue; q = cur + 1; break;
            case 1: if (cur + W >= n) continue; q = cur + W; break;
            case 2: if (x == 0) continue; q = cur - 1; break;
            default: if (cur < W) continue; q = cur - W; break;
            }
            const uint16_t s = st[q];
            if (s & ACC) continue;
            mark(q, s);
            const uint32_t l = s & LEVEL;
            if (l >= cl) {
                if (l < HI) { link[q] = head[l] << 3; head[l] = q; }      // (the bucket of the sentinel level is never popped)
                if (l < priority) priority = l;
            } else {
                if (cl < HI) { link[cur] = (head[cl] << 3) | (edge + 1); head[cl] = cur; }
                if (cl < priority) priority = cl;
                cur = q; cl = l;
                x = edge == 0 ? x + 1 : edge == 2 ? x - 1 : x;
                edge = 0;
                descended = true;
                break;
            }
        }
        if (descended) continue;
        if (priority == HI) break;
        cur = head[priority];
        const uint32_t v = link[cur];
        head[priority] = v >> 3; edge = v & 7u; cl = priority;
        x = cur % W;
        while (priority < HI && head[priority] == NIL) ++priority;
    }
}

// ---- the pool ----------------------------------------------------------------------------------------------------------------
// CPUs this process can keep busy: hardware threads, cut down to the cgroup's CPU quota where there is one (v2: cpu.max "quota period",
// v1: cpu.cfs_quota_us / cpu.cfs_period_us)
int flood_host_cpus()
{
    unsigned hw = std::thread::hardware_concurrency();
    int n = hw ? (int)hw : 2;
    auto read2 = [](const char *path, long long &a, long long &b) -> bool {
        std::FILE *f = std::fopen(path, "r");
        if (!f) return false;
        char tok[64] = {0};
        const int got = std::fscanf(f, "%63s %lld", tok, &b);
        std::fclose(f);
        if (got < 1 || std::strcmp(tok, "max") == 0) return false;
        a = std::atoll(tok);
        return got == 2 && a > 0 && b > 0;
    };
    long long quota = 0, period = 0;
    // cgroup v2: the process's own group ("0::/path" in /proc/self/cgroup) and every ancestor up to the root may carry a quota: the tightest one counts
    bool have = false;
    {
        char line[512] = {0}, path[640];
        std::FILE *f = std::fopen("/proc/self/cgroup", "r");
        std::string own;
        while (f && std::fgets(line, sizeof line, f))
            if (std::strncmp(line, "0::", 3) == 0) { own = line + 3; while (!own.empty() && (own.back() == '\n' || own.back() == '/')) own.pop_back(); }
        if (f) std::fclose(f);
        for (;;) {
            std::snprintf(path, sizeof path, "/sys/fs/cgroup%s/cpu.max", own.c_str());
            long long q = 0, pr = 0;
            if (read2(path, q, pr) && (!have || q * period < quota * pr)) { quota = q; period = pr; have = true; }
            const size_t cut = own.rfind('/');
            if (own.empty()) break;
            own.erase(cut == std::string::npos ? 0 : cut);
        }
    }
    if (!have) {
        long long dummy = 0;
        std::FILE *f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        if (f) { if (std::fscanf(f, "%lld", &quota) != 1) quota = 0; std::fclose(f); }
        f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (f) { if (std::fscanf(f, "%lld", &period) != 1) period = 0; std::fclose(f); }
        (void)dummy;
        have = quota > 0 && period > 0;
    }
    if (have) {
        const long long q = (quota + period - 1) / period;
        if (q >= 1 && q < n) n = (int)q;
    }
    return n < 1 ? 1 : n;
}

namespace {

struct JobSet {                        // the walks of one call
    size_t n = 0;
    void (*fn)(size_t, void *) = nullptr;
    void *arg = nullptr;
    std::atomic<size_t> next{0}, done{0};
    std::atomic<int> err{0};
};

struct Pool {
    std::mutex mu;
    std::condition_variable work, finished;
    std::deque<JobSet *> active;       // sets that still have indices to hand out
    int n_threads = 0;
};

void run_one(JobSet *js, size_t k)
{
    try { js->fn(k, js->arg); }
    catch (const std::bad_alloc &) { js->err.store(-1); }
    catch (...) { js->err.store(-2); }
}

Pool *pool()
{
    static Pool *p = [] {
        Pool *q = new Pool();          // (never destroyed: its threads outlive every context and wait on it)
        // As many threads as the process may really run at once, at most 64: the walks of a batch are independent and latency-bound.  "May
        // run" is not hardware_concurrency(): a container's CPU quota (cgroup cpu.max) caps the process below the core count -- the round-3
        // boxes report 256 cores and grant 16, which is why 32 walks at once took twice as long each as 16 there (round 3 read that as
        // memory latency and capped the pool at 16 for every host).
        int n = flood_host_cpus();
        if (const char *e = std::getenv("STR_ER_WALK_THREADS")) n = std::atoi(e);
        q->n_threads = n < 1 ? 1 : (n > 64 ? 64 : n);
        for (int i = 0; i < q->n_threads - 1; ++i) {      // (the caller of flood_walks_run is the n-th worker of its own set)
            try {
                std::thread([q] {
                    std::unique_lock<std::mutex> lk(q->mu);
                    for (;;) {
                        q->work.wait(lk, [q] { return !q->active.empty(); });
                        JobSet *js = q->active.front();
                        const size_t k = js->next.fetch_add(1);
                        if (k >= js->n) { if (!q->active.empty() && q->active.front() == js) q->active.pop_front(); continue; }
                        if (k + 1 >= js->n) q->active.pop_front();
                        lk.unlock();
                        run_one(js, k);
                        lk.lock();
                        if (js->done.fetch_add(1) + 1 == js->n) q->finished.notify_all();
                    }
                }).detach();
            } catch (...) { q->n_threads = i + 1; break; }       // (no more threads to be had: the callers do the rest themselves)
        }
        return q;
    }();
    return p;
}

} // namespace

int flood_walk_threads() { return pool()->n_threads; }

int flood_walks_run(size_t n, void (*fn)(size_t, void *), void *arg)
{
    if (n == 0) return 0;
    JobSet js;
    js.n = n; js.fn = fn; js.arg = arg;
    Pool *q = pool();
    if (n > 1 && q->n_threads > 1) {
        std::lock_guard<std::mutex> lk(q->mu);
        q->active.push_back(&js);
        q->work.notify_all();
    }
    for (;;) {                          // the caller takes indices of its own set like any worker
        const size_t k = js.next.fetch_add(1);
        if (k >= n) break;
        run_one(&js, k);
        js.done.fetch_add(1);
    }
    std::unique_lock<std::mutex> lk(q->mu);
    for (auto it = q->active.begin(); it != q->active.end(); ++it) if (*it == &js) { q->active.erase(it); break; }
    q->finished.wait(lk, [&] { return js.done.load() >= n; });
    return js.err.load();
}

} // namespace str_er
