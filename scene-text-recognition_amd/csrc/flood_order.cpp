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

namespace {

// State per pixel: quantised level (src/ER.cpp:250: 8U -> 8U convertTo with scale 1/step = round-half-even of float(p) * float(1/step)) in the low LBITS bits,
// "watched" above them, "accessible" on top -- ONE byte where the sentinel level 255 / THRESH_STEP + 1 fits 6 bits (THRESH_STEP >= 5), two otherwise.  The plane is framed by a border of accessible
// cells, so the four neighbours of a pixel are cur + 1, cur + S, cur - 1, cur - S without any test for the plane's edge.
// The 256 LIFO buckets of src/ER.cpp:254-255 are contiguous stacks inside one array, level l's from base[l] on with room for every pixel of that level (a
// pixel is in at most one bucket at a time; the counts come from the pass that quantises the plane): a push and a pop touch the top of a stack, not a link
// word next to the pixel -- round 5, 5-10 % less per walk than the linked lists through a second per-pixel array (same stamps: tests/test_flood_order.py).
// (The two arrays are kept per thread: a fresh allocation per walk is thousands of page faults and a memset -- a third of a walk that stops early; both are
// written before they are read.  Round 4, measured and not adopted: the level looked up from the plane when the walk first touches a pixel -- no pass over the
// whole plane in front of a walk that stops early -- is 11 % SLOWER on the boxes' EPYC 9575F: two dependent loads per new pixel instead of one, and the walk is
// a chain of mispredicted branches and dependent loads, ~8 ns per pixel, whatever the bytes.)
template <typename T, int LBITS, bool ALL>
void flood_walk(const uint8_t *pix, int w, int h, int64_t stride, int invert, float qscale, int hi, const uint32_t *watch, uint32_t n_watch,
                uint32_t *stamp, const uint32_t *group, std::vector<T> &st_buf, std::vector<uint32_t> &stk_buf)
{
    const uint32_t n = (uint32_t)w * (uint32_t)h;
    constexpr bool all = ALL;          // (n_watch == 0xFFFFFFFF: every pixel is stamped; a build of its own -- the test sits in the walk's innermost step)
    constexpr T ACC = (T)(1u << (LBITS + 1)), WATCH = (T)(1u << LBITS), LEVEL = (T)((1u << LBITS) - 1u);
    T lut[256];
    for (int v = 0; v < 256; ++v) lut[v] = (T)std::lrintf((float)v * qscale);
    const uint32_t S = (uint32_t)w + 2u;
    const size_t   NP = (size_t)S * ((size_t)h + 2u);
    if (st_buf.size() < NP) st_buf.resize(NP);
    if (stk_buf.size() < (size_t)n + 8) stk_buf.resize((size_t)n + 8);
    T *const st = st_buf.data();
    // (four counters per level: neighbouring pixels mostly share a level, and one counter would make the pass a chain of dependent read-modify-writes)
    uint32_t cnt4[4][257] = {{0}};
    for (uint32_t x = 0; x < S; ++x) { st[x] = ACC; st[(size_t)(h + 1) * S + x] = ACC; }
    for (int y = 0; y < h; ++y) {
        const uint8_t *row = pix + (size_t)y * stride;
        T             *o = st + (size_t)(y + 1) * S + 1;
        o[-1] = ACC; o[w] = ACC;
        int x = 0;
        for (; x + 4 <= w; x += 4) {
            const T l0 = lut[row[x] ^ invert], l1 = lut[row[x + 1] ^ invert], l2 = lut[row[x + 2] ^ invert], l3 = lut[row[x + 3] ^ invert];
            o[x] = l0; o[x + 1] = l1; o[x + 2] = l2; o[x + 3] = l3;
            ++cnt4[0][l0]; ++cnt4[1][l1]; ++cnt4[2][l2]; ++cnt4[3][l3];
        }
        for (; x < w; ++x) { const T l = lut[row[x] ^ invert]; o[x] = l; ++cnt4[0][l]; }
    }
    uint32_t cnt[257];
    for (int l = 0; l < 257; ++l) cnt[l] = cnt4[0][l] + cnt4[1][l] + cnt4[2][l] + cnt4[3][l];
    auto pad = [&](uint32_t p) -> uint32_t { return (p / (uint32_t)w + 1u) * S + p % (uint32_t)w + 1u; };
    auto unpad = [&](uint32_t q) -> uint32_t { const uint32_t y = q / S - 1u; return y * (uint32_t)w + (q - (y + 1u) * S - 1u); };
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
            if (watch[i] < n && !(st[pad(watch[i])] & WATCH)) {
                st[pad(watch[i])] |= WATCH;
                if (group) { if (group_left[open_in_group[i]]++ > 0) ++remaining; }      // k members -> k - 1 needed
                else ++remaining;
            }
        if (group) for (uint32_t i = 0; i < n_watch; ++i) stamp[i] = 0xFFFFFFFFu;
        if (remaining == 0) return;
    }
    uint32_t *const stk = stk_buf.data();
    uint32_t top[257], base[257];
    { uint32_t a = 0; for (int l = 0; l < 257; ++l) { top[l] = base[l] = a; a += cnt[l]; } }
    uint32_t priority = (uint32_t)hi, counter = 0;
    const uint32_t HI = (uint32_t)hi;
    auto mark = [&](uint32_t q, T s) {
        ++counter;
        st[q] = (T)(s | ACC);
        if (all) stamp[unpad(q)] = counter;
        else if (s & WATCH) {
            const uint32_t p = unpad(q);
            for (uint32_t j = 0; j < n_watch; ++j)
                if (watch[j] == p) {          // (a handful per plane)
                    stamp[j] = counter;
                    if (!group) --remaining;
                    else if (--group_left[open_in_group[j]] >= 1) --remaining;       // the group's last member needs no stamp
                    break;
                }
        }
    };
    // same start pixel, same edge order (right, bottom, left, top), same LIFO buckets per level, same "priority == highest_level means empty" rule as
    // src/ER.cpp:254-345: pixels at the sentinel level are marked but never pushed
    uint32_t cur = S + 1u, edge = 0, cl = st[cur] & LEVEL;
    mark(cur, st[cur]);
    const int32_t off[4] = {1, (int32_t)S, -1, -(int32_t)S};
    // The newest push is held back in registers (dv at level dl): in a flat region a step pushes one neighbour and the next step pops that very entry, and
    // through the stack that is a store and two dependent loads on the walk's critical path.  `priority` describes the stacks in memory only.
    uint32_t dv = 0, dl = 0;
    bool     held = false;
    auto push = [&](uint32_t l, uint32_t v) {
        if (l >= HI) return;                                   // (the bucket of the sentinel level is never popped)
        if (held) { stk[top[dl]++] = dv; if (dl < priority) priority = dl; }
        dv = v; dl = l; held = true;
    };
    while (remaining != 0) {
        bool descended = false;
        // the four neighbours' states at once (independent loads), then only the ones not accessible yet, in edge order: marking one does not change another
        const T  sn[4] = {st[cur + 1u], st[cur + S], st[cur - 1u], st[cur - S]};
        uint32_t m = ((sn[0] & ACC) ? 0u : 1u) | ((sn[1] & ACC) ? 0u : 2u) | ((sn[2] & ACC) ? 0u : 4u) | ((sn[3] & ACC) ? 0u : 8u);
        m &= ~((1u << edge) - 1u);
        while (m) {
            const uint32_t k = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            const uint32_t q = cur + (uint32_t)off[k];
            const T        s = sn[k];
            mark(q, s);
            const uint32_t l = s & LEVEL;
            if (l >= cl) push(l, q << 3);
            else {
                push(cl, (cur << 3) | (k + 1u));
                cur = q; cl = l; edge = 0;
                descended = true;
                break;
            }
        }
        if (descended) continue;
        uint32_t v;
        if (held && dl <= priority) { v = dv; cl = dl; held = false; }       // the entry held back is the newest of the lowest level
        else {
            if (held) { stk[top[dl]++] = dv; held = false; }                   // (dl > priority: it stays where it would have been)
            if (priority == HI) break;
            v = stk[--top[priority]];
            cl = priority;
            while (priority < HI && top[priority] == base[priority]) ++priority;
        }
        cur = v >> 3; edge = v & 7u;
    }
}

} // namespace

void flood_order_host(const uint8_t *pix, int w, int h, int64_t stride, int invert, float qscale, int hi, const uint32_t *watch,
                      uint32_t n_watch, uint32_t *stamp, const uint32_t *group)
{
    if (w <= 0 || h <= 0) return;
    // Kept for the planes a video stream brings again and again, up to one 1920 x 1080 plane (2 MB of state + 8 MB of stacks a thread); the scratch of a
    // larger plane (one 4K plane: 42 MB) is handed back when its walk ends, so that up to 64 pool threads and every caller thread do not hold it for the
    // life of the process.
    thread_local std::vector<uint8_t>  st8;
    thread_local std::vector<uint16_t> st16;
    thread_local std::vector<uint32_t> stk;
    struct Shrink {
        std::vector<uint8_t> &a; std::vector<uint16_t> &b; std::vector<uint32_t> &c;
        ~Shrink()
        {
            if (c.size() > ((size_t)1 << 21) + ((size_t)1 << 16)) { std::vector<uint8_t>().swap(a); std::vector<uint16_t>().swap(b); std::vector<uint32_t>().swap(c); }
        }
    } shrink{st8, st16, stk};
    const bool all = n_watch == 0xFFFFFFFFu;
    if (hi <= 63) {
        if (all) flood_walk<uint8_t, 6, true>(pix, w, h, stride, invert, qscale, hi, watch, n_watch, stamp, group, st8, stk);
        else flood_walk<uint8_t, 6, false>(pix, w, h, stride, invert, qscale, hi, watch, n_watch, stamp, group, st8, stk);
    } else {
        if (all) flood_walk<uint16_t, 9, true>(pix, w, h, stride, invert, qscale, hi, watch, n_watch, stamp, group, st16, stk);
        else flood_walk<uint16_t, 9, false>(pix, w, h, stride, invert, qscale, hi, watch, n_watch, stamp, group, st16, stk);
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
