// stream_api.cpp -- frame ingest: host frames in, results out, with the uploads of one batch
// overlapping the kernels of the others.
//
// The reference reads frames on the host (cv::imread, `cap >> frame`, src/utils.cpp:31, 59-82, 109) and hands each one to
// text_detect.  Here a *stream* owns `depth` detector contexts (each with its own HIP stream and workspace), one
// page-locked staging buffer per context and one worker thread per context: the producer acquires a staging buffer,
// decodes / copies its frames straight into it, submits; the worker issues the H2D copy (true DMA, the buffer is pinned)
// and the kernels on its context's stream while the other contexts compute; results come back in submission order.
// Built on the public C ABI only (str_er_create / str_er_detect_bgr / ...).
//
// Round 4: the uploads take TURNS, in submission order, and a batch's kernels start only behind its own upload.  Before, every worker handed its
// pinned buffer to str_er_detect_bgr (upload + kernels in one call): the `depth` uploads in flight shared the host link and the `depth` batches of
// kernels shared the GPU -- jobs of two stages whose resources are both time-shared fall into step (all uploading, then all computing), and the
// stream ran at 1 / (upload + compute) instead of 1 / max(upload, compute): NV12 frames, a third of the link's capacity, at 0.81 of the
// device-resident rate.  Now one upload owns the link at a time (a batch's frames go into a device buffer of its slot, own copy stream), the
// next one starts when it has landed, and the kernels of the batches before it run meanwhile.
#include "../../include/str_er.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct str_er_stream {
    static constexpr int MAX_UPLOAD_STREAMS = 4;
    struct Slot {
        str_er_ctx *ctx = nullptr;
        uint8_t    *pinned = nullptr;
        uint8_t    *d_in = nullptr;        // the batch's frames on the device (upload target)
        hipStream_t copy[MAX_UPLOAD_STREAMS] = {};          // the upload's streams: the parts of a batch travel side by side (one DMA engine each)
        hipEvent_t  landed[MAX_UPLOAD_STREAMS] = {};
        // job
        bool     busy = false, has_job = false, done = false;
        int32_t  w = 0, h = 0, n_frames = 0;
        int64_t  stride = 0, pitch = 0;
        uint32_t stages = 0;
        bool     nv12 = false;
        uint64_t ticket = 0;
        int      rc = STR_ER_OK;
        str_er_result *result = nullptr;
        std::string err;
        std::thread worker;
    };
    std::vector<Slot> slots;
    size_t   slot_bytes = 0;
    int      device = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> order;       // slots in submission order, oldest first
    uint64_t next_ticket = 1;
    uint64_t upload_turn = 1;    // ticket of the batch whose upload is enqueued next
    hipEvent_t last_landed[MAX_UPLOAD_STREAMS] = {};               // events of the upload enqueued last (touched by the worker whose turn it is)
    size_t   upload_piece = (size_t)1 << 40;       // (developer knob STR_ER_UPLOAD_PIECE_MB: the upload in pieces; 4 / 16 MB made no difference that stands out of the run-to-run scatter)
    int      upload_streams = 2;                   // (developer knob STR_ER_UPLOAD_STREAMS = 1 .. 4; 1: the whole batch through one copy stream, as in rounds 4 and 5)
    bool     stop = false;
    std::string err;
};

namespace {

void worker_main(str_er_stream *s, int idx)
{
    str_er_stream::Slot &sl = s->slots[(size_t)idx];
    (void)hipSetDevice(s->device);
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->stop || sl.has_job; });
            if (s->stop && !sl.has_job) return;
        }
        str_er_result *r = nullptr;
        int rc = STR_ER_OK;
        bool upload_failed = false;
        // (nothing may leave this thread as an exception -- std::terminate -- and the upload turn must pass on whatever happens: every later ticket waits for it)
        try {
            // the upload, when it is this batch's turn (submission order); the turn passes on when the bytes have landed
            const int64_t rows = sl.nv12 ? (int64_t)sl.h + sl.h / 2 : (int64_t)sl.h;
            const size_t bytes = (size_t)(sl.n_frames - 1) * (size_t)sl.pitch + (size_t)sl.stride * (size_t)rows;
            {
                std::unique_lock<std::mutex> lk(s->mu);
                s->cv.wait(lk, [&] { return s->upload_turn == sl.ticket; });
            }
            hipError_t e = hipSuccess;
            int        parts = 1;
            {
                // (the turn passes on as soon as this upload is ENQUEUED: the next one is ordered behind it on the device -- its streams wait for this one's
                // events -- so the link never idles while a worker thread wakes up; round 5 passed the turn when the bytes had landed: ~0.2 ms of an idle
                // link per 6 ms upload)
                struct TurnGuard {
                    str_er_stream *s; uint64_t next;
                    ~TurnGuard() { { std::lock_guard<std::mutex> lk(s->mu); s->upload_turn = next; } s->cv.notify_all(); }
                } pass_on{s, sl.ticket + 1};
                const size_t piece = s->upload_piece;
                // the batch in parts on as many streams: a copy stream is served by one DMA engine, which alone does not fill the link
                parts = bytes >= ((size_t)8 << 20) ? std::max(1, std::min(s->upload_streams, (int)str_er_stream::MAX_UPLOAD_STREAMS)) : 1;
                const size_t part = ((bytes / (size_t)parts + 4095) & ~(size_t)4095);
                for (int k = 0; k < str_er_stream::MAX_UPLOAD_STREAMS; ++k)
                    if (s->last_landed[k])
                        for (int j = 0; j < parts && e == hipSuccess; ++j) e = hipStreamWaitEvent(sl.copy[j], s->last_landed[k], 0);
                for (int j = 0; j < parts; ++j) {
                    const size_t lo = std::min(bytes, (size_t)j * part), hi = j + 1 == parts ? bytes : std::min(bytes, (size_t)(j + 1) * part);
                    for (size_t at = lo; at < hi && e == hipSuccess; at += piece)
                        e = hipMemcpyAsync(sl.d_in + at, sl.pinned + at, std::min(piece, hi - at), hipMemcpyHostToDevice, sl.copy[j]);
                    if (e == hipSuccess) e = hipEventRecord(sl.landed[j], sl.copy[j]);
                }
                for (int k = 0; k < str_er_stream::MAX_UPLOAD_STREAMS; ++k) s->last_landed[k] = e == hipSuccess && k < parts ? sl.landed[k] : nullptr;
            }
            // (poll, then sleep between polls: a spinning wait per slot would keep `depth` host cores busy)
            for (int spins = 0; e == hipSuccess;) {
                hipError_t q = hipSuccess;
                for (int j = 0; j < parts && q == hipSuccess; ++j) q = hipEventQuery(sl.landed[j]);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) { e = q; break; }
                if (++spins > 50) std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            if (e != hipSuccess) { rc = STR_ER_EHIP; upload_failed = true; sl.err = std::string("upload: ") + hipGetErrorString(e); }
            if (rc == STR_ER_OK)
                rc = sl.nv12 ? str_er_detect_nv12(sl.ctx, sl.d_in, sl.w, sl.h, sl.stride, sl.pitch, sl.n_frames, STR_ER_MEM_DEVICE, sl.stages, &r)
                             : str_er_detect_bgr(sl.ctx, sl.d_in, sl.w, sl.h, sl.stride, sl.pitch, sl.n_frames, STR_ER_MEM_DEVICE, sl.stages, &r);
        } catch (...) {
            rc = STR_ER_ENOMEM; r = nullptr; upload_failed = true;
            try { sl.err = "stream worker: out of host memory"; } catch (...) { }
        }
        {
            std::lock_guard<std::mutex> lk(s->mu);
            sl.rc = rc;
            sl.result = r;
            if (rc != STR_ER_OK && !upload_failed) { try { sl.err = str_er_last_error(sl.ctx); } catch (...) { } }
            sl.has_job = false;
            sl.done = true;
        }
        s->cv.notify_all();
    }
}

} // namespace

extern "C" {

// (nothing is thrown across the C ABI: thread start, vectors and strings can run out of resources)
#define STREAM_GUARD(st)                                                                                          \
    catch (const std::bad_alloc &) { str_er_stream *s_ = (st); if (s_) { try { s_->err = "out of host memory"; } catch (...) { } } return STR_ER_ENOMEM; } \
    catch (...) { str_er_stream *s_ = (st); if (s_) { try { s_->err = "internal error (exception)"; } catch (...) { } } return STR_ER_EHIP; }

int str_er_stream_create(const str_er_params *p, int32_t depth, str_er_stream **out)
try {
    if (!p || !out || depth < 1 || depth > 16) return STR_ER_EINVAL;
    *out = nullptr;
    str_er_stream *s = new (std::nothrow) str_er_stream();
    if (!s) return STR_ER_ENOMEM;
    s->device = p->device;
    if (const char *e = std::getenv("STR_ER_UPLOAD_STREAMS")) s->upload_streams = std::max(1, std::min(std::atoi(e), (int)str_er_stream::MAX_UPLOAD_STREAMS));
    if (const char *e = std::getenv("STR_ER_UPLOAD_PIECE_MB")) { const long v = std::atol(e); if (v >= 1 && v <= 4096) s->upload_piece = (size_t)v << 20; }
    s->slot_bytes = (size_t)p->max_frames * (size_t)p->max_width * (size_t)p->max_height * 3;
    s->slots.resize((size_t)depth);
    int rc = STR_ER_OK;
    for (int i = 0; i < depth && rc == STR_ER_OK; ++i) {
        str_er_params q = *p;
        q.stream = nullptr;                                   // every context gets its own stream
        rc = str_er_create(&q, &s->slots[(size_t)i].ctx);
        if (rc == STR_ER_OK && hipHostMalloc(reinterpret_cast<void **>(&s->slots[(size_t)i].pinned), s->slot_bytes, hipHostMallocDefault) != hipSuccess)
            rc = STR_ER_ENOMEM;
        if (rc == STR_ER_OK && (hipSetDevice(p->device) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&s->slots[(size_t)i].d_in), s->slot_bytes) != hipSuccess))
            rc = STR_ER_ENOMEM;
        for (int k = 0; k < str_er_stream::MAX_UPLOAD_STREAMS && rc == STR_ER_OK; ++k)
            if (k < s->upload_streams && (hipStreamCreateWithFlags(&s->slots[(size_t)i].copy[k], hipStreamNonBlocking) != hipSuccess ||
                                          hipEventCreateWithFlags(&s->slots[(size_t)i].landed[k], hipEventDisableTiming) != hipSuccess))
                rc = STR_ER_EHIP;
    }
    if (rc != STR_ER_OK) {
        for (auto &sl : s->slots) {
            if (sl.pinned) (void)hipHostFree(sl.pinned);
            if (sl.d_in) (void)hipFree(sl.d_in);
            for (hipEvent_t ev : sl.landed) if (ev) (void)hipEventDestroy(ev);
            for (hipStream_t cs : sl.copy) if (cs) (void)hipStreamDestroy(cs);
            if (sl.ctx) str_er_destroy(sl.ctx);
        }
        delete s;
        return rc;
    }
    for (int i = 0; i < depth; ++i) s->slots[(size_t)i].worker = std::thread(worker_main, s, i);
    *out = s;
    return STR_ER_OK;
} STREAM_GUARD(nullptr)

void str_er_stream_destroy(str_er_stream *s)
{
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv.notify_all();
    for (auto &sl : s->slots) if (sl.worker.joinable()) sl.worker.join();
    for (auto &sl : s->slots) {
        if (sl.result) str_er_result_free(sl.result);
        if (sl.pinned) (void)hipHostFree(sl.pinned);
        if (sl.d_in) (void)hipFree(sl.d_in);
        for (hipEvent_t ev : sl.landed) if (ev) (void)hipEventDestroy(ev);
        for (hipStream_t cs : sl.copy) if (cs) (void)hipStreamDestroy(cs);
        if (sl.ctx) str_er_destroy(sl.ctx);
    }
    delete s;
}

int32_t str_er_stream_depth(const str_er_stream *s) { return s ? (int32_t)s->slots.size() : 0; }

str_er_ctx *str_er_stream_context(str_er_stream *s, int32_t i)
{
    return (s && i >= 0 && (size_t)i < s->slots.size()) ? s->slots[(size_t)i].ctx : nullptr;
}

const char *str_er_stream_last_error(const str_er_stream *s) { return s ? s->err.c_str() : "null stream"; }

int str_er_stream_load_cascade(str_er_stream *s, int which, const char *path)
try {
    if (!s) return STR_ER_EINVAL;
    for (auto &sl : s->slots) {
        const int rc = str_er_load_cascade(sl.ctx, which, path);
        if (rc != STR_ER_OK) { s->err = str_er_last_error(sl.ctx); return rc; }
    }
    return STR_ER_OK;
} STREAM_GUARD(s)

int str_er_stream_acquire(str_er_stream *s, int32_t *slot, uint8_t **buffer, int64_t *capacity)
try {
    if (!s || !slot || !buffer) return STR_ER_EINVAL;
    std::lock_guard<std::mutex> lk(s->mu);
    for (size_t i = 0; i < s->slots.size(); ++i)
        if (!s->slots[i].busy) {
            s->slots[i].busy = true;
            *slot = (int32_t)i;
            *buffer = s->slots[i].pinned;
            if (capacity) *capacity = (int64_t)s->slot_bytes;
            return STR_ER_OK;
        }
    s->err = s->order.empty() ? "all staging buffers are acquired and none is submitted"
                              : "all staging buffers are in flight: collect a result with str_er_stream_next first";
    return STR_ER_ESTATE;
} STREAM_GUARD(s)

static int submit_impl(str_er_stream *s, int32_t slot, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch, int32_t n_frames,
                       uint32_t stages, uint64_t *ticket, bool nv12)
{
    if (!s || slot < 0 || (size_t)slot >= s->slots.size()) return STR_ER_EINVAL;
    str_er_stream::Slot &sl = s->slots[(size_t)slot];
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!sl.busy || sl.has_job || sl.done) { s->err = "slot was not acquired (or is already submitted)"; return STR_ER_ESTATE; }
        const int64_t row = nv12 ? (int64_t)w : (int64_t)w * 3, rows = nv12 ? (int64_t)h + h / 2 : (int64_t)h;
        if (w < 1 || h < 1 || n_frames < 1 || stride < row || (n_frames > 1 && frame_pitch < stride * rows) ||
            (uint64_t)(n_frames - 1) * (uint64_t)frame_pitch + (uint64_t)stride * (uint64_t)rows > (uint64_t)s->slot_bytes) {
            s->err = "frames do not fit the staging buffer";
            return STR_ER_EINVAL;
        }
        sl.w = w; sl.h = h; sl.stride = stride; sl.pitch = frame_pitch; sl.n_frames = n_frames; sl.stages = stages; sl.nv12 = nv12;
        sl.ticket = s->next_ticket++;
        sl.has_job = true;
        s->order.push_back(slot);
        if (ticket) *ticket = sl.ticket;
    }
    s->cv.notify_all();
    return STR_ER_OK;
}

int str_er_stream_submit(str_er_stream *s, int32_t slot, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch, int32_t n_frames,
                         uint32_t stages, uint64_t *ticket)
try {
    return submit_impl(s, slot, w, h, stride, frame_pitch, n_frames, stages, ticket, false);
} STREAM_GUARD(s)

int str_er_stream_submit_nv12(str_er_stream *s, int32_t slot, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch, int32_t n_frames,
                              uint32_t stages, uint64_t *ticket)
try {
    return submit_impl(s, slot, w, h, stride, frame_pitch, n_frames, stages, ticket, true);
} STREAM_GUARD(s)

int str_er_stream_submit_copy(str_er_stream *s, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch,
                              int32_t n_frames, uint32_t stages, uint64_t *ticket)
try {
    if (!s || !bgr || w < 1 || h < 1 || n_frames < 1 || stride < (int64_t)w * 3) return STR_ER_EINVAL;
    int32_t  slot = -1;
    uint8_t *buf = nullptr;
    int rc = str_er_stream_acquire(s, &slot, &buf, nullptr);
    if (rc != STR_ER_OK) return rc;
    const size_t row = (size_t)w * 3, fb = row * (size_t)h;
    if (fb * (size_t)n_frames > s->slot_bytes) {
        std::lock_guard<std::mutex> lk(s->mu);
        s->slots[(size_t)slot].busy = false;
        s->err = "frames do not fit the staging buffer";
        return STR_ER_ECAPACITY;
    }
    for (int f = 0; f < n_frames; ++f)
        for (int y = 0; y < h; ++y)
            std::memcpy(buf + (size_t)f * fb + (size_t)y * row, bgr + (size_t)f * (size_t)frame_pitch + (size_t)y * (size_t)stride, row);
    return str_er_stream_submit(s, slot, w, h, (int64_t)row, (int64_t)fb, n_frames, stages, ticket);
} STREAM_GUARD(s)

int str_er_stream_next(str_er_stream *s, str_er_result **out, uint64_t *ticket)
try {
    if (!s || !out) return STR_ER_EINVAL;
    *out = nullptr;
    std::unique_lock<std::mutex> lk(s->mu);
    if (s->order.empty()) { s->err = "nothing submitted"; return STR_ER_ESTATE; }
    const int idx = s->order.front();
    str_er_stream::Slot &sl = s->slots[(size_t)idx];
    s->cv.wait(lk, [&] { return sl.done; });
    s->order.pop_front();
    const int rc = sl.rc;
    *out = sl.result;
    if (ticket) *ticket = sl.ticket;
    if (rc != STR_ER_OK) s->err = sl.err;
    sl.result = nullptr; sl.done = false; sl.busy = false;
    lk.unlock();
    s->cv.notify_all();
    return rc;
} STREAM_GUARD(s)

int32_t str_er_stream_pending(str_er_stream *s)
{
    if (!s) return 0;
    std::lock_guard<std::mutex> lk(s->mu);
    return (int32_t)s->order.size();
}

} // extern "C"
