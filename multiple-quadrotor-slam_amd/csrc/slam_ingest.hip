// Frame ingest of the device-resident loop (BASELINE configs[4]): the reference reads every frame INSIDE its loop
// (Work/SLAM/application/own/slam2.py:1209-1213: cv2.imread per iteration); here a frame arrives in host memory and goes to the device
// on a stream of its own while the loop's kernels work on the frames before it.
//
// A ring of device images owned by the handle, and ONE worker thread of the library: `mqs_slam_upload` only posts a job (a mutex and
// a notification -- the calling thread goes straight on to mqs_slam_track, which blocks for the frame's result); the worker stages a
// pageable source through the slot's pinned staging buffer, enqueues the copy on the upload stream and records the slot's event.
// `mqs_slam_wait_upload` hands out the slot's device pointer; the LOOP'S STREAM waits for that event (hipStreamWaitEvent: a device-side wait,
// the host does not block) when one of its kernels is about to read the image (mqs_slam_ingest_main_wait) -- the caller passes the pointer to
// mqs_slam_start / mqs_slam_track like any other image.
// (From the interpreter the same thing -- a thread issuing torch copies -- cost 30 % of the plain loop's frame rate: the two threads
// hand the interpreter lock back and forth.)
#include "mqs_common.h"
#include "slam_state.h"
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>

struct mqs_slam_ingest {
    int slots;
    size_t bytes;                       // W * H
    uint8_t *dev;                       // [slots][bytes]
    uint8_t *stage;                     // pinned [slots][bytes]
    hipStream_t up;
    hipEvent_t ev[MQS_SLAM_INGEST_MAX_SLOTS];
    unsigned long long posted[MQS_SLAM_INGEST_MAX_SLOTS], done[MQS_SLAM_INGEST_MAX_SLOTS];
    bool main_pending[MQS_SLAM_INGEST_MAX_SLOTS];     // mqs_slam_wait_upload named the slot, the loop's stream has not waited for its upload yet (it does when it reads the image)
    struct Job { int slot; const uint8_t *host; int pinned; };
    std::deque<Job> jobs;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::thread worker;
    bool stop;
    hipError_t error;
    int device;
};

namespace {

void ingest_worker(mqs_slam_ingest *g)
{
    (void)hipSetDevice(g->device);
    for (;;) {
        mqs_slam_ingest::Job j;
        {
            std::unique_lock<std::mutex> lk(g->m);
            g->cv_job.wait(lk, [g] { return g->stop || !g->jobs.empty(); });
            if (g->jobs.empty()) return;                                   // (stop, nothing left)
            j = g->jobs.front();
            g->jobs.pop_front();
        }
        const uint8_t *src = j.host;
        if (!j.pinned) {
            memcpy(g->stage + (size_t)j.slot * g->bytes, j.host, g->bytes);
            src = g->stage + (size_t)j.slot * g->bytes;
        }
        hipError_t e = hipMemcpyAsync(g->dev + (size_t)j.slot * g->bytes, src, g->bytes, hipMemcpyHostToDevice, g->up);
        if (e == hipSuccess) e = hipEventRecord(g->ev[j.slot], g->up);
        {
            std::lock_guard<std::mutex> lk(g->m);
            if (e != hipSuccess && g->error == hipSuccess) g->error = e;
            g->done[j.slot] += 1;
        }
        g->cv_done.notify_all();
    }
}

}  // namespace

void mqs_slam_ingest_release(mqs_slam *s)
{
    mqs_slam_ingest *g = s ? s->ingest : nullptr;
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->stop = true;
    }
    g->cv_job.notify_all();
    if (g->worker.joinable()) g->worker.join();
    (void)hipStreamSynchronize(g->up);
    for (int k = 0; k < g->slots; ++k) (void)hipEventDestroy(g->ev[k]);
    (void)hipStreamDestroy(g->up);
    if (g->dev) (void)hipFree(g->dev);
    if (g->stage) (void)hipHostFree(g->stage);
    delete g;
    s->ingest = nullptr;
}

bool mqs_slam_ingest_slot(mqs_slam *s, int slot, const uint8_t **image_dev, hipEvent_t *uploaded)
{
    mqs_slam_ingest *g = s ? s->ingest : nullptr;
    if (!g || slot < 0 || slot >= g->slots) return false;
    {
        std::unique_lock<std::mutex> lk(g->m);
        if (g->posted[slot] == 0) return false;
        g->cv_done.wait(lk, [g, slot] { return g->done[slot] == g->posted[slot]; });
        if (g->error != hipSuccess) return false;
    }
    *image_dev = g->dev + (size_t)slot * g->bytes;
    *uploaded = g->ev[slot];
    return true;
}

// The loop's stream is about to read `img`: if it is a ring slot the caller took with mqs_slam_wait_upload and the stream has not waited for its
// upload yet, it does now.  (The wait used to be enqueued by mqs_slam_wait_upload itself -- a cross-queue barrier packet on the loop's stream per
// frame, between one frame's decision and the next one's hypotheses, although with the pair tracked ahead nothing on that stream reads the image:
// the side stream does, and waits for the upload itself.  ~10 us of a frame's ~100.)
int mqs_slam_ingest_main_wait(mqs_slam *s, const uint8_t *img)
{
    mqs_slam_ingest *g = s ? s->ingest : nullptr;
    if (!g || img < g->dev || img >= g->dev + (size_t)g->slots * g->bytes) return MQS_OK;
    const int slot = (int)((size_t)(img - g->dev) / g->bytes);
    if (!g->main_pending[slot]) return MQS_OK;
    MQS_HIP_CHECK(hipStreamWaitEvent(s->stream, g->ev[slot], 0));
    g->main_pending[slot] = false;
    return MQS_OK;
}

extern "C" {

int mqs_slam_ingest_enable(mqs_slam *s, int slots)
{
    MQS_ARG_CHECK(s != nullptr && slots >= 2 && slots <= MQS_SLAM_INGEST_MAX_SLOTS, "handle; 2 <= slots <= MQS_SLAM_INGEST_MAX_SLOTS");
    MQS_ARG_CHECK(s->ingest == nullptr, "once per handle");
    MQS_HIP_CHECK(hipSetDevice(s->device));
    mqs_slam_ingest *g = new (std::nothrow) mqs_slam_ingest();
    if (!g) { mqs_set_error("out of host memory"); return MQS_E_NOMEM; }
    g->slots = slots; g->bytes = (size_t)s->p.W * s->p.H; g->dev = nullptr; g->stage = nullptr; g->stop = false; g->error = hipSuccess;
    g->device = s->device;
    for (int k = 0; k < MQS_SLAM_INGEST_MAX_SLOTS; ++k) { g->posted[k] = 0; g->done[k] = 0; g->main_pending[k] = false; }
    hipError_t e = hipMalloc((void **)&g->dev, (size_t)slots * g->bytes);
    if (e == hipSuccess) e = hipHostMalloc((void **)&g->stage, (size_t)slots * g->bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        if (g->dev) (void)hipFree(g->dev);
        delete g;
        mqs_set_error("mqs_slam_ingest_enable: %s", hipGetErrorString(e));
        return MQS_E_NOMEM;
    }
    // (default priority.  The LOWEST priority -- a hardware queue that is certainly neither the loop's nor its side stream's -- was tried: inside
    // bench.py the legs with upload lost 10-13 %: the queue's packets, the upload events among them, wait while the other queues have work, and
    // the pipelined loop always has.  Sharing a queue with the loop's stream costs an upload at most a frame's kernels, and it is two frames ahead.)
    e = hipStreamCreateWithFlags(&g->up, hipStreamNonBlocking);
    int made = 0;
    for (; e == hipSuccess && made < slots; ++made) e = hipEventCreateWithFlags(&g->ev[made], hipEventDisableTiming);
    if (e != hipSuccess) {
        for (int k = 0; k < made - 1; ++k) (void)hipEventDestroy(g->ev[k]);
        (void)hipFree(g->dev); (void)hipHostFree(g->stage);
        delete g;
        mqs_set_error("mqs_slam_ingest_enable: %s", hipGetErrorString(e));
        return MQS_E_HIP;
    }
    g->worker = std::thread(ingest_worker, g);
    s->ingest = g;
    return MQS_OK;
}

int mqs_slam_upload(mqs_slam *s, int slot, const uint8_t *host_img, int pinned)
{
    MQS_ARG_CHECK(s != nullptr && s->ingest != nullptr, "mqs_slam_ingest_enable first");
    mqs_slam_ingest *g = s->ingest;
    MQS_ARG_CHECK(slot >= 0 && slot < g->slots && host_img != nullptr, "0 <= slot < slots; image must not be null");
    // a pyramid launch that read the slot's last image on the side stream (mqs_slam_prepare_next) has to be over before the slot is written again
    // (it is: the frame it was prepared for has been tracked or rejected since -- this makes it certain)
    {
        const uint8_t *img = g->dev + (size_t)slot * g->bytes;
        for (int k = 0; k < 2; ++k)
            if (s->prep[k].has_event && (s->prep[k].prev == img || s->prep[k].next == img)) {
                MQS_HIP_CHECK(hipEventSynchronize(s->prep[k].done));
                if (s->prep[k].valid) s->prep[k].valid = false;      // (prepared for a pair that never came)
                s->prep[k].prev = nullptr; s->prep[k].next = nullptr;
            }
    }
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->posted[slot] += 1;
        g->main_pending[slot] = false;                 // (a new image: mqs_slam_wait_upload names it again)
        g->jobs.push_back({slot, host_img, pinned});
    }
    g->cv_job.notify_one();
    return MQS_OK;
}

int mqs_slam_wait_upload(mqs_slam *s, int slot, const uint8_t **image_dev)
{
    MQS_ARG_CHECK(s != nullptr && s->ingest != nullptr, "mqs_slam_ingest_enable first");
    mqs_slam_ingest *g = s->ingest;
    MQS_ARG_CHECK(slot >= 0 && slot < g->slots && image_dev != nullptr, "0 <= slot < slots; image_dev must not be null");
    hipError_t err;
    {
        std::unique_lock<std::mutex> lk(g->m);
        MQS_ARG_CHECK(g->posted[slot] > 0, "nothing was uploaded into this slot");
        g->cv_done.wait(lk, [g, slot] { return g->done[slot] == g->posted[slot]; });       // (the worker has ENQUEUED the copy; usually long ago)
        err = g->error;
    }
    if (err != hipSuccess) { mqs_set_error("mqs_slam_upload: %s", hipGetErrorString(err)); return MQS_E_HIP; }
    g->main_pending[slot] = true;                      // (the loop's stream waits when it is about to read the image: mqs_slam_ingest_main_wait)
    *image_dev = g->dev + (size_t)slot * g->bytes;
    return MQS_OK;
}

}  // extern "C"
