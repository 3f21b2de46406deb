// rb_pool.cpp -- single-process multi-GPU form of the classifier: one engine and one host thread per
// device, every filter replicated in each device's HBM, batches cut into contiguous read slices, no collective.
// The reference's scaling model is N classify threads popping one queue (src/main/adaptive_sampling.hpp:745-751);
// here the N workers are GPUs.  Micro-batches are not split (latency): they go to one device, round-robin.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "rb_internal.h"

namespace {

// one rb_pool_classify_batch call: its parts run on one or several workers; the caller sleeps on `cv` until all are done
struct Job {
    std::mutex mu;
    std::condition_variable cv;
    size_t pending = 0;
    int rc = RB_OK;
    std::string error;
};

struct Task {
    std::function<int()> fn;
    Job *job;
};

// One engine, one host thread, one FIFO of tasks per device.  Callers on different host threads only meet in the short
// critical section that picks a worker: their micro-batches run on different engines at the same time, which is the
// reference's N classification threads behind one queue (src/main/adaptive_sampling.hpp:745-751).
struct Worker {
    int device = 0;
    std::vector<rb_dibf *> filters;  // owned replicas, deplete first
    rb_engine *engine = nullptr;
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Task> queue;
    size_t load = 0;  // queued + running tasks (guarded by mu; read under the pool's pick lock as a hint)
    bool stop = false;

    void loop()
    {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return !queue.empty() || stop; });
                if (queue.empty()) return;  // stop requested and nothing left to run
                t = std::move(queue.front());
                queue.pop_front();
            }
            const int rc = t.fn();
            const std::string err = rc == RB_OK ? std::string() : std::string(rb_last_error());
            {
                std::lock_guard<std::mutex> lock(mu);
                --load;
            }
            {
                std::lock_guard<std::mutex> lock(t.job->mu);
                if (rc != RB_OK && t.job->rc == RB_OK) { t.job->rc = rc; t.job->error = err; }
                --t.job->pending;
                // notify under the lock: the Job lives on the caller's stack and goes away as soon as pending hits zero
                t.job->cv.notify_all();
            }
        }
    }
    void submit(Task t)
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            queue.push_back(std::move(t));
            ++load;
        }
        cv.notify_one();
    }
    size_t current_load()
    {
        std::lock_guard<std::mutex> lock(mu);
        return load;
    }
};

}  // namespace

struct rb_pool {
    std::vector<Worker *> workers;
    size_t nd = 0, nt = 0;
    size_t next = 0;              // round-robin cursor: breaks ties between equally loaded workers
    size_t min_split_reads = 4096;  // per-device slice below which splitting does not pay
    bool serialize = false;       // diagnostic: one call at a time (what round 2 did)
    std::mutex pick_mu;           // worker selection + enqueueing of one call's parts (short)
    std::mutex call_mu;           // held for a whole call only when `serialize` is set
};

extern "C" {

void rb_pool_destroy(rb_pool *p)
{
    if (!p) return;
    for (Worker *w : p->workers) {
        if (w->thread.joinable()) {
            {
                std::lock_guard<std::mutex> lock(w->mu);
                w->stop = true;
            }
            w->cv.notify_all();
            w->thread.join();
        }
        if (w->engine) rb_engine_destroy(w->engine);
        for (rb_dibf *f : w->filters) rb_dibf_free(f);
        delete w;
    }
    delete p;
}

int rb_pool_create(const int *devices, size_t n_devices, const rb_ibf *const *deplete, size_t n_deplete,
                   const rb_ibf *const *target, size_t n_target, rb_pool **out)
{
    if (!out || !devices || n_devices == 0) return rb::fail(RB_ERR_INVALID_ARG, "no devices");
    if (n_deplete + n_target == 0) return rb::fail(RB_ERR_NULL_FILTER, "No IBF provided to classify the read!");
    rb_pool *p = new (std::nothrow) rb_pool();
    if (!p) return rb::fail(RB_ERR_NOMEM, "alloc");
    p->nd = n_deplete;
    p->nt = n_target;
    for (size_t d = 0; d < n_devices; ++d) {
        Worker *w = new (std::nothrow) Worker();
        if (!w) { rb_pool_destroy(p); return rb::fail(RB_ERR_NOMEM, "alloc"); }
        p->workers.push_back(w);
        w->device = devices[d];
        for (size_t i = 0; i < n_deplete + n_target; ++i) {
            const rb_ibf *img = i < n_deplete ? deplete[i] : target[i - n_deplete];
            rb_dibf *f = nullptr;
            const int rc = img ? rb_dibf_upload(w->device, img, &f) : rb::fail(RB_ERR_INVALID_ARG, "null filter image");
            if (rc != RB_OK) { rb_pool_destroy(p); return rc; }
            w->filters.push_back(f);
        }
        const int rc = rb_engine_create(w->device, w->filters.data(), n_deplete, w->filters.data() + n_deplete, n_target,
                                        &w->engine);
        if (rc != RB_OK) { rb_pool_destroy(p); return rc; }
        w->thread = std::thread([w] { w->loop(); });
    }
    *out = p;
    return RB_OK;
}

int rb_pool_create_from_files(const int *devices, size_t n_devices, const char *const *deplete_paths, size_t n_deplete,
                              const char *const *target_paths, size_t n_target, rb_pool **out, double *replication_seconds)
{
    if (!out || !devices || n_devices == 0) return rb::fail(RB_ERR_INVALID_ARG, "no devices");
    if (n_deplete + n_target == 0) return rb::fail(RB_ERR_NULL_FILTER, "No IBF provided to classify the read!");
    rb_pool *p = new (std::nothrow) rb_pool();
    if (!p) return rb::fail(RB_ERR_NOMEM, "alloc");
    p->nd = n_deplete;
    p->nt = n_target;
    for (size_t d = 0; d < n_devices; ++d) {
        Worker *w = new (std::nothrow) Worker();
        if (!w) { rb_pool_destroy(p); return rb::fail(RB_ERR_NOMEM, "alloc"); }
        w->device = devices[d];
        p->workers.push_back(w);
    }
    double copy_s = 0.0;
    for (size_t i = 0; i < n_deplete + n_target; ++i) {
        const char *path = i < n_deplete ? deplete_paths[i] : target_paths[i - n_deplete];
        rb_dibf *first = nullptr;
        int rc = path ? rb_dibf_open(devices[0], path, &first) : rb::fail(RB_ERR_INVALID_ARG, "null filter path");
        if (rc != RB_OK) { rb_pool_destroy(p); return rc; }
        p->workers[0]->filters.push_back(first);
        // all peers at once: one stream per destination, each copy on its own xGMI link
        std::vector<void *> streams(n_devices, nullptr);
        std::vector<bool> need_file(n_devices, false);
        const auto t0 = std::chrono::steady_clock::now();
        // test hook (tests/test_gpu_parity.py): RB_POOL_TEST_FAIL_CLONE=start|finish makes every device-to-device copy
        // "fail" at that step, so that the ladder below -- peer copy, staged copy, own file stream -- is walked on a box
        // where the copies themselves cannot fail
        const char *inject = std::getenv("RB_POOL_TEST_FAIL_CLONE");
        const bool fail_start = inject && std::strcmp(inject, "start") == 0;
        const bool fail_finish = inject && std::strcmp(inject, "finish") == 0;
        for (size_t d = 1; d < n_devices; ++d) {
            rb_dibf *f = nullptr;
            int peer = 0;
            if (!fail_start && rb_dibf_clone_start(first, devices[d], &f, &streams[d], &peer) == RB_OK) p->workers[d]->filters.push_back(f);
            else need_file[d] = true;
        }
        for (size_t d = 1; d < n_devices; ++d) {
            if (need_file[d]) continue;
            if (rb_dibf_clone_finish(streams[d]) != RB_OK || fail_finish) {
                rb_dibf_free(p->workers[d]->filters.back());
                p->workers[d]->filters.pop_back();
                need_file[d] = true;
            }
        }
        copy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (size_t d = 1; d < n_devices; ++d) {  // no device-to-device path: this device streams the file itself
            if (!need_file[d]) continue;
            rb_dibf *f = nullptr;
            rc = rb_dibf_open(devices[d], path, &f);
            if (rc != RB_OK) { rb_pool_destroy(p); return rc; }
            p->workers[d]->filters.push_back(f);
        }
    }
    for (Worker *w : p->workers) {
        const int rc = rb_engine_create(w->device, w->filters.data(), n_deplete, w->filters.data() + n_deplete, n_target,
                                        &w->engine);
        if (rc != RB_OK) { rb_pool_destroy(p); return rc; }
        w->thread = std::thread([w] { w->loop(); });
    }
    if (replication_seconds) *replication_seconds = copy_s;
    *out = p;
    return RB_OK;
}

size_t rb_pool_size(const rb_pool *p) { return p ? p->workers.size() : 0; }

int rb_pool_set_min_split(rb_pool *p, size_t reads_per_device)
{
    if (!p) return rb::fail(RB_ERR_INVALID_ARG, "null pool");
    p->min_split_reads = reads_per_device ? reads_per_device : 1;
    return RB_OK;
}

int rb_pool_set_serialize(rb_pool *p, int enabled)
{
    if (!p) return rb::fail(RB_ERR_INVALID_ARG, "null pool");
    std::lock_guard<std::mutex> lock(p->pick_mu);
    p->serialize = enabled != 0;
    return RB_OK;
}

int rb_pool_classify_batch(rb_pool *p, const char *seqs, const uint64_t *offsets, const uint32_t *lens, size_t n_reads,
                           double error_rate, double significance, int mode, uint16_t *out_maxcount,
                           int32_t *out_best_target, uint8_t *out_decision, uint8_t *out_status)
{
    if (!p) return rb::fail(RB_ERR_INVALID_ARG, "null pool");
    if (n_reads == 0) return RB_OK;
    std::unique_lock<std::mutex> whole_call(p->call_mu, std::defer_lock);
    const size_t nf = p->nd + p->nt;
    Job job;
    {
        // Callers only meet here: pick the workers, queue the parts, go.  An unsplit micro-batch goes to the least loaded
        // worker (ties: round-robin), so K calling threads keep K engines busy; a large batch is cut into contiguous slices
        // over all workers, each slice behind whatever that worker still has queued (FIFO per worker).
        std::unique_lock<std::mutex> pick(p->pick_mu);
        if (p->serialize) {
            pick.unlock();
            whole_call.lock();
            pick.lock();
        }
        const size_t nw = p->workers.size();
        const size_t parts = std::min(nw, std::max<size_t>(1, n_reads / p->min_split_reads));
        const size_t per = (n_reads + parts - 1) / parts;  // contiguous slices of ceil(n/parts) reads
        size_t first = p->next % nw;
        if (parts == 1) {
            size_t best_load = ~(size_t)0;
            for (size_t k = 0; k < nw; ++k) {
                const size_t i = (p->next + k) % nw;
                const size_t l = p->workers[i]->current_load();
                if (l < best_load) { best_load = l; first = i; }
            }
            p->next = (first + 1) % nw;
        }
        for (size_t k = 0; k < parts; ++k)  // count first: no worker sees `job` before the first submit
            if (std::min(n_reads, k * per) < n_reads) ++job.pending;
        for (size_t k = 0; k < parts; ++k) {
            const size_t b = std::min(n_reads, k * per), e = std::min(n_reads, b + per);
            if (b == e) continue;
            Worker *w = p->workers[(first + k) % nw];
            w->submit(Task{[=] {
                return rb_classify_batch(w->engine, seqs, offsets + b, lens + b, e - b, error_rate, significance, mode,
                                         out_maxcount ? out_maxcount + b * nf : nullptr,
                                         out_best_target ? out_best_target + b : nullptr,
                                         out_decision ? out_decision + b : nullptr, out_status ? out_status + b : nullptr);
            }, &job});
        }
    }
    {
        std::unique_lock<std::mutex> lock(job.mu);
        job.cv.wait(lock, [&] { return job.pending == 0; });
    }
    if (job.rc != RB_OK) return rb::fail(job.rc, job.error);
    return RB_OK;
}

}  // extern "C"
