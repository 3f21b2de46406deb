// rb_pool.cpp -- single-process multi-GPU form of the classifier: one engine and one host thread per
// device, every filter replicated in each device's HBM, batches cut into contiguous read slices, no collective.
// The reference's scaling model is N classify threads popping one queue (src/main/adaptive_sampling.hpp:745-751);
// here the N workers are GPUs.  Micro-batches are not split (latency): they go to one device, round-robin.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "rb_internal.h"

#include "rb_workq.h"

namespace {

// One engine, one host thread (rbq::Worker) and one FIFO of tasks per device.  Callers on different host threads only meet in
// the short critical section that picks a worker (rbq::Dispatcher): their micro-batches run on different engines at the same
// time, which is the reference's N classification threads behind one queue (src/main/adaptive_sampling.hpp:745-751).
struct Device {
    int device = 0;
    std::vector<rb_dibf *> filters;  // owned replicas, deplete first
    rb_engine *engine = nullptr;
    std::atomic<uint64_t> busy_ns{0}, reads{0}, calls{0};  // time inside rb_classify_batch, reads and parts served (rb_pool_get_stats)
    rbq::Worker worker{[] { return std::string(rb_last_error()); }};
};

}  // namespace

struct rb_pool {
    std::vector<Device *> workers;
    std::unique_ptr<rbq::Dispatcher> dispatcher;
    size_t nd = 0, nt = 0;
    size_t min_split_reads = 4096;  // per-device slice below which splitting does not pay
    bool serialize = false;         // diagnostic: one call at a time (what round 2 did)
    std::mutex cfg_mu;              // guards the two settings above
    std::mutex call_mu;             // held for a whole call only when `serialize` is set
    void start()
    {
        std::vector<rbq::Worker *> ws;
        for (Device *d : workers) {
            d->worker.start();
            ws.push_back(&d->worker);
        }
        dispatcher.reset(new rbq::Dispatcher(ws));
    }
};

extern "C" void rb_pool_destroy(rb_pool *p);

// One resident filter -> a replica on each of devices[from .. n): all peers at once, one stream per destination, each copy on its
// own xGMI link.  A device whose copy cannot be started or does not finish gets the filter by the next rung of the ladder:
// the runtime's staged copy (rb_dibf_clone_to), then -- when the filter came from a file -- its own stream of that file.
static int replicate(rb_pool *p, rb_dibf *first, const int *devices, size_t n_devices, size_t from, const char *path, double *copy_s)
{
    std::vector<void *> streams(n_devices, nullptr);
    std::vector<bool> fallback(n_devices, false);
    const auto t0 = std::chrono::steady_clock::now();
    // Fault injection, compiled into the TESTING build of the library only (make testing: libreadbouncer_amd_testing.so,
    // -DRB_TESTING): RB_POOL_TEST_FAIL_CLONE=start|finish makes every device-to-device copy "fail" at that step, so that the
    // ladder below is walked on a box where the copies themselves cannot fail (tests/test_gpu_parity.py runs that build in a
    // child process).  The product library has no such switch.
#ifdef RB_TESTING
    const char *inject = std::getenv("RB_POOL_TEST_FAIL_CLONE");
    const bool fail_start = inject && std::strcmp(inject, "start") == 0;
    const bool fail_finish = inject && std::strcmp(inject, "finish") == 0;
    // RB_POOL_TEST_THREAD_PER_WORKER=1: workers that share a GPU are started from threads of their own as if they sat on different GPUs
    // (the side-by-side start below is otherwise never walked on a one-GPU box)
    const char *tpw = std::getenv("RB_POOL_TEST_THREAD_PER_WORKER");
    const bool thread_per_worker = tpw && tpw[0] == '1';
#else
    const bool fail_start = false, fail_finish = false, thread_per_worker = false;
#endif
    // Allocating a replica includes the placement trial of its table (1-2 s for a table of 1 GiB and more): the destinations do that side
    // by side, one thread per distinct GPU (workers that share a GPU take turns in their thread: trials probe the device they run on).
    {
        std::vector<rb_dibf *> made(n_devices, nullptr);
        std::vector<int> distinct;
        for (size_t d = from; d < n_devices; ++d)
            if (std::find(distinct.begin(), distinct.end(), devices[d]) == distinct.end()) distinct.push_back(devices[d]);
        auto start_one = [&](size_t d) {
            int peer = 0;
            if (fail_start || rb_dibf_clone_start(first, devices[d], &made[d], &streams[d], &peer) != RB_OK) made[d] = nullptr;
        };
        auto start_on = [&](int dev) {
            for (size_t d = from; d < n_devices; ++d)
                if (devices[d] == dev) start_one(d);
        };
        if (thread_per_worker) {
            std::vector<std::thread> th;
            for (size_t d = from; d < n_devices; ++d) th.emplace_back(start_one, d);
            for (std::thread &t : th) t.join();
        } else if (distinct.size() <= 1) {
            for (int dev : distinct) start_on(dev);
        } else {
            std::vector<std::thread> th;
            for (int dev : distinct) th.emplace_back(start_on, dev);
            for (std::thread &t : th) t.join();
        }
        for (size_t d = from; d < n_devices; ++d) {
            if (made[d]) p->workers[d]->filters.push_back(made[d]);
            else fallback[d] = true;
        }
    }
    for (size_t d = from; d < n_devices; ++d) {
        if (fallback[d]) continue;
        if (rb_dibf_clone_finish(streams[d]) != RB_OK || fail_finish) {
            rb_dibf_free(p->workers[d]->filters.back());
            p->workers[d]->filters.pop_back();
            fallback[d] = true;
        }
    }
    *copy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (size_t d = from; d < n_devices; ++d) {
        if (!fallback[d]) continue;
        rb_dibf *f = nullptr;
        // no device-to-device path: this device streams the file itself, or (no file) takes the runtime's synchronous copy
        const int rc = path ? rb_dibf_open(devices[d], path, &f) : rb_dibf_clone_to(first, devices[d], &f);
        if (rc != RB_OK) return rc;
        p->workers[d]->filters.push_back(f);
    }
    return RB_OK;
}

extern "C" {

void rb_pool_destroy(rb_pool *p)
{
    if (!p) return;
    for (Device *w : p->workers) {
        w->worker.stop();  // queued tasks are still run
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
        Device *w = new (std::nothrow) Device();
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
    }
    p->start();
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
        Device *w = new (std::nothrow) Device();
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
        rc = replicate(p, first, devices, n_devices, 1, path, &copy_s);
        if (rc != RB_OK) { rb_pool_destroy(p); return rc; }
    }
    for (Device *w : p->workers) {
        const int rc = rb_engine_create(w->device, w->filters.data(), n_deplete, w->filters.data() + n_deplete, n_target,
                                        &w->engine);
        if (rc != RB_OK) { rb_pool_destroy(p); return rc; }
    }
    p->start();
    if (replication_seconds) *replication_seconds = copy_s;
    *out = p;
    return RB_OK;
}

int rb_pool_create_from_device(const int *devices, size_t n_devices, rb_dibf *const *deplete, size_t n_deplete, rb_dibf *const *target,
                               size_t n_target, rb_pool **out, double *replication_seconds)
{
    if (!out || !devices || n_devices == 0) return rb::fail(RB_ERR_INVALID_ARG, "no devices");
    if (n_deplete + n_target == 0) return rb::fail(RB_ERR_NULL_FILTER, "No IBF provided to classify the read!");
    rb_pool *p = new (std::nothrow) rb_pool();
    if (!p) return rb::fail(RB_ERR_NOMEM, "alloc");
    p->nd = n_deplete;
    p->nt = n_target;
    for (size_t d = 0; d < n_devices; ++d) {
        Device *w = new (std::nothrow) Device();
        if (!w) { rb_pool_destroy(p); return rb::fail(RB_ERR_NOMEM, "alloc"); }
        w->device = devices[d];
        p->workers.push_back(w);
    }
    double copy_s = 0.0;
    for (size_t i = 0; i < n_deplete + n_target; ++i) {
        rb_dibf *src = i < n_deplete ? deplete[i] : target[i - n_deplete];
        // every device gets a replica of its own, the source's device included: the pool owns what it classifies against
        const int rc = src ? replicate(p, src, devices, n_devices, 0, nullptr, &copy_s) : rb::fail(RB_ERR_INVALID_ARG, "null filter");
        if (rc != RB_OK) { rb_pool_destroy(p); return rc; }
    }
    for (Device *w : p->workers) {
        const int rc = rb_engine_create(w->device, w->filters.data(), n_deplete, w->filters.data() + n_deplete, n_target, &w->engine);
        if (rc != RB_OK) { rb_pool_destroy(p); return rc; }
    }
    p->start();
    if (replication_seconds) *replication_seconds = copy_s;
    *out = p;
    return RB_OK;
}

int rb_pool_get_stats(rb_pool *p, size_t n, int *devices, double *busy_seconds, uint64_t *reads, uint64_t *calls, int reset)
{
    if (!p) return rb::fail(RB_ERR_INVALID_ARG, "null pool");
    for (size_t i = 0; i < n && i < p->workers.size(); ++i) {
        Device *w = p->workers[i];
        if (devices) devices[i] = w->device;
        if (busy_seconds) busy_seconds[i] = (double)w->busy_ns.load() * 1e-9;
        if (reads) reads[i] = w->reads.load();
        if (calls) calls[i] = w->calls.load();
        if (reset) { w->busy_ns = 0; w->reads = 0; w->calls = 0; }
    }
    return RB_OK;
}

size_t rb_pool_size(const rb_pool *p) { return p ? p->workers.size() : 0; }

int rb_pool_set_timing(rb_pool *p, int enabled)
{
    if (!p) return rb::fail(RB_ERR_INVALID_ARG, "null pool");
    for (Device *w : p->workers) {
        const int rc = rb_engine_set_timing(w->engine, enabled);
        if (rc != RB_OK) return rc;
    }
    return RB_OK;
}

int rb_pool_kernel_time(rb_pool *p, size_t n, double *total_ms, uint64_t *n_launches)
{
    if (!p) return rb::fail(RB_ERR_INVALID_ARG, "null pool");
    for (size_t i = 0; i < n && i < p->workers.size(); ++i) {
        double ms = 0.0;
        uint64_t calls = 0;
        const int rc = rb_engine_kernel_time(p->workers[i]->engine, &ms, &calls);
        if (rc != RB_OK) return rc;
        if (total_ms) total_ms[i] = ms;
        if (n_launches) n_launches[i] = calls;
    }
    return RB_OK;
}

int rb_pool_set_min_split(rb_pool *p, size_t reads_per_device)
{
    if (!p) return rb::fail(RB_ERR_INVALID_ARG, "null pool");
    std::lock_guard<std::mutex> lock(p->cfg_mu);
    p->min_split_reads = reads_per_device ? reads_per_device : 1;
    return RB_OK;
}

int rb_pool_set_serialize(rb_pool *p, int enabled)
{
    if (!p) return rb::fail(RB_ERR_INVALID_ARG, "null pool");
    std::lock_guard<std::mutex> lock(p->cfg_mu);
    p->serialize = enabled != 0;
    return RB_OK;
}

int rb_pool_classify_batch(rb_pool *p, const char *seqs, const uint64_t *offsets, const uint32_t *lens, size_t n_reads,
                           double error_rate, double significance, int mode, uint16_t *out_maxcount,
                           int32_t *out_best_target, uint8_t *out_decision, uint8_t *out_status)
{
    if (!p) return rb::fail(RB_ERR_INVALID_ARG, "null pool");
    if (n_reads == 0) return RB_OK;
    size_t min_split;
    bool serialize;
    {
        std::lock_guard<std::mutex> lock(p->cfg_mu);
        min_split = p->min_split_reads;
        serialize = p->serialize;
    }
    std::unique_lock<std::mutex> whole_call(p->call_mu, std::defer_lock);
    if (serialize) whole_call.lock();
    const size_t nf = p->nd + p->nt;
    // an unsplit micro-batch goes to the least loaded worker, so K calling threads keep K engines busy; a large batch is cut
    // into contiguous slices of ceil(n/parts) reads over consecutive workers
    size_t parts = std::min(p->workers.size(), std::max<size_t>(1, n_reads / min_split));
    const size_t per = (n_reads + parts - 1) / parts;
    parts = (n_reads + per - 1) / per;  // no empty slices
    rbq::Job job;
    p->dispatcher->dispatch(parts, job, [&](size_t k, size_t w) -> std::function<int()> {
        const size_t b = k * per, e = std::min(n_reads, b + per);
        Device *dev = p->workers[w];
        rb_engine *eng = dev->engine;
        return [=] {
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = rb_classify_batch(eng, seqs, offsets + b, lens + b, e - b, error_rate, significance, mode,
                                             out_maxcount ? out_maxcount + b * nf : nullptr, out_best_target ? out_best_target + b : nullptr,
                                             out_decision ? out_decision + b : nullptr, out_status ? out_status + b : nullptr);
            dev->busy_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            dev->reads += e - b;
            dev->calls += 1;
            return rc;
        };
    });
    job.wait();
    if (job.rc != RB_OK) return rb::fail(job.rc, job.error);
    return RB_OK;
}

}  // extern "C"
