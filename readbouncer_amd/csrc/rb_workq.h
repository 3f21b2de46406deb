// rb_workq.h -- the host-thread machinery of the one-process pool (rb_pool.cpp), free of anything GPU so that it can be
// exercised on a CPU under ThreadSanitizer (tests/cpp/test_workq.cpp): per-worker FIFOs of tasks, jobs that callers sleep
// on, and the pick of the least loaded worker.  The reference's counterpart is N classification threads popping one
// SafeQueue (src/main/adaptive_sampling.hpp:745-751, src/util/SafeQueue.hpp:14).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace rbq {

// one call of a client: its parts run on one or several workers; the caller sleeps on `cv` until all are done
struct Job {
    std::mutex mu;
    std::condition_variable cv;
    size_t pending = 0;
    int rc = 0;  // first non-zero result of a part
    std::string error;
    void wait()
    {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return pending == 0; });
    }
};

struct Task {
    std::function<int()> fn;
    Job *job;
};

// One host thread and one FIFO of tasks.  `error_text` is asked (on the worker's thread) for the text of a failed task.
class Worker {
public:
    explicit Worker(std::function<std::string()> error_text = nullptr) : error_text_(std::move(error_text)) {}
    ~Worker() { stop(); }
    Worker(const Worker &) = delete;
    Worker &operator=(const Worker &) = delete;

    void start()
    {
        thread_ = std::thread([this] { loop(); });
    }
    // tasks already queued are still run; returns when the thread has ended
    void stop()
    {
        if (!thread_.joinable()) return;
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        thread_.join();
    }
    void submit(Task t)
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            queue_.push_back(std::move(t));
            ++load_;
        }
        cv_.notify_one();
    }
    // queued + running tasks
    size_t load()
    {
        std::lock_guard<std::mutex> lock(mu_);
        return load_;
    }

private:
    void loop()
    {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return !queue_.empty() || stop_; });
                if (queue_.empty()) return;  // stop requested and nothing left to run
                t = std::move(queue_.front());
                queue_.pop_front();
            }
            const int rc = t.fn();
            const std::string err = (rc != 0 && error_text_) ? error_text_() : std::string();
            {
                std::lock_guard<std::mutex> lock(mu_);
                --load_;
            }
            {
                std::lock_guard<std::mutex> lock(t.job->mu);
                if (rc != 0 && t.job->rc == 0) { t.job->rc = rc; t.job->error = err; }
                --t.job->pending;
                // notify under the lock: the Job lives on the caller's stack and goes away as soon as pending hits zero
                t.job->cv.notify_all();
            }
        }
    }
    std::function<std::string()> error_text_;
    std::thread thread_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Task> queue_;
    size_t load_ = 0;
    bool stop_ = false;
};

// Picks workers for the parts of a call and queues them.  Callers on different host threads only meet inside dispatch():
// the `parts` parts of a call go to the `parts` least loaded workers (ties: round-robin from a cursor that moves on past the
// last worker chosen), each part behind whatever that worker still has queued (FIFO per worker).  An unsplit call is the
// case parts == 1; a call of as many parts as there are workers takes them all.  (Until round 3 a multi-part call always
// started at the cursor without moving it: concurrent 2-part calls on 8 workers all queued on the same two.)
class Dispatcher {
public:
    explicit Dispatcher(std::vector<Worker *> workers) : workers_(std::move(workers)) {}
    size_t size() const { return workers_.size(); }
    // part k of `parts` is make_task(k, worker index); returns after queueing, the caller then waits on job.  EVERY part is queued:
    // a call of more parts than workers wraps around the chosen workers (several parts behind each other on one worker) -- a
    // caller's slices are computed from its own part count, so a dropped part would be reads that are never classified.
    void dispatch(size_t parts, Job &job, const std::function<std::function<int()>(size_t, size_t)> &make_task)
    {
        std::lock_guard<std::mutex> lock(mu_);
        const size_t nw = workers_.size();
        if (nw == 0 || parts == 0) {  // nobody to queue on (or nothing to queue): the caller's wait must not hang, and `part % 0` below must not happen
            job.pending = 0;
            return;
        }
        const size_t all_parts = parts;
        if (parts > nw) parts = nw;  // workers to choose
        // workers in cursor order with their loads; a stable selection of the `parts` smallest keeps the round-robin among ties
        std::vector<size_t> order(nw), load(nw);
        for (size_t k = 0; k < nw; ++k) {
            order[k] = (next_ + k) % nw;
            load[k] = workers_[order[k]]->load();
        }
        std::vector<size_t> pick;  // positions in `order`
        std::vector<bool> taken(nw, false);
        for (size_t p = 0; p < parts; ++p) {
            size_t best = nw;
            for (size_t k = 0; k < nw; ++k)
                if (!taken[k] && (best == nw || load[k] < load[best])) best = k;
            taken[best] = true;
            pick.push_back(best);
        }
        size_t last = 0;
        for (size_t k : pick) last = std::max(last, k);
        next_ = (order[last] + 1) % nw;
        job.pending = all_parts;  // no worker sees `job` before the first submit
        std::vector<size_t> chosen;  // in cursor order: part 0 on the first chosen worker after the cursor
        for (size_t k = 0; k < nw; ++k)
            if (taken[k]) chosen.push_back(order[k]);
        for (size_t part = 0; part < all_parts; ++part) {
            const size_t w = chosen[part % chosen.size()];
            workers_[w]->submit(Task{make_task(part, w), &job});
        }
    }

private:
    std::vector<Worker *> workers_;
    std::mutex mu_;
    size_t next_ = 0;
};

}  // namespace rbq
