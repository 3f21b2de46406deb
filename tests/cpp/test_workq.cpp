// test_workq.cpp -- the host-thread machinery of the one-process pool (readbouncer_amd/csrc/rb_workq.h) on a CPU: K client
// threads, W workers, micro-calls and split calls mixed.  Every part runs exactly once, a caller's calls stay in order, an
// unsplit call lands on an idle worker while another is busy (the reference's N classification threads behind one queue,
// src/main/adaptive_sampling.hpp:745-751), errors come back to the caller that owns them, shutdown runs what is queued.
// Built plain for the CPU test suite and with -fsanitize=thread by profiles/sanitize_cpu.sh.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <thread>
#include <vector>

#include "../../readbouncer_amd/csrc/rb_workq.h"

static int failures = 0;
#define CHECK(c)                                                                 \
    do {                                                                         \
        if (!(c)) { ++failures; std::fprintf(stderr, "%s:%d CHECK(%s) failed\n", __FILE__, __LINE__, #c); } \
    } while (0)

int main()
{
    const size_t W = 4, K = 6, CALLS = 400;
    std::vector<std::unique_ptr<rbq::Worker>> workers;
    std::vector<rbq::Worker *> raw;
    for (size_t i = 0; i < W; ++i) {
        workers.emplace_back(new rbq::Worker([] { return std::string("task failed"); }));
        workers.back()->start();
        raw.push_back(workers.back().get());
    }
    rbq::Dispatcher disp(raw);
    std::vector<std::atomic<int>> ran(K * CALLS * W);
    for (auto &x : ran) x = 0;
    std::atomic<int> running{0}, max_running{0};
    std::vector<std::atomic<int>> per_worker(W);
    for (auto &x : per_worker) x = 0;
    std::vector<std::thread> clients;
    std::atomic<int> order_errors{0}, error_seen{0};
    for (size_t c = 0; c < K; ++c) {
        clients.emplace_back([&, c] {
            size_t last_done = 0;
            for (size_t i = 0; i < CALLS; ++i) {
                const size_t parts = (i % 7 == 0) ? W : (i % 11 == 0 ? 2 : 1);
                const bool fail_one = (i % 50 == 49);
                rbq::Job job;
                disp.dispatch(parts, job, [&](size_t k, size_t w) -> std::function<int()> {
                    return [&, k, w, c, i, fail_one] {
                        const int now = ++running;
                        int m = max_running.load();
                        while (now > m && !max_running.compare_exchange_weak(m, now)) {}
                        ++per_worker[w];
                        ++ran[(c * CALLS + i) * W + k];
                        std::this_thread::sleep_for(std::chrono::microseconds(20));
                        --running;
                        return (fail_one && k == 0) ? 7 : 0;
                    };
                });
                job.wait();
                if (fail_one) {
                    if (job.rc == 7 && job.error == "task failed") ++error_seen;
                } else if (job.rc != 0) {
                    ++order_errors;
                }
                if (i < last_done) ++order_errors;  // calls of one client are sequential by construction: each returned first
                last_done = i;
                for (size_t k = 0; k < parts; ++k)
                    if (ran[(c * CALLS + i) * W + k] != 1) ++order_errors;  // every part done when the call returns
            }
        });
    }
    for (auto &t : clients) t.join();
    size_t total = 0, expected = 0;
    for (size_t c = 0; c < K; ++c)
        for (size_t i = 0; i < CALLS; ++i) {
            const size_t parts = (i % 7 == 0) ? W : (i % 11 == 0 ? 2 : 1);
            expected += parts;
            for (size_t k = 0; k < W; ++k) {
                const int r = ran[(c * CALLS + i) * W + k];
                CHECK(r == (k < parts ? 1 : 0));
                total += (size_t)r;
            }
        }
    CHECK(total == expected);
    CHECK(order_errors == 0);
    CHECK(error_seen == (int)(K * (CALLS / 50)));
    CHECK(max_running >= 2);  // calls of different clients ran on different workers at the same time
    for (size_t w = 0; w < W; ++w) CHECK(per_worker[w] > (int)(expected / W / 4));  // the load is spread
    // an unsplit call goes around a busy worker
    {
        std::atomic<bool> release{false};
        rbq::Job blocker;
        disp.dispatch(1, blocker, [&](size_t, size_t) -> std::function<int()> {
            return [&] { while (!release) std::this_thread::sleep_for(std::chrono::microseconds(50)); return 0; };
        });
        for (int i = 0; i < 20; ++i) {
            rbq::Job quick;
            const auto t0 = std::chrono::steady_clock::now();
            disp.dispatch(1, quick, [&](size_t, size_t) -> std::function<int()> { return [] { return 0; }; });
            quick.wait();
            CHECK(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.5);
        }
        release = true;
        blocker.wait();
    }
    // concurrent calls of FEWER parts than there are workers are spread over all of them (until round 3 every multi-part call
    // started at the same cursor position: four 2-part calls on this pool queued on workers 0 and 1 while 2 and 3 sat idle)
    {
        std::atomic<bool> release{false};
        std::vector<std::unique_ptr<rbq::Job>> jobs;
        std::vector<std::atomic<int>> hits(W);
        for (auto &x : hits) x = 0;
        for (size_t j = 0; j < W / 2; ++j) {
            jobs.emplace_back(new rbq::Job());
            disp.dispatch(2, *jobs.back(), [&](size_t, size_t w) -> std::function<int()> {
                ++hits[w];  // counted at dispatch: which worker each part was queued on
                return [&] { while (!release) std::this_thread::sleep_for(std::chrono::microseconds(50)); return 0; };
            });
        }
        for (size_t w = 0; w < W; ++w) CHECK(hits[w] == 1);  // W/2 calls x 2 parts: one part per worker
        // one more 2-part call while every worker holds one blocked part: it still takes two DIFFERENT workers
        rbq::Job extra;
        std::vector<int> where;
        disp.dispatch(2, extra, [&](size_t, size_t w) -> std::function<int()> { where.push_back((int)w); return [] { return 0; }; });
        CHECK(where.size() == 2 && where[0] != where[1]);
        release = true;
        for (auto &j : jobs) j->wait();
        extra.wait();
    }
    // a call of MORE parts than there are workers: every part runs exactly once (a clamp to the worker count would have dropped the
    // tail of the caller's slices without an error), wrapped around the workers
    {
        const size_t parts = 3 * W + 1;
        std::vector<std::atomic<int>> ran(parts);
        for (auto &x : ran) x = 0;
        std::vector<int> per_worker(W, 0);
        rbq::Job job;
        disp.dispatch(parts, job, [&](size_t k, size_t w) -> std::function<int()> {
            ++per_worker[w];
            return [&ran, k] { ++ran[k]; return 0; };
        });
        job.wait();
        CHECK(job.rc == 0);
        for (size_t k = 0; k < parts; ++k) CHECK(ran[k] == 1);
        for (size_t w = 0; w < W; ++w) CHECK(per_worker[w] == 3 || per_worker[w] == 4);
    }
    // shutdown runs what is queued
    {
        std::atomic<int> done{0};
        std::vector<std::unique_ptr<rbq::Job>> jobs;
        for (int i = 0; i < 50; ++i) {
            jobs.emplace_back(new rbq::Job());
            disp.dispatch(1, *jobs.back(), [&](size_t, size_t) -> std::function<int()> { return [&] { ++done; return 0; }; });
        }
        for (auto &w : workers) w->stop();
        CHECK(done == 50);
        for (auto &j : jobs) j->wait();
    }
    std::printf("workq checks done, failures: %d\n", failures);
    return failures ? 1 : 0;
}
