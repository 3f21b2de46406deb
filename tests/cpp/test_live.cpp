// tests/cpp/test_live.cpp -- the host logic of the live step (csrc/rb_live.cpp: the once_seen map, the speculative concatenation, the
// 1500 bp cut-off, decision -> action, the exception path that stores nothing; src/main/adaptive_sampling.hpp:214-356) on a CPU.
// rb_live.cpp reaches the GPU through ONE call, rb_classify_batch; this harness links rb_live.cpp + rb_host.cpp and supplies that
// symbol itself -- a deterministic stand-in whose decision is a function of the sequence -- so the product's own object code runs
// under ASan / UBSan / TSan (profiles/sanitize_cpu.sh) without a device.  What is compared: rb_live_process over micro-batches of
// every size (ids repeated inside a batch, short reads, reads that stay undecided past the cut-off) against a chunk-by-chunk
// restatement of the reference's loop with the same stand-in classifier; then four threads on one handle (TSan).
// The GPU-side twin of this test is tests/test_gpu_live.py (the real engine against the oracle-driven restatement).
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/readbouncer_amd.h"

// ---- the stand-in for the engine: decision and status from the bytes of the read alone
static void fake_classify(const char *s, uint32_t len, uint8_t *decision, uint8_t *status)
{
    if (len < 13) {  // src/IBF/IBFClassify.cpp:185-190: shorter than k -> exception
        *decision = 0;
        *status = RB_ERR_SHORT_READ;
        return;
    }
    uint32_t h = 2166136261u;
    for (uint32_t i = 0; i < len; ++i) h = (h ^ (uint8_t)s[i]) * 16777619u;
    // mostly undecided, so that reads live through several chunks and reach the cut-off
    const uint32_t r = (h >> 8) % 16;
    *decision = r == 0 ? 1 : r == 1 ? 2 : 0;
    *status = RB_OK;
}

static int g_fail_next_call = 0;  // > 0: the next call fails as a whole (a HIP error): rb_live_process must pass it on and change nothing

extern "C" int rb_classify_batch(rb_engine *, const char *seqs, const uint64_t *offsets, const uint32_t *lens, size_t n_reads, double, double,
                                 int mode, uint16_t *, int32_t *, uint8_t *out_decision, uint8_t *out_status)
{
    assert(mode == RB_MODE_CHECK_UNBLOCK && out_decision && out_status);
    if (g_fail_next_call > 0) {
        --g_fail_next_call;
        return RB_ERR_HIP;
    }
    for (size_t i = 0; i < n_reads; ++i) fake_classify(seqs + offsets[i], lens[i], &out_decision[i], &out_status[i]);
    return RB_OK;
}

// ---- the reference's loop, one chunk at a time (adaptive_sampling.hpp:227-350), with the same classifier
struct Model {
    std::map<std::string, std::string> once_seen;
    uint32_t max_undecided = 1500;
    void step(const std::string &id, const std::string &chunk, uint8_t *action, uint8_t *status, uint32_t *clen)
    {
        *action = 0;
        *clen = (uint32_t)chunk.size();
        uint8_t d, st;
        fake_classify(chunk.data(), (uint32_t)chunk.size(), &d, &st);
        *status = st;
        if (st != RB_OK) return;              // :340-349 logged, nothing pushed, nothing stored
        if (d == 1 || d == 2) {               // :241-275
            once_seen.erase(id);
            *action = d;
            return;
        }
        auto f = once_seen.find(id);
        if (f == once_seen.end()) {           // :336
            once_seen[id] = chunk;
            return;
        }
        const std::string cc = f->second + chunk;  // :284-288
        *clen = (uint32_t)cc.size();
        fake_classify(cc.data(), (uint32_t)cc.size(), &d, &st);
        if (st != RB_OK) {
            *status = st;
            return;
        }
        if (d == 1 || d == 2) {
            once_seen.erase(id);
            *action = d;
        } else if (cc.size() > max_undecided) {   // :315-325 "we assume read to be on target"
            once_seen.erase(id);
            *action = 2;
        } else {
            once_seen[id] = cc;                   // :329
        }
    }
};

struct Stream {
    std::vector<std::string> ids, chunks;
};

static uint32_t rnd(uint32_t &x)
{
    x = x * 1664525u + 1013904223u;
    return x >> 8;
}

static Stream make_stream(uint32_t seed, size_t n, const std::string &prefix)
{
    Stream s;
    uint32_t x = seed;
    for (size_t i = 0; i < n; ++i) {
        s.ids.push_back(prefix + std::to_string(rnd(x) % 40));  // 40 reads alive: plenty of repeats, also inside one micro-batch
        const uint32_t len = (rnd(x) % 20 == 0) ? rnd(x) % 13 : 150 + rnd(x) % 350;  // a few chunks shorter than k
        std::string c(len, 'A');
        for (uint32_t j = 0; j < len; ++j) c[j] = "ACGTN"[rnd(x) % 5];
        s.chunks.push_back(c);
    }
    return s;
}

static int run_stream(rb_live *lv, const Stream &s, uint32_t batch_seed, size_t max_batch, Model &model)
{
    int failures = 0;
    uint32_t x = batch_seed;
    size_t at = 0;
    while (at < s.ids.size()) {
        const size_t m = std::min(s.ids.size() - at, (size_t)(1 + rnd(x) % max_batch));
        std::string ids, seqs;
        std::vector<uint64_t> id_off(m), off(m);
        std::vector<uint32_t> id_len(m), len(m), clen(m, 0);
        for (size_t i = 0; i < m; ++i) {
            id_off[i] = ids.size();
            id_len[i] = (uint32_t)s.ids[at + i].size();
            ids += s.ids[at + i];
            off[i] = seqs.size();
            len[i] = (uint32_t)s.chunks[at + i].size();
            seqs += s.chunks[at + i];
        }
        if (seqs.empty()) seqs.push_back('N');
        std::vector<uint8_t> act(m, 9), st(m, 9);
        const int rc = rb_live_process(lv, ids.data(), id_off.data(), id_len.data(), seqs.data(), off.data(), len.data(), m, act.data(), st.data(), clen.data());
        if (rc != RB_OK) return 1000000;
        for (size_t i = 0; i < m; ++i) {
            uint8_t a, t;
            uint32_t c;
            model.step(s.ids[at + i], s.chunks[at + i], &a, &t, &c);
            if (a != act[i] || t != st[i] || (t == RB_OK && c != clen[i])) {
                if (failures < 5) std::printf("mismatch at chunk %zu (batch of %zu): action %d/%d status %d/%d len %u/%u\n", at + i, m, act[i], a, st[i], t, clen[i], c);
                ++failures;
            }
        }
        at += m;
    }
    return failures;
}

int main()
{
    int failures = 0;
    rb_engine *none = reinterpret_cast<rb_engine *>(uintptr_t(16));  // never dereferenced: the stand-in ignores it
    // 1. micro-batches of every size against the chunk-by-chunk loop
    for (size_t max_batch : {(size_t)1, (size_t)2, (size_t)7, (size_t)64, (size_t)500}) {
        rb_live *lv = nullptr;
        assert(rb_live_create(none, 0.1, 0.95, 1500, &lv) == RB_OK && lv);
        Model model;
        const Stream s = make_stream(1234u + (uint32_t)max_batch, 6000, "read");
        failures += run_stream(lv, s, 77u, max_batch, model);
        if (rb_live_pending(lv) != model.once_seen.size()) {
            std::printf("pending %zu, model %zu\n", rb_live_pending(lv), model.once_seen.size());
            ++failures;
        }
        // forget: the entry is gone, the next chunk of that read is "seen for the first time" again
        if (!model.once_seen.empty()) {
            const std::string id = model.once_seen.begin()->first;
            assert(rb_live_forget(lv, id.data(), (uint32_t)id.size()) == RB_OK);
            model.once_seen.erase(id);
            assert(rb_live_pending(lv) == model.once_seen.size());
        }
        // a call that fails as a whole changes nothing
        const size_t before = rb_live_pending(lv);
        g_fail_next_call = 1;
        const char *one_id = "zz";
        const std::string chunk(300, 'C');
        uint64_t z = 0;
        uint32_t l2 = 2, l300 = 300, cl = 0;
        uint8_t a = 9, t = 9;
        assert(rb_live_process(lv, one_id, &z, &l2, chunk.data(), &z, &l300, 1, &a, &t, &cl) == RB_ERR_HIP);
        assert(rb_live_pending(lv) == before);
        // null / empty contract
        assert(rb_live_process(lv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr) == RB_OK);
        assert(rb_live_process(lv, one_id, &z, &l2, chunk.data(), &z, &l300, 1, nullptr, &t, &cl) == RB_ERR_INVALID_ARG);
        assert(rb_live_process(nullptr, one_id, &z, &l2, chunk.data(), &z, &l300, 1, &a, &t, &cl) == RB_ERR_INVALID_ARG);
        rb_live_destroy(lv);
    }
    // 2. four threads on ONE handle (their reads are disjoint): every thread sees what its own sequential model says
    {
        rb_live *lv = nullptr;
        assert(rb_live_create(none, 0.1, 0.95, 1500, &lv) == RB_OK);
        int fails[4] = {0, 0, 0, 0};
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&, t] {
                Model model;
                const Stream s = make_stream(99u + (uint32_t)t, 3000, "t" + std::to_string(t) + "_");
                fails[t] = run_stream(lv, s, 5u + (uint32_t)t, 32, model);
            });
        for (std::thread &x : th) x.join();
        for (int f : fails) failures += f;
        rb_live_destroy(lv);
    }
    std::printf("failures: %d\n", failures);
    return failures ? 1 : 0;
}
