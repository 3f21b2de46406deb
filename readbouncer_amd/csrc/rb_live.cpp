// rb_live.cpp -- micro-batch form of classify_live_reads (src/main/adaptive_sampling.hpp:214-356): the step
// between the reference's classification_queue and action_queue.  Host logic only (the once_seen map, the
// concatenation of undecided chunks, the 1500 bp cut-off, decision -> action); all classification goes through
// rb_classify_batch, i.e. the GPU.  Per read and in arrival order the outcome equals the reference's loop:
//   decision = check_unblock(new chunk)                                            (:237)
//   1 -> unblock, forget the read (:241-264);  2 -> stop_receiving, forget it (:265-275)
//   0 -> if seen before: classify stored + new chunk (:284-288); 1/2 as above; still 0: longer than
//        1500 bp -> stop_receiving ("we assume read to be on target", :315-325) else store the
//        concatenation (:329); if not seen before: store the chunk, no action (:336)
//   exception (short read ...) -> logged, no action, state untouched (:340-349)
// The action codes follow Data::sendActions (src/minknow/Data.cpp:169-187): unblock=true -> unblock_read,
// unblock=false -> stop_receiving_data.
#include <algorithm>
#include <chrono>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "rb_internal.h"

struct rb_live {
    rb_engine *engine = nullptr;
    double error_rate = 0.1, significance = 0.95;
    uint32_t max_undecided_len = 1500;
    std::unordered_map<std::string, std::pair<std::string, uint8_t>> once_seen;  // id -> (sequence so far, iterstep)
    std::mutex mu;
};

namespace {

struct Item {
    size_t index;        // position in the caller's batch
    std::string id;
    const char *seq;
    uint32_t len;
};

int classify_strings(rb_live *lv, const std::vector<const char *> &ptrs, const std::vector<uint32_t> &lens,
                     std::vector<uint8_t> &decision, std::vector<uint8_t> &status)
{
    const size_t n = ptrs.size();
    decision.assign(n, 0);
    status.assign(n, 0);
    if (n == 0) return RB_OK;
    std::string flat;
    size_t total = 0;
    for (uint32_t l : lens) total += l;
    flat.reserve(total + 1);
    std::vector<uint64_t> offs(n);
    for (size_t i = 0; i < n; ++i) {
        offs[i] = flat.size();
        flat.append(ptrs[i], lens[i]);
    }
    if (flat.empty()) flat.push_back('N');
    return rb_classify_batch(lv->engine, flat.data(), offs.data(), lens.data(), n, lv->error_rate, lv->significance,
                             RB_MODE_CHECK_UNBLOCK, nullptr, nullptr, decision.data(), status.data());
}

}  // namespace

extern "C" {

int rb_live_create(rb_engine *e, double error_rate, double significance, uint32_t max_undecided_len, rb_live **out)
{
    if (!e || !out) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    rb_live *lv = new (std::nothrow) rb_live();
    if (!lv) return rb::fail(RB_ERR_NOMEM, "alloc");
    lv->engine = e;
    lv->error_rate = error_rate;
    lv->significance = significance;
    lv->max_undecided_len = max_undecided_len;
    // once_seen holds the reads that wait for more data: at most one per sequencing channel (48 PromethION flow cells: 144 k).  Room for
    // that many from the start, so that the map never rehashes in the middle of a run -- a rehash at 600 k pending reads stalled one call
    // of a 20 s synthetic replay for 38 ms (profiles/r05/c5_replay_20s.txt; pending reads only pile up like that when no read ever ends)
    lv->once_seen.reserve((size_t)1 << 18);
    *out = lv;
    return RB_OK;
}

void rb_live_destroy(rb_live *lv) { delete lv; }

size_t rb_live_pending(rb_live *lv)
{
    if (!lv) return 0;
    std::lock_guard<std::mutex> lock(lv->mu);
    return lv->once_seen.size();
}

int rb_live_forget(rb_live *lv, const char *id, uint32_t id_len)
{
    if (!lv || !id) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lock(lv->mu);
    lv->once_seen.erase(std::string(id, id_len));
    return RB_OK;
}

int rb_live_process(rb_live *lv, const char *ids, const uint64_t *id_offsets, const uint32_t *id_lens, const char *seqs,
                    const uint64_t *offsets, const uint32_t *lens, size_t n, uint8_t *out_action, uint8_t *out_status,
                    uint32_t *out_classified_len)
{
    if (!lv) return rb::fail(RB_ERR_INVALID_ARG, "null live handle");
    if (n == 0) return RB_OK;
    if (!ids || !id_offsets || !id_lens || !seqs || !offsets || !lens || !out_action)
        return rb::fail(RB_ERR_INVALID_ARG, "null buffer");
    std::lock_guard<std::mutex> lock(lv->mu);
    std::vector<Item> todo(n);
    for (size_t i = 0; i < n; ++i) {
        todo[i] = Item{i, std::string(ids + id_offsets[i], id_lens[i]), seqs + offsets[i], lens[i]};
        out_action[i] = 0;
        if (out_status) out_status[i] = RB_OK;
        if (out_classified_len) out_classified_len[i] = lens[i];
    }
    // A read id may occur more than once in a micro-batch (two chunks of one read).  The reference handles them
    // one after the other through once_seen; rounds keep that order: each round takes the first pending chunk of every id.
    while (!todo.empty()) {
        std::vector<Item> round, later;
        {
            std::unordered_map<std::string, int> taken;
            for (Item &it : todo) {
                if (taken.emplace(it.id, 1).second) round.push_back(std::move(it));
                else later.push_back(std::move(it));
            }
        }
        // ONE GPU call per round: the new chunks on their own, and -- speculatively, in the same batch -- the concatenation
        // stored + new chunk for every read that once_seen already holds.  The reference classifies the concatenation only
        // after the chunk alone came back undecided (adaptive_sampling.hpp:276-288); asking for both at once costs GPU work
        // for the reads that are decided by the chunk alone (rare: a read in once_seen was undecided before) and saves the
        // second call, i.e. half the latency of the live step.  Which answer counts is decided exactly as in the reference.
        std::vector<const char *> ptrs;
        std::vector<uint32_t> ls;
        for (const Item &it : round) { ptrs.push_back(it.seq); ls.push_back(it.len); }
        std::vector<size_t> extra(round.size(), (size_t)-1);  // position of the read's concatenation in the batch
        std::vector<std::string> concat;
        concat.reserve(round.size());
        for (size_t j = 0; j < round.size(); ++j) {
            auto f = lv->once_seen.find(round[j].id);
            if (f == lv->once_seen.end()) continue;
            extra[j] = round.size() + concat.size();
            concat.push_back(f->second.first + std::string(round[j].seq, round[j].len));
        }
        for (const std::string &c : concat) { ptrs.push_back(c.data()); ls.push_back((uint32_t)c.size()); }
        std::vector<uint8_t> dec, st;
        int rc = classify_strings(lv, ptrs, ls, dec, st);
        if (rc != RB_OK) return rc;
        for (size_t j = 0; j < round.size(); ++j) {
            const Item &it = round[j];
            if (st[j] != RB_OK) {  // exception path: nothing is pushed, nothing is stored
                if (out_status) out_status[it.index] = st[j];
                continue;
            }
            if (dec[j] == 1 || dec[j] == 2) {  // decided by the chunk alone (:241-275)
                lv->once_seen.erase(it.id);
                out_action[it.index] = dec[j];
                continue;
            }
            if (extra[j] == (size_t)-1) {  // undecided for the first time: remember the chunk (:336)
                lv->once_seen[it.id] = std::make_pair(std::string(it.seq, it.len), (uint8_t)1);
                continue;
            }
            // undecided, seen before: the concatenation decides (:284-329)
            const size_t a = extra[j];
            std::string &cc = concat[a - round.size()];
            if (out_classified_len) out_classified_len[it.index] = (uint32_t)cc.size();
            if (st[a] != RB_OK) {
                if (out_status) out_status[it.index] = st[a];
                continue;
            }
            if (dec[a] == 1 || dec[a] == 2) {
                lv->once_seen.erase(it.id);
                out_action[it.index] = dec[a];
            } else if (cc.size() > lv->max_undecided_len) {
                lv->once_seen.erase(it.id);
                out_action[it.index] = 2;  // unblock = false -> stop_receiving_data
            } else {
                auto f = lv->once_seen.find(it.id);
                const uint8_t step = f != lv->once_seen.end() ? (uint8_t)(f->second.second + 1) : (uint8_t)1;
                lv->once_seen[it.id] = std::make_pair(std::move(cc), step);
            }
        }
        todo.swap(later);
    }
    return RB_OK;
}

// Work-conserving replay of an arrival process (BASELINE configs[4], the 48-flowcell scenario): chunk i -- read_len bytes
// at seqs + i * read_len -- becomes available arrival_s[i] seconds after the start (ascending).  Whenever the engine is
// free the dispatcher takes everything that has arrived, at most max_batch chunks, through ONE rb_classify_batch call
// (host buffers in, decisions back on the host) and otherwise spins on the steady clock.  Per chunk: the decision and the
// latency decision - arrival; per call: its size and its service time, so that queueing (waiting for the engine) and
// service (the call itself) can be told apart.  This is the reference's classification thread (adaptive_sampling.hpp:
// 214-356 pops one read at a time) with a queue drained in micro-batches; it lives in the library so that the
// measurement does not carry an interpreter's loop in its percentiles.
// (spin with a pause hint: the dispatcher thread owns its core for the length of the replay, like the reference's
// classification thread, which polls classification_queue.empty() in a tight loop -- adaptive_sampling.hpp:226-228)
static inline void spin_pause()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

static int check_arrivals(const double *arrival_s, size_t n)
{
    for (size_t i = 1; i < n; ++i)
        if (!(arrival_s[i] >= arrival_s[i - 1])) return rb::fail(RB_ERR_INVALID_ARG, "arrival times must be ascending");
    return RB_OK;
}

int rb_replay_arrivals(rb_engine *e, const char *seqs, uint32_t read_len, size_t n, const double *arrival_s, size_t max_batch,
                       double error_rate, double significance, uint8_t *out_decision, double *out_latency_s,
                       uint32_t *out_call_reads, double *out_call_service_s, size_t call_cap, size_t *out_calls,
                       double *out_elapsed_s)
{
    if (!e || !seqs || !arrival_s || !out_latency_s || read_len == 0) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    if (out_calls) *out_calls = 0;
    if (out_elapsed_s) *out_elapsed_s = 0.0;
    if (n == 0) return RB_OK;
    int rc = check_arrivals(arrival_s, n);
    if (rc != RB_OK) return rc;
    if (max_batch == 0) max_batch = 16384;
    max_batch = std::min(std::min(max_batch, n), (size_t)1 << 20);  // four staging vectors of this length below
    std::vector<uint64_t> offs(max_batch);
    std::vector<uint32_t> lens(max_batch, read_len);
    for (size_t i = 0; i < max_batch; ++i) offs[i] = (uint64_t)i * read_len;
    std::vector<uint8_t> dec(max_batch), st(max_batch);
    using clk = std::chrono::steady_clock;
    const clk::time_point t0 = clk::now();
    auto now_s = [&] { return std::chrono::duration<double>(clk::now() - t0).count(); };
    size_t done = 0, hi = 0, calls = 0;
    while (done < n) {
        const double now = now_s();
        while (hi < n && arrival_s[hi] <= now) ++hi;
        if (hi <= done) { spin_pause(); continue; }  // spin until the next chunk arrives
        const size_t m = std::min(hi - done, max_batch);
        const double a = now_s();
        rc = rb_classify_batch(e, seqs + done * (size_t)read_len, offs.data(), lens.data(), m, error_rate, significance,
                               RB_MODE_CHECK_UNBLOCK, nullptr, nullptr, dec.data(), st.data());
        if (rc != RB_OK) return rc;
        const double b = now_s();
        for (size_t i = 0; i < m; ++i) {
            out_latency_s[done + i] = b - arrival_s[done + i];
            if (out_decision) out_decision[done + i] = dec[i];
        }
        if (calls < call_cap) {
            if (out_call_reads) out_call_reads[calls] = (uint32_t)m;
            if (out_call_service_s) out_call_service_s[calls] = b - a;
        }
        ++calls;
        done += m;
        hi = std::max(hi, done);
    }
    if (out_calls) *out_calls = calls;
    if (out_elapsed_s) *out_elapsed_s = now_s();
    return RB_OK;
}

// The same dispatcher in front of the LIVE step (rb_live_process): chunk i belongs to read read_ids[i] (its id is the four
// bytes of that number), so an undecided read's next chunk is classified as the concatenation with what once_seen holds --
// up to the 1500 bp cut-off, i.e. the 16-plane kernels -- exactly as the reference's classification thread does it
// (adaptive_sampling.hpp:276-338).  Latency = action (or "keep waiting") known on the host - arrival of the chunk.
int rb_live_replay_arrivals(rb_live *lv, const uint32_t *read_ids, const char *seqs, uint32_t read_len, size_t n,
                            const double *arrival_s, size_t max_batch, uint8_t *out_action, double *out_latency_s,
                            uint32_t *out_classified_len, uint32_t *out_call_reads, double *out_call_service_s, size_t call_cap,
                            size_t *out_calls, double *out_elapsed_s)
{
    if (!lv || !read_ids || !seqs || !arrival_s || !out_latency_s || read_len == 0) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    if (out_calls) *out_calls = 0;
    if (out_elapsed_s) *out_elapsed_s = 0.0;
    if (n == 0) return RB_OK;
    int rc = check_arrivals(arrival_s, n);
    if (rc != RB_OK) return rc;
    if (max_batch == 0) max_batch = 16384;
    max_batch = std::min(std::min(max_batch, n), (size_t)1 << 20);
    std::vector<uint64_t> offs(max_batch), id_offs(max_batch);
    std::vector<uint32_t> lens(max_batch, read_len), id_lens(max_batch, 4u), clen(max_batch);
    for (size_t i = 0; i < max_batch; ++i) { offs[i] = (uint64_t)i * read_len; id_offs[i] = (uint64_t)i * 4; }
    std::vector<uint8_t> act(max_batch), st(max_batch);
    using clk = std::chrono::steady_clock;
    const clk::time_point t0 = clk::now();
    auto now_s = [&] { return std::chrono::duration<double>(clk::now() - t0).count(); };
    size_t done = 0, hi = 0, calls = 0;
    while (done < n) {
        const double now = now_s();
        while (hi < n && arrival_s[hi] <= now) ++hi;
        if (hi <= done) { spin_pause(); continue; }
        const size_t m = std::min(hi - done, max_batch);
        const double a = now_s();
        rc = rb_live_process(lv, reinterpret_cast<const char *>(read_ids + done), id_offs.data(), id_lens.data(),
                             seqs + done * (size_t)read_len, offs.data(), lens.data(), m, act.data(), st.data(), clen.data());
        if (rc != RB_OK) return rc;
        const double b = now_s();
        for (size_t i = 0; i < m; ++i) {
            out_latency_s[done + i] = b - arrival_s[done + i];
            if (out_action) out_action[done + i] = act[i];
            if (out_classified_len) out_classified_len[done + i] = clen[i];
        }
        if (calls < call_cap) {
            if (out_call_reads) out_call_reads[calls] = (uint32_t)m;
            if (out_call_service_s) out_call_service_s[calls] = b - a;
        }
        ++calls;
        done += m;
        hi = std::max(hi, done);
    }
    if (out_calls) *out_calls = calls;
    if (out_elapsed_s) *out_elapsed_s = now_s();
    return RB_OK;
}

}  // extern "C"
