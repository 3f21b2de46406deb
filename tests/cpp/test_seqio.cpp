// test_seqio.cpp -- CPU-only checks of the parallel ingest's failure paths (readbouncer_amd/host/seqio.hpp):
// a page-locked allocator that refuses falls back to the heap with the same records and prefixes; an allocator that
// throws inside a worker thread ends the stream with an error segment instead of std::terminate.
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#include <cstring>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <new>
#include <string>

#include "../../readbouncer_amd/host/seqio.hpp"

static int failures = 0;
#define CHECK(c)                                                                  \
    do {                                                                          \
        if (!(c)) { ++failures; std::cerr << "FAILED " #c " at line " << __LINE__ << std::endl; } \
    } while (0)

static std::string make_fastq(size_t n)
{
    std::string s;
    uint32_t x = 99;
    for (size_t i = 0; i < n; ++i) {
        const size_t len = 30 + (i * 37) % 400;
        std::string seq;
        for (size_t k = 0; k < len; ++k) { x = x * 1664525u + 1013904223u; seq += "ACGT"[(x >> 24) & 3]; }
        s += "@r" + std::to_string(i) + "\n" + seq + "\n+\n" + std::string(len, 'I') + "\n";
    }
    return s;
}

struct Digest { size_t records = 0, rows = 0; uint64_t h = 1469598103934665603ull; std::string error; };

static Digest drain(seqio::ParallelReader& rd, uint32_t prefix_len)
{
    Digest d;
    while (std::unique_ptr<seqio::Segment> seg = rd.next()) {
        d.records += seg->batch.records.size();
        d.rows += seg->prefix_idx.size();
        for (size_t i = 0; i < seg->prefix_idx.size() * (size_t)prefix_len; ++i) { d.h ^= (unsigned char)seg->prefix[i]; d.h *= 1099511628211ull; }
        if (!seg->batch.error.empty()) d.error = seg->batch.error;
    }
    return d;
}

int main()
{
    const std::string fq = make_fastq(3000);
    seqio::ParallelReader plain(fq.data(), fq.size(), 3, 8192, 100);
    const Digest a = drain(plain, 100);
    CHECK(a.records == 3000 && a.rows > 1000 && a.error.empty());

    seqio::BlockAllocator refusing;  // "page-locked memory is exhausted"
    refusing.alloc = [](size_t) -> void* { return nullptr; };
    refusing.release = [](void*) { std::abort(); };  // must never see a heap block
    {
        seqio::ParallelReader rd(fq.data(), fq.size(), 3, 8192, 100, refusing);
        const Digest b = drain(rd, 100);
        CHECK(b.records == a.records && b.rows == a.rows && b.h == a.h && b.error.empty());
    }

    seqio::BlockAllocator throwing;
    throwing.alloc = [](size_t) -> void* { throw std::bad_alloc(); };
    throwing.release = [](void*) {};
    {
        seqio::ParallelReader rd(fq.data(), fq.size(), 3, 8192, 100, throwing);
        const Digest c = drain(rd, 100);
        CHECK(!c.error.empty() && c.error.find("ingest worker") != std::string::npos);
        CHECK(c.records <= a.records);
    }
    // OrderedOutput: ranges reserved in order, filled by several threads at once (mapped windows, or heap windows written
    // positionally) -- the file is the concatenation of the pieces, cut to the bytes reserved
    for (int use_mmap = 0; use_mmap < 2; ++use_mmap) {
        const std::string path = std::string("/tmp/rb_test_ordered_output_") + std::to_string((long)getpid()) + (use_mmap ? "_m" : "_w");
        std::string expect;
        std::vector<std::string> pieces;
        for (int i = 0; i < 200; ++i) {
            pieces.emplace_back((size_t)((i * 7919) % 30011) + (i % 5 == 0 ? 0 : 1), (char)('a' + i % 26));  // some empty pieces
            if (i % 5 == 0) pieces.back().clear();
            if (i == 77 || i == 78) pieces.back().assign(((size_t)5 << 20) + 3 + (size_t)i, (char)('A' + i % 26));  // ranges of several chunk buffers
            expect += pieces.back();
        }
        {
            seqio::OrderedOutput out;
            CHECK(out.open(path, use_mmap != 0));
            std::vector<uint64_t> at(pieces.size());
            for (size_t i = 0; i < pieces.size(); ++i) at[i] = out.reserve(pieces[i].size());
            std::vector<std::thread> th;
            std::atomic<int> short_ranges{0};
            for (int t = 0; t < 4; ++t)
                th.emplace_back([&, t] {
                    for (size_t i = (size_t)t; i < pieces.size(); i += 4) {
                        seqio::OrderedOutput::Writer w = out.writer(at[i], pieces[i].size());
                        // records of uneven sizes, some larger than nothing, one larger than the chunk buffer would be with a small kChunk
                        size_t pos = 0, step = 1;
                        while (pos < pieces[i].size()) {
                            const size_t n = std::min(step, pieces[i].size() - pos);
                            std::memcpy(w.take(n), pieces[i].data() + pos, n);
                            pos += n;
                            step = step * 3 + 1;
                        }
                        if (!w.finish()) ++short_ranges;
                    }
                });
            for (auto& x : th) x.join();
            CHECK(out.bytes() == expect.size() && short_ranges == 0);
            {   // a range that is not used up is reported, one that is overrun throws
                seqio::OrderedOutput::Writer w = out.writer(out.reserve(0), 0);
                bool threw = false;
                try { w.take(1); } catch (const std::exception&) { threw = true; }
                CHECK(threw && w.finish());
            }
            out.close();
            CHECK(out.ok());
        }
        std::ifstream in(path, std::ios::binary);
        std::stringstream got;
        got << in.rdbuf();
        CHECK(got.str() == expect);
        std::remove(path.c_str());
    }
    {
        // more formatting threads than a file has pool buffers, TWO output files, every thread holding a half-filled
        // buffer of one file while it asks for a buffer of the other (what a classifier thread does: one Writer per output file for
        // the whole format pass of its segment).  Half of the threads take file A first, half file B first.  Used to hang with
        // 0 of 30 threads finishing; a watchdog turns a hang into a failure.
        const int n_threads = 30;
        static_assert(n_threads > 2 * (int)seqio::OrderedOutput::kPoolBuffers, "more threads than both pools hold");
        const std::string pa = std::string("/tmp/rb_test_two_files_") + std::to_string((long)getpid()) + "_a";
        const std::string pb = std::string("/tmp/rb_test_two_files_") + std::to_string((long)getpid()) + "_b";
        seqio::OrderedOutput a, b;
        CHECK(a.open(pa, false) && b.open(pb, false));
        const size_t piece = 3000;
        std::vector<uint64_t> at_a(n_threads), at_b(n_threads);
        for (int t = 0; t < n_threads; ++t) { at_a[t] = a.reserve(piece); at_b[t] = b.reserve(piece); }
        std::atomic<int> holding{0}, done{0};
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&, t] {
                seqio::OrderedOutput::Writer wa = a.writer(at_a[t], piece), wb = b.writer(at_b[t], piece);
                seqio::OrderedOutput::Writer& first = (t & 1) ? wb : wa;
                seqio::OrderedOutput::Writer& second = (t & 1) ? wa : wb;
                std::memset(first.take(piece / 2), 'a' + t % 26, piece / 2);  // holds a buffer of its first file from here on
                ++holding;
                while (holding.load() < n_threads && done.load() == 0) std::this_thread::yield();  // everybody holds one
                std::memset(second.take(piece), 'A' + t % 26, piece);
                std::memset(first.take(piece - piece / 2), 'a' + t % 26, piece - piece / 2);
                if (wa.finish() && wb.finish()) ++done;
            });
        for (int ms = 0; ms < 20000 && done.load() < n_threads; ms += 10) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        CHECK(done.load() == n_threads);
        if (done.load() < n_threads) {
            std::cerr << "two output files, " << n_threads << " threads: " << done.load() << " finished -- deadlock" << std::endl;
            std::_Exit(1);  // the stuck threads cannot be joined
        }
        for (auto& x : th) x.join();
        a.close();
        b.close();
        CHECK(a.ok() && b.ok() && a.bytes() == (uint64_t)n_threads * piece && b.bytes() == (uint64_t)n_threads * piece);
        for (const std::string* path : {&pa, &pb}) {
            std::ifstream in(*path, std::ios::binary);
            std::stringstream got;
            got << in.rdbuf();
            const std::string text = got.str();
            bool good = text.size() == (size_t)n_threads * piece;
            for (int t = 0; good && t < n_threads; ++t) {
                const bool first_file = ((t & 1) != 0) == (path == &pb);
                const char want = (char)((first_file ? 'a' : 'A') + t % 26);
                for (size_t i = 0; i < piece; ++i) good = good && text[(size_t)t * piece + i] == want;
            }
            CHECK(good);
            std::remove(path->c_str());
        }
    }
    {
        seqio::OrderedOutput nothing;  // a file nobody wrote to ends up empty
        const std::string path = std::string("/tmp/rb_test_ordered_output_") + std::to_string((long)getpid()) + "_e";
        CHECK(nothing.open(path, true));
        nothing.close();
        std::ifstream in(path, std::ios::binary | std::ios::ate);
        CHECK(in.good() && in.tellg() == 0);
        std::remove(path.c_str());
        seqio::OrderedOutput bad;
        CHECK(!bad.open("/nonexistent_dir_rb/x.fasta", true));
    }
    std::cout << "seqio checks done, failures: " << failures << std::endl;
    return failures ? 1 : 0;
}
