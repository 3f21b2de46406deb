// seqio.hpp -- FASTA/FASTQ ingest for the host driver (the reference uses seqan::SeqFileIn/SeqFileOut,
// src/main/classify.hpp:217-237,301).  The file is memory-mapped and scanned with memchr; records are views into
// the mapping (zero copy) unless a FASTA sequence spans several lines, in which case it is joined into an arena
// owned by the batch.  Format is detected per record from the first character ('>' FASTA, '@' FASTQ);
// multi-line FASTA, CRLF and blank lines are accepted.  A reader thread can parse batch i+1 while the GPU
// classifies batch i (host ingest is the end-to-end bottleneck above ~1 M reads/s).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <ostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace seqio
{

class MappedFile
{
    const char* data_ = nullptr;
    size_t size_ = 0;
    int fd_ = -1;
public:
    explicit MappedFile(const std::string& path)
    {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) return;
        struct stat st;
        if (fstat(fd_, &st) != 0) { ::close(fd_); fd_ = -1; return; }
        size_ = (size_t)st.st_size;
        if (size_ == 0) { data_ = ""; return; }
        void* p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (p == MAP_FAILED) { ::close(fd_); fd_ = -1; size_ = 0; return; }
        madvise(p, size_, MADV_SEQUENTIAL);
        data_ = (const char*)p;
    }
    ~MappedFile()
    {
        if (data_ && size_) munmap((void*)data_, size_);
        if (fd_ >= 0) ::close(fd_);
    }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    bool is_open() const { return fd_ >= 0; }
    const char* data() const { return data_; }
    size_t size() const { return size_; }
};

struct Record
{
    const char* id = nullptr;
    uint32_t id_len = 0;
    const char* seq = nullptr;  // into the mapping or into the batch arena
    uint32_t seq_len = 0;
};

struct Batch
{
    std::vector<Record> records;
    std::deque<std::string> arena;  // joined multi-line sequences (deque: stable addresses)
    bool eof = false;
    std::string error;              // non-empty: parsing stopped at a malformed record
};

class Parser
{
    const char* p_;
    const char* end_;

    // [p_, eol) without the trailing \r; advances p_ past the \n
    bool line(const char*& b, const char*& e)
    {
        if (p_ >= end_) return false;
        const char* nl = (const char*)memchr(p_, '\n', (size_t)(end_ - p_));
        b = p_;
        e = nl ? nl : end_;
        p_ = nl ? nl + 1 : end_;
        if (e > b && e[-1] == '\r') --e;
        return true;
    }

public:
    Parser(const char* data, size_t size) : p_(data), end_(data + size) {}

    // false at end of input; throws std::runtime_error on malformed records
    bool next(Record& r, std::deque<std::string>& arena)
    {
        const char *b, *e;
        do {
            if (!line(b, e)) return false;
        } while (b == e);
        if (*b == '>') {
            r.id = b + 1;
            r.id_len = (uint32_t)(e - b - 1);
            r.seq = e;
            r.seq_len = 0;
            int n_lines = 0;
            std::string* joined = nullptr;
            while (p_ < end_ && *p_ != '>') {
                const char *sb, *se;
                line(sb, se);
                if (sb == se) continue;
                if (n_lines == 0) {
                    r.seq = sb;
                    r.seq_len = (uint32_t)(se - sb);
                } else {
                    if (!joined) {
                        arena.emplace_back(r.seq, r.seq_len);
                        joined = &arena.back();
                    }
                    joined->append(sb, (size_t)(se - sb));
                }
                ++n_lines;
            }
            if (joined) {
                r.seq = joined->data();
                r.seq_len = (uint32_t)joined->size();
            }
            return true;
        }
        if (*b == '@') {
            r.id = b + 1;
            r.id_len = (uint32_t)(e - b - 1);
            const char *sb, *se, *pb, *pe, *qb, *qe;
            if (!line(sb, se)) throw std::runtime_error("FASTQ: truncated record " + std::string(r.id, r.id_len));
            if (!line(pb, pe) || pb == pe || *pb != '+')
                throw std::runtime_error("FASTQ: '+' line expected in " + std::string(r.id, r.id_len));
            if (!line(qb, qe)) throw std::runtime_error("FASTQ: quality line missing in " + std::string(r.id, r.id_len));
            r.seq = sb;
            r.seq_len = (uint32_t)(se - sb);
            return true;
        }
        throw std::runtime_error("unrecognised sequence record starting with '" + std::string(b, (size_t)std::min<ptrdiff_t>(10, e - b)) + "'");
    }

    // up to max_records records, or fewer at the end of input / at a malformed record
    void next_batch(Batch& out, size_t max_records)
    {
        out.records.clear();
        out.arena.clear();
        out.eof = false;
        out.error.clear();
        Record r;
        try {
            while (out.records.size() < max_records) {
                if (!next(r, out.arena)) { out.eof = true; break; }
                out.records.push_back(r);
            }
        } catch (const std::exception& ex) {
            out.error = ex.what();
            out.eof = true;
        }
    }
};

// ---- parallel ingest ------------------------------------------------------------------------------------------------
// The mapping is cut into segments that start at record boundaries; worker threads parse whole segments (and copy the
// first `prefix_len` bases of every long-enough read into one flat buffer: the classifier's first chunk), the consumer
// takes them in file order.  One parser thread does 6 GB/s of FASTQ; the GPU wants more than that.
// where the prefix buffers live: plain heap by default; the GPU driver passes page-locked memory so that the copy to
// the device is a plain DMA.  Blocks are pooled by the reader (page-locking is slow).
struct BlockAllocator
{
    void* (*alloc)(size_t bytes) = nullptr;
    void (*release)(void* p) = nullptr;
};

class BlockPool
{
    BlockAllocator a_;
    std::mutex mu_;
    std::vector<std::pair<char*, size_t>> idle_;
    std::vector<char*> heap_;  // blocks that fell back to the heap
public:
    explicit BlockPool(BlockAllocator a) : a_(a) {}
    ~BlockPool()
    {
        std::vector<std::pair<char*, size_t>> idle;
        {
            std::lock_guard<std::mutex> lock(mu_);
            idle.swap(idle_);
        }
        for (auto& b : idle) free_block(b.first);
    }
    BlockPool(const BlockPool&) = delete;
    BlockPool& operator=(const BlockPool&) = delete;
    // page-locked memory can run out (or be refused): such a block comes from the heap instead -- slower copies, same results
    char* raw_alloc(size_t bytes)
    {
        char* p = a_.alloc ? (char*)a_.alloc(bytes) : nullptr;
        if (p) return p;
        p = (char*)std::malloc(bytes);
        if (p && a_.alloc) {
            std::lock_guard<std::mutex> lock(mu_);
            heap_.push_back(p);
        }
        return p;
    }
    void free_block(char* p)
    {
        bool heap = !a_.release;
        if (!heap) {
            std::lock_guard<std::mutex> lock(mu_);
            auto it = std::find(heap_.begin(), heap_.end(), p);
            if (it != heap_.end()) { heap_.erase(it); heap = true; }
        }
        if (heap) std::free(p); else a_.release(p);
    }
    // a block of at least `bytes` (capacity returned through cap); throws std::bad_alloc
    char* get(size_t bytes, size_t* cap)
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            for (size_t i = 0; i < idle_.size(); ++i) {
                if (idle_[i].second >= bytes) {
                    std::pair<char*, size_t> b = idle_[i];
                    idle_.erase(idle_.begin() + (ptrdiff_t)i);
                    *cap = b.second;
                    return b.first;
                }
            }
        }
        const size_t want = bytes + bytes / 8 + 4096;
        char* p = raw_alloc(want);
        if (!p) throw std::bad_alloc();
        *cap = want;
        return p;
    }
    void put(char* p, size_t cap)
    {
        std::lock_guard<std::mutex> lock(mu_);
        idle_.emplace_back(p, cap);
    }
};

struct Segment
{
    Batch batch;                       // every record of the segment (views into the mapping / the arena)
    char* prefix = nullptr;            // prefix_len bases of each record with seq_len >= prefix_len, back to back
    size_t prefix_cap = 0;
    std::vector<uint32_t> prefix_idx;  // record index of each prefix row
    BlockPool* pool = nullptr;
    Segment() = default;
    Segment(const Segment&) = delete;
    Segment& operator=(const Segment&) = delete;
    ~Segment()
    {
        if (prefix && pool) pool->put(prefix, prefix_cap);
    }
    size_t index = 0;                  // position of this segment in the file (ParallelReader::release)
};

namespace detail
{
inline size_t line_end(const char* d, size_t n, size_t pos)  // index of the '\n' that ends the line at pos, or n
{
    const char* nl = (const char*)memchr(d + pos, '\n', n - pos);
    return nl ? (size_t)(nl - d) : n;
}
inline size_t trimmed_len(const char* d, size_t b, size_t e) { return (e > b && d[e - 1] == '\r') ? e - b - 1 : e - b; }

// a four-line FASTQ record starts at line start L: '@' line, sequence, '+' line, quality of the sequence's length
inline bool fastq_record_at(const char* d, size_t n, size_t L, size_t* next)
{
    if (L >= n || d[L] != '@') return false;
    const size_t e0 = line_end(d, n, L);
    if (e0 >= n) return false;
    const size_t s1 = e0 + 1, e1 = line_end(d, n, s1);
    if (e1 >= n) return false;
    const size_t s2 = e1 + 1, e2 = line_end(d, n, s2);
    if (s2 >= n || d[s2] != '+' || e2 >= n) return false;
    const size_t s3 = e2 + 1, e3 = line_end(d, n, s3);
    if (trimmed_len(d, s1, e1) != trimmed_len(d, s3, e3)) return false;
    *next = e3 < n ? e3 + 1 : n;
    return true;
}

// first record boundary at or after pos ('>' line for FASTA; for FASTQ a line that starts two well-formed records in a
// row or the last one -- a quality line may begin with '@', its successors do not fit the pattern)
inline size_t find_record_start(const char* d, size_t n, size_t pos, char fmt)
{
    if (pos == 0) return 0;
    if (pos >= n) return n;
    size_t L = line_end(d, n, pos - 1);
    L = L < n ? L + 1 : n;
    while (L < n) {
        if (fmt == '>') {
            if (d[L] == '>') return L;
        } else {
            size_t nx = 0, nx2 = 0;
            if (fastq_record_at(d, n, L, &nx)) {
                size_t R = nx;
                while (R < n && (d[R] == '\n' || d[R] == '\r')) ++R;  // blank lines between records
                if (R >= n || fastq_record_at(d, n, R, &nx2)) return L;
            }
        }
        const size_t e = line_end(d, n, L);
        L = e < n ? e + 1 : n;
    }
    return n;
}
}  // namespace detail

class ParallelReader
{
    const char* data_;
    size_t size_;
    uint32_t prefix_len_;
    BlockPool pool_;              // declared before the slots: segments hand their blocks back on destruction
    std::vector<size_t> starts_;  // segment s = [starts_[s], starts_[s + 1])
    std::vector<std::unique_ptr<Segment>> slots_;
    std::vector<std::thread> workers_;
    std::atomic<size_t> next_seg_{0};
    size_t consumed_ = 0, window_;
    bool stop_ = false;
    std::mutex mu_;
    std::condition_variable cv_ready_, cv_room_;
    std::mutex rel_mu_;
    std::vector<bool> done_;      // segments whose records nobody looks at any more
    size_t released_ = 0;         // segments [0, released_) have left the address space

    void work()
    {
        for (;;) {
            const size_t s = next_seg_.fetch_add(1);
            if (s + 1 >= starts_.size()) return;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_room_.wait(lock, [&] { return stop_ || s < consumed_ + window_; });
                if (stop_) return;
            }
            std::unique_ptr<Segment> seg(new Segment());
            seg->index = s;
            // nothing may escape a worker thread (std::terminate, with the consumers blocked in next()): a failure becomes
            // a segment whose batch.error is set -- the consumer's ordinary error path, and the last segment it sees
            try {
            Parser parser(data_ + starts_[s], starts_[s + 1] - starts_[s]);
            parser.next_batch(seg->batch, (size_t)-1);
            seg->batch.eof = seg->batch.eof && (!seg->batch.error.empty() || s + 2 == starts_.size());
            if (prefix_len_) {
                size_t rows = 0;
                for (const Record& r : seg->batch.records) rows += r.seq_len >= prefix_len_;
                seg->pool = &pool_;
                seg->prefix = pool_.get(rows * (size_t)prefix_len_ + 1, &seg->prefix_cap);
                seg->prefix_idx.reserve(rows);
                char* out = seg->prefix;
                for (size_t i = 0; i < seg->batch.records.size(); ++i) {
                    const Record& r = seg->batch.records[i];
                    if (r.seq_len < prefix_len_) continue;
                    std::memcpy(out, r.seq, prefix_len_);
                    out += prefix_len_;
                    seg->prefix_idx.push_back((uint32_t)i);
                }
            }
            } catch (const std::exception& ex) {
                seg->batch.error = std::string("ingest worker: ") + ex.what();
                seg->batch.eof = true;
            } catch (...) {
                seg->batch.error = "ingest worker: unknown failure";
                seg->batch.eof = true;
            }
            {
                std::lock_guard<std::mutex> lock(mu_);
                slots_[s] = std::move(seg);
            }
            cv_ready_.notify_all();
        }
    }

public:
    // prefix_len = 0: no prefix buffers.  segment_bytes ~ one classifier batch worth of file.
    ParallelReader(const char* data, size_t size, unsigned threads, size_t segment_bytes, uint32_t prefix_len,
                   BlockAllocator allocator = BlockAllocator())
        : data_(data), size_(size), prefix_len_(prefix_len), pool_(allocator)
    {
        if (threads == 0) threads = 1;
        if (segment_bytes < 4096) segment_bytes = 4096;
        size_t first = 0;
        while (first < size && (data[first] == '\n' || data[first] == '\r')) ++first;
        const char fmt = first < size ? data[first] : '>';
        starts_.push_back(0);
        if (fmt == '>' || fmt == '@') {  // anything else: one segment, the parser reports the malformed record
            for (size_t pos = segment_bytes; pos < size; pos += segment_bytes) {
                const size_t b = detail::find_record_start(data, size, pos, fmt);
                if (b >= size) break;
                if (b > starts_.back()) starts_.push_back(b);
                if (b > pos) pos = b;
            }
        }
        starts_.push_back(size);
        slots_.resize(starts_.size() - 1);
        window_ = 2 * (size_t)threads + 2;
        const size_t n_workers = std::min<size_t>(threads, slots_.size());
        for (size_t i = 0; i < n_workers; ++i) workers_.emplace_back([this] { work(); });
    }
    ~ParallelReader()
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_room_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    ParallelReader(const ParallelReader&) = delete;
    ParallelReader& operator=(const ParallelReader&) = delete;

    size_t segments() const { return slots_.size(); }

    // The consumer is done with the records of segment `index` (their views into the mapping are dead).  Once a stretch of at least
    // 512 MB of consecutive finished segments has built up from the front of the file, its whole pages are dropped from the address
    // space (the data stays in the page cache): a 30 GB read file otherwise ends the run with 8 M page-table entries to tear down at
    // once and a resident set the size of the file.  In large steps on purpose: every MADV_DONTNEED is a TLB shoot-down on all the
    // process's CPUs (per 32 MB segment it cost the pipeline 10-15 % of its rate).
    void release(size_t index)
    {
        const char *lo = nullptr, *hi = nullptr;
        {
            std::lock_guard<std::mutex> lock(rel_mu_);
            if (done_.size() < slots_.size()) done_.resize(slots_.size(), false);
            if (index >= done_.size()) return;
            done_[index] = true;
            size_t upto = released_;
            while (upto < done_.size() && done_[upto]) ++upto;
            if (upto == released_) return;
            const size_t bytes = starts_[upto] - starts_[released_];
            if (bytes < ((size_t)512 << 20) && upto != done_.size()) return;
            lo = data_ + starts_[released_];
            hi = data_ + starts_[upto];
            released_ = upto;
        }
        const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
        const uintptr_t a = ((uintptr_t)lo + page - 1) / page * page, b = (uintptr_t)hi / page * page;
        if (b > a) madvise((void*)a, b - a, MADV_DONTNEED);
    }

    // the next segment in file order; nullptr after the last one.  A segment whose batch.error is set is the last.
    std::unique_ptr<Segment> next()
    {
        std::unique_lock<std::mutex> lock(mu_);
        if (consumed_ >= slots_.size()) return nullptr;
        cv_ready_.wait(lock, [&] { return slots_[consumed_] != nullptr; });
        std::unique_ptr<Segment> seg = std::move(slots_[consumed_]);
        ++consumed_;
        if (!seg->batch.error.empty()) {
            consumed_ = slots_.size();
            stop_ = true;
        }
        lock.unlock();
        cv_room_.notify_all();
        return seg;
    }
};

// convenience for small inputs (reference FASTA files): whole-record strings
class Reader
{
    MappedFile file_;
    Parser parser_;
    std::deque<std::string> arena_;
public:
    explicit Reader(const std::string& path) : file_(path), parser_(file_.data(), file_.size()) {}
    bool is_open() const { return file_.is_open(); }
    bool read_record(std::string& id, std::string& seq)
    {
        Record r;
        arena_.clear();
        if (!parser_.next(r, arena_)) return false;
        id.assign(r.id, r.id_len);
        seq.assign(r.seq, r.seq_len);
        return true;
    }
};

// ---- parallel, ordered output ----------------------------------------------------------------------------------------
// An output file that several threads fill at once, each its own byte range, in an order fixed beforehand: reserve() hands
// out consecutive ranges (the callers take turns in file order -- a few arithmetic instructions under a lock), writer()
// gives a cursor over one range that the caller fills record by record (take(n) = the next n bytes).  Two ways to the page
// cache behind that cursor:
//   positional writes (default): the text is formatted into 2 MiB buffers from a small pool that belongs to the file; a full
//     buffer is handed to the file's OWN writer thread, which issues the pwrite() and returns the buffer to the pool.  Buffered
//     writes to one inode are serialised by its lock whoever makes them (measured on the GPU box's tmpfs: 8.4 GB/s from one
//     thread, 5.4-7.8 GB/s from two to six; 27-40 GB/s into a file per thread), so one thread per file is all a file can use --
//     and the formatting threads never queue on that lock (six of them calling pwrite() themselves swung between 14 and
//     23 M reads/s from run to run);
//   a shared mapping of the file (`use_mmap`): the text is written once, straight into the page cache; the file is mapped
//     into one reserved stretch of address space as it grows, pages come by faults (slower on tmpfs: every fresh page is a fault).
// close() waits for the writer thread and cuts the file to the bytes reserved.  Errors are sticky and reported by ok().
class OrderedOutput
{
public:
    static constexpr size_t kChunk = (size_t)2 << 20;  // positional writes: bytes formatted between two hand-overs
    static constexpr size_t kPoolBuffers = 12;         // per file: formatting threads wait while that many are queued for the writer thread

private:
    struct Job
    {
        std::vector<char>* buf;
        size_t n;
        uint64_t off;
    };
    int fd_ = -1;
    bool use_mmap_ = false;
    std::mutex mu_;
    uint64_t reserved_ = 0, file_len_ = 0;
    char* va_ = nullptr;        // mapped mode: the reserved stretch of address space, file offset 0 at va_
    uint64_t va_len_ = 0, mapped_len_ = 0;
    std::atomic<bool> failed_{false};
    std::string error_;
    // positional mode: the file's writer thread, its queue and the buffer pool
    std::thread writer_;
    std::mutex qmu_;
    std::condition_variable qcv_, pool_cv_;
    std::deque<Job> queue_;
    std::vector<std::unique_ptr<std::vector<char>>> buffers_;
    std::vector<std::vector<char>*> free_;
    size_t in_flight_ = 0;  // buffers queued for, or being written by, the writer thread: the only ones that come back by themselves
    bool stop_ = false;

    void fail(const std::string& what)
    {
        std::lock_guard<std::mutex> lock(mu_);
        if (!failed_.exchange(true)) error_ = what;
    }
    void writer_loop()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lock(qmu_);
                qcv_.wait(lock, [&] { return !queue_.empty() || stop_; });
                if (queue_.empty()) return;
                j = queue_.front();
                queue_.pop_front();
            }
            if (fd_ >= 0) pwrite_all(j.buf->data(), j.n, j.off);
            {
                std::lock_guard<std::mutex> lock(qmu_);
                free_.push_back(j.buf);
                --in_flight_;
            }
            pool_cv_.notify_one();
        }
    }
    // A caller waits only for buffers that are ON THEIR WAY BACK (queued for the writer thread).  Buffers that sit half-filled in
    // other threads' Writers return when those threads get on -- and they may be waiting for a buffer of ANOTHER file that this
    // caller holds: with two output files and more formatting threads than pool buffers, twelve threads held file A's buffers
    // and waited for B while twelve held B's and waited for A.  So when nothing is in flight the pool grows instead:
    // at most one buffer per formatting thread and file.
    std::vector<char>* acquire()
    {
        std::unique_lock<std::mutex> lock(qmu_);
        for (;;) {
            if (!free_.empty()) {
                std::vector<char>* b = free_.back();
                free_.pop_back();
                return b;
            }
            if (buffers_.size() < kPoolBuffers || in_flight_ == 0) {
                buffers_.emplace_back(new std::vector<char>(kChunk));
                return buffers_.back().get();
            }
            pool_cv_.wait(lock);
        }
    }
    void release(std::vector<char>* b)
    {
        {
            std::lock_guard<std::mutex> lock(qmu_);
            free_.push_back(b);
        }
        pool_cv_.notify_one();
    }
    void submit(std::vector<char>* b, size_t n, uint64_t off)
    {
        {
            std::lock_guard<std::mutex> lock(qmu_);
            if (!writer_.joinable()) writer_ = std::thread([this] { writer_loop(); });
            queue_.push_back(Job{b, n, off});
            ++in_flight_;
        }
        qcv_.notify_one();
    }

public:
    // cursor over one reserved range
    class Writer
    {
        friend class OrderedOutput;
        OrderedOutput* owner_ = nullptr;
        uint64_t off_ = 0;       // file offset of the next byte to go out
        uint64_t left_ = 0;      // bytes of the range not handed out yet
        char* map_cur_ = nullptr;
        std::vector<char>* buf_ = nullptr;  // positional mode: the buffer being filled (from the file's pool)
        size_t fill_ = 0;
        void flush(bool more)
        {
            if (fill_) {
                owner_->submit(buf_, fill_, off_);
                off_ += fill_;
                fill_ = 0;
                buf_ = more ? owner_->acquire() : nullptr;
            } else if (!more && buf_) {
                owner_->release(buf_);
                buf_ = nullptr;
            }
        }
    public:
        Writer() = default;
        Writer(const Writer&) = delete;
        Writer& operator=(const Writer&) = delete;
        Writer(Writer&& o) noexcept { *this = std::move(o); }
        Writer& operator=(Writer&& o) noexcept
        {
            owner_ = o.owner_; off_ = o.off_; left_ = o.left_; map_cur_ = o.map_cur_; buf_ = o.buf_; fill_ = o.fill_;
            o.owner_ = nullptr; o.buf_ = nullptr; o.map_cur_ = nullptr; o.fill_ = 0; o.left_ = 0;
            return *this;
        }
        ~Writer()
        {
            if (owner_ && !map_cur_) flush(false);  // (an abandoned range: what was filled still goes out, the buffer goes home)
        }
        // the next n bytes of the range, contiguous, to be filled before the next take()
        char* take(size_t n)
        {
            if (n > left_) throw std::runtime_error("output layout mismatch");
            left_ -= n;
            if (map_cur_) {
                char* p = map_cur_;
                map_cur_ += n;
                return p;
            }
            if (!buf_) buf_ = owner_->acquire();
            if (fill_ + n > buf_->size()) {
                flush(true);
                if (n > buf_->size()) buf_->resize(n);
            }
            char* p = buf_->data() + fill_;
            fill_ += n;
            return p;
        }
        // everything handed out has been filled: positional mode hands the rest over; false if the range was not used up
        bool finish()
        {
            if (owner_ && !map_cur_) flush(false);
            return left_ == 0;
        }
    };

    OrderedOutput() = default;
    ~OrderedOutput() { close(); }
    OrderedOutput(const OrderedOutput&) = delete;
    OrderedOutput& operator=(const OrderedOutput&) = delete;

    bool open(const std::string& path, bool use_mmap)
    {
        fd_ = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd_ < 0) return false;
        use_mmap_ = false;
        if (use_mmap) {
            // address space for the whole file, whatever it grows to (nothing is committed): the file is mapped into it piece by piece
            const uint64_t want = (uint64_t)1 << 40;
            void* p = mmap(nullptr, want, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (p != MAP_FAILED) {
                va_ = (char*)p;
                va_len_ = want;
                use_mmap_ = true;
            }
        }
        return true;
    }
    bool is_open() const { return fd_ >= 0; }
    bool ok() const { return !failed_.load(); }
    std::string error()
    {
        std::lock_guard<std::mutex> lock(mu_);
        return error_;
    }
    uint64_t bytes() const { return reserved_; }

    // the next `bytes` bytes of the file; callers take turns in output order.  Mapped mode grows the file (sparse) and its
    // mapping ahead in large steps.
    uint64_t reserve(uint64_t bytes)
    {
        std::lock_guard<std::mutex> lock(mu_);
        const uint64_t off = reserved_;
        reserved_ += bytes;
        if (use_mmap_ && reserved_ > mapped_len_) {
            uint64_t want = std::max<uint64_t>(mapped_len_ * 2, (uint64_t)1 << 30);
            while (want < reserved_) want *= 2;
            bool grown = want <= va_len_ && ftruncate(fd_, (off_t)want) == 0;
            if (grown) {
                file_len_ = want;
                void* p = mmap(va_ + mapped_len_, want - mapped_len_, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd_, (off_t)mapped_len_);
                grown = p != MAP_FAILED;
            }
            if (grown) mapped_len_ = want;
            // (not grown: ranges beyond the mapping go out by positional writes -- writer() looks at mapped_len_)
        }
        return off;
    }

    // cursor over [off, off + bytes)
    Writer writer(uint64_t off, uint64_t bytes)
    {
        Writer w;
        w.owner_ = this;
        w.off_ = off;
        w.left_ = bytes;
        bool mapped;
        {
            std::lock_guard<std::mutex> lock(mu_);
            mapped = use_mmap_ && off + bytes <= mapped_len_;
        }
        if (mapped) w.map_cur_ = va_ + off;
        return w;
    }

    void pwrite_all(const char* p, size_t n, uint64_t off)
    {
        while (n) {
            const ssize_t k = ::pwrite(fd_, p, n, (off_t)off);
            if (k <= 0) { fail("write failed"); return; }
            p += k; n -= (size_t)k; off += (uint64_t)k;
        }
    }

    void close()
    {
        {
            std::lock_guard<std::mutex> lock(qmu_);
            stop_ = true;
        }
        qcv_.notify_all();
        if (writer_.joinable()) writer_.join();  // everything queued is written first
        if (va_) munmap(va_, va_len_);
        va_ = nullptr;
        if (fd_ < 0) return;
        if (file_len_ != reserved_ && file_len_ != 0 && ftruncate(fd_, (off_t)reserved_) != 0) fail("truncate failed");
        ::close(fd_);
        fd_ = -1;
    }
};

inline void write_fasta(std::ostream& out, const char* id, size_t id_len, const char* seq, size_t seq_len)
{
    out.put('>');
    out.write(id, (std::streamsize)id_len);
    out.put('\n');
    out.write(seq, (std::streamsize)seq_len);
    out.put('\n');
}

}  // namespace seqio
