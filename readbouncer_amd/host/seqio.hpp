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

#include <cstring>
#include <deque>
#include <ostream>
#include <stdexcept>
#include <string>
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

inline void write_fasta(std::ostream& out, const char* id, size_t id_len, const char* seq, size_t seq_len)
{
    out.put('>');
    out.write(id, (std::streamsize)id_len);
    out.put('\n');
    out.write(seq, (std::streamsize)seq_len);
    out.put('\n');
}

}  // namespace seqio
