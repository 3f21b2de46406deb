// Host-side bulk IO of the filter paths (IBF::load_filter at GRCh38 scale, src/IBF/IBFBuild.cpp:329-396): one thread moves 4-5 GB/s out
// of the page cache or between two host buffers, PCIe takes ten times that -- so reads from a file and copies into page-locked staging
// are spread over a few threads.
//
// IoGang: the threads of ONE load (rb_ibf_open, rb_dibf_open, rb_dibf_upload), created once and parked between the chunks of that load --
// round 5 created and joined up to eight std::threads per 64 MiB chunk (a thousand creations for an 8 GiB file), and a std::thread
// constructor that threw after the first one had started destroyed a vector of joinable threads: std::terminate.  Here
//  * a worker that cannot be created only makes the gang smaller (the calling thread always takes a share itself);
//  * the destructor stops and joins whatever was started, on every path out of the load, exceptions included;
//  * a part is read in pieces of 4 MiB and every piece looks at the shared `failed` flag first: after the first short read or error
//    the other threads stop within one piece instead of reading a file that is known to be bad to its end.
// Plain C++ (no HIP): rb_host.cpp and rb_engine.hip include it; tests/cpp/test_io.cpp runs it under ASan / UBSan / TSan.
#pragma once
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstddef>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

namespace rb {

inline int io_threads(size_t bytes)
{
    if (bytes < ((size_t)8 << 20)) return 1;
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t by_size = bytes >> 22;  // at least 4 MiB per thread
    return (int)std::max<size_t>(1, std::min<size_t>({(size_t)8, hw ? (size_t)hw : (size_t)1, by_size}));
}

// dst[0, bytes) <- fd[off, off + bytes); false on an error or when the file ends first.  `stop` (optional): give up between pieces
// once somebody else has failed.
inline bool pread_full(int fd, off_t off, char *dst, size_t bytes, const std::atomic<bool> *stop = nullptr)
{
    constexpr size_t kPiece = (size_t)4 << 20;
    while (bytes) {
        if (stop && stop->load(std::memory_order_relaxed)) return false;
        const ssize_t r = ::pread(fd, dst, std::min(bytes, kPiece), off);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return false;  // error or end of file before `bytes`
        dst += r;
        off += r;
        bytes -= (size_t)r;
    }
    return true;
}

class IoGang {
public:
    // `threads` in all, the calling thread included (threads - 1 workers are started; fewer if the system refuses)
    explicit IoGang(int threads)
    {
        for (int i = 1; i < threads; ++i) {
            try {
                workers_.emplace_back([this] { work(); });
            } catch (...) {  // std::system_error: no more threads -- a smaller gang, not a dead process
                break;
            }
        }
    }
    IoGang(const IoGang &) = delete;
    IoGang &operator=(const IoGang &) = delete;
    ~IoGang()
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        wake_.notify_all();
        for (std::thread &t : workers_)
            if (t.joinable()) t.join();
    }
    int size() const { return (int)workers_.size() + 1; }

    // fn(part) for part = 0 .. parts - 1, spread over the gang (the caller works too); returns when all have run
    void run(int parts, const std::function<void(int)> &fn)
    {
        if (parts <= 0) return;
        if (workers_.empty() || parts == 1) {
            for (int p = 0; p < parts; ++p) fn(p);
            return;
        }
        {
            std::lock_guard<std::mutex> lock(mu_);
            fn_ = &fn;
            next_ = 0;
            parts_ = parts;
            pending_ = parts;
            ++generation_;
        }
        wake_.notify_all();
        drain();  // the calling thread takes parts like everybody else
        std::unique_lock<std::mutex> lock(mu_);
        done_.wait(lock, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

    // dst[0, bytes) <- fd[off, off + bytes) in page-aligned parts; false on a short read or an error (the other parts stop at their
    // next piece)
    bool pread(int fd, off_t off, void *dst, size_t bytes)
    {
        const int t = std::min(size(), io_threads(bytes));
        if (t <= 1) return pread_full(fd, off, (char *)dst, bytes);
        const size_t part = ((bytes + (size_t)t - 1) / (size_t)t + 4095) & ~(size_t)4095;
        const int parts = (int)((bytes + part - 1) / part);
        std::atomic<bool> failed{false};
        run(parts, [&](int i) {
            const size_t b = (size_t)i * part;
            const size_t n = std::min(part, bytes - b);
            if (!pread_full(fd, off + (off_t)b, (char *)dst + b, n, &failed)) failed.store(true);
        });
        return !failed.load();
    }

    void memcpy(void *dst, const void *src, size_t bytes)
    {
        const int t = std::min(size(), io_threads(bytes));
        if (t <= 1) {
            std::memcpy(dst, src, bytes);
            return;
        }
        const size_t part = ((bytes + (size_t)t - 1) / (size_t)t + 4095) & ~(size_t)4095;
        const int parts = (int)((bytes + part - 1) / part);
        run(parts, [&](int i) {
            const size_t b = (size_t)i * part;
            std::memcpy((char *)dst + b, (const char *)src + b, std::min(part, bytes - b));
        });
    }

private:
    void drain()
    {
        for (;;) {
            const std::function<void(int)> *fn = nullptr;
            int p = -1;
            {
                std::lock_guard<std::mutex> lock(mu_);
                if (!fn_ || next_ >= parts_) return;
                fn = fn_;
                p = next_++;
            }
            (*fn)(p);
            bool last = false;
            {
                std::lock_guard<std::mutex> lock(mu_);
                last = --pending_ == 0;
            }
            if (last) done_.notify_all();
        }
    }
    void work()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(mu_);
                wake_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            drain();
        }
    }

    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable wake_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    int next_ = 0, parts_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

}  // namespace rb
