// Host-side bulk IO of the filter paths (IBF::load_filter at GRCh38 scale): one thread moves 4-5 GB/s out of the page cache or between two
// host buffers, PCIe takes ten times that -- so reads from a file and copies into page-locked staging are spread over a few threads.
#pragma once
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstddef>
#include <cstring>
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

inline bool pread_full(int fd, off_t off, char *dst, size_t bytes)
{
    while (bytes) {
        const ssize_t r = ::pread(fd, dst, bytes, off);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return false;  // error or end of file before `bytes`
        dst += r;
        off += r;
        bytes -= (size_t)r;
    }
    return true;
}

// dst[0, bytes) <- fd[off, off + bytes), split over io_threads(bytes) threads (page-aligned parts); false on a short read or an error
inline bool pread_parallel(int fd, off_t off, void *dst, size_t bytes)
{
    const int t = io_threads(bytes);
    if (t <= 1) return pread_full(fd, off, (char *)dst, bytes);
    const size_t part = ((bytes + (size_t)t - 1) / (size_t)t + 4095) & ~(size_t)4095;
    std::atomic<bool> ok{true};
    std::vector<std::thread> th;
    for (int i = 0; i < t; ++i) {
        const size_t b = (size_t)i * part;
        if (b >= bytes) break;
        const size_t n = std::min(part, bytes - b);
        th.emplace_back([=, &ok] {
            if (!pread_full(fd, off + (off_t)b, (char *)dst + b, n)) ok.store(false);
        });
    }
    for (std::thread &x : th) x.join();
    return ok.load();
}

inline void memcpy_parallel(void *dst, const void *src, size_t bytes)
{
    const int t = io_threads(bytes);
    if (t <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t part = ((bytes + (size_t)t - 1) / (size_t)t + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (int i = 0; i < t; ++i) {
        const size_t b = (size_t)i * part;
        if (b >= bytes) break;
        const size_t n = std::min(part, bytes - b);
        th.emplace_back([=] { std::memcpy((char *)dst + b, (const char *)src + b, n); });
    }
    for (std::thread &x : th) x.join();
}

}  // namespace rb
