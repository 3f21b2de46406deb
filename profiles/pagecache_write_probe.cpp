// pagecache_write_probe.cpp -- how fast bytes enter the page cache of one directory: the floor under the CLI's output side.
// T threads write BYTES in all with pwrite() of 2 MiB buffers (already filled, cache resident -- what the CLI's classifier threads
// do with formatted FASTA text), either into disjoint ranges of ONE file (unclassified.fasta: buffered writes to one inode are
// serialised by its lock) or into a file each.  Prints GB/s per setting.
//   g++ -O2 -std=c++17 -pthread profiles/pagecache_write_probe.cpp -o /tmp/pcw && /tmp/pcw /dev/shm/x 2700000000
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double run(const std::string &dir, size_t bytes, int threads, bool one_file)
{
    const size_t chunk = (size_t)2 << 20;
    std::vector<int> fds;
    for (int t = 0; t < (one_file ? 1 : threads); ++t) {
        const std::string p = dir + "/pcw_probe_" + std::to_string(t);
        fds.push_back(::open(p.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644));
        if (fds.back() < 0) { std::perror("open"); std::exit(1); }
    }
    const size_t per = bytes / (size_t)threads / chunk * chunk;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            std::vector<char> buf(chunk, (char)('A' + t));
            const int fd = fds[one_file ? 0 : t];
            // ranges interleaved like the CLI's segments: thread t writes every threads-th stretch of 32 MiB
            const size_t stretch = (size_t)32 << 20;
            size_t done = 0, k = 0;
            while (done < per) {
                const size_t base = one_file ? (k * (size_t)threads + (size_t)t) * stretch : k * stretch;
                for (size_t o = 0; o < stretch && done < per; o += chunk, done += chunk)
                    if (::pwrite(fd, buf.data(), chunk, (off_t)(base + o)) != (ssize_t)chunk) { std::perror("pwrite"); std::exit(1); }
                ++k;
            }
        });
    for (auto &x : th) x.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (size_t i = 0; i < fds.size(); ++i) {
        ::close(fds[i]);
        ::unlink((dir + "/pcw_probe_" + std::to_string(i)).c_str());
    }
    return (double)(per * (size_t)threads) / s / 1e9;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s <dir> <bytes>\n", argv[0]); return 1; }
    const std::string dir = argv[1];
    const size_t bytes = (size_t)std::strtoull(argv[2], nullptr, 10);
    run(dir, bytes / 4, 1, true);  // warm-up
    for (int t : {1, 2, 4, 6}) std::printf("page-cache writes, ONE file, %d thread(s): %.2f GB/s\n", t, run(dir, bytes, t, true));
    for (int t : {4, 6}) std::printf("page-cache writes, a file per thread, %d threads: %.2f GB/s\n", t, run(dir, bytes, t, false));
    return 0;
}
