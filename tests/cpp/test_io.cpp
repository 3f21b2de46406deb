// tests/cpp/test_io.cpp -- csrc/rb_io.h (the threads of one filter load) on a CPU, under ASan / UBSan / TSan (profiles/sanitize_cpu.sh)
// and in the CPU suite (tests/test_capi_cpu.py).  What round 5's version could not do: a thousand chunks through ONE set of threads,
// a short file that stops the other readers early, a file that shrinks while it is read, a gang that could not get its threads.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "../../readbouncer_amd/csrc/rb_io.h"

static std::vector<unsigned char> pattern(size_t n, unsigned seed)
{
    std::vector<unsigned char> v(n);
    unsigned x = seed * 2654435761u + 1u;
    for (size_t i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        v[i] = (unsigned char)(x >> 24);
    }
    return v;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const std::string path = dir + "/rb_io_test_" + std::to_string((long)getpid()) + ".bin";
    const size_t bytes = ((size_t)40 << 20) + 12345;  // five parts of a gang of five, a ragged tail
    const std::vector<unsigned char> src = pattern(bytes, 7);
    {
        FILE *fp = std::fopen(path.c_str(), "wb");
        assert(fp && std::fwrite(src.data(), 1, bytes, fp) == bytes);
        std::fclose(fp);
    }
    const int fd = ::open(path.c_str(), O_RDONLY);
    assert(fd >= 0);

    // 1. one gang, many chunks: the same threads serve every call (round 5 made and joined threads per chunk)
    {
        rb::IoGang gang(5);
        assert(gang.size() >= 1 && gang.size() <= 5);
        std::vector<unsigned char> dst(bytes);
        for (int rep = 0; rep < 40; ++rep) {
            std::memset(dst.data(), 0, 4096);
            assert(gang.pread(fd, 0, dst.data(), bytes));
            assert(std::memcmp(dst.data(), src.data(), bytes) == 0);
        }
        // offsets and sizes that are not page multiples, sizes below the one-thread limit
        for (size_t off : {(size_t)0, (size_t)1, (size_t)4097, (size_t)(9u << 20) + 3}) {
            for (size_t n : {(size_t)0, (size_t)1, (size_t)4096, (size_t)(8u << 20), (size_t)(8u << 20) + 1, bytes - off}) {
                if (off + n > bytes) continue;
                std::vector<unsigned char> d(n + 1, 0xAB);
                assert(gang.pread(fd, (off_t)off, d.data(), n));
                assert(std::memcmp(d.data(), src.data() + off, n) == 0 && d[n] == 0xAB);
            }
        }
        std::vector<unsigned char> copy(bytes + 1, 0xCD);
        gang.memcpy(copy.data(), src.data(), bytes);
        assert(std::memcmp(copy.data(), src.data(), bytes) == 0 && copy[bytes] == 0xCD);
        // 2. a request beyond the end of the file: a short read, reported, the gang still usable afterwards
        std::vector<unsigned char> big(bytes + ((size_t)16 << 20));
        assert(!gang.pread(fd, 0, big.data(), big.size()));
        assert(!gang.pread(fd, (off_t)bytes - 10, big.data(), (size_t)9 << 20));
        assert(gang.pread(fd, 0, dst.data(), bytes) && std::memcmp(dst.data(), src.data(), bytes) == 0);
    }
    // 3. a file that shrinks while it is being read: some part meets the new end; false, no crash, no hang
    {
        rb::IoGang gang(4);
        std::vector<unsigned char> dst(bytes);
        assert(::truncate(path.c_str(), (off_t)(bytes / 2)) == 0);
        assert(!gang.pread(fd, 0, dst.data(), bytes));
        assert(gang.pread(fd, 0, dst.data(), bytes / 2) && std::memcmp(dst.data(), src.data(), bytes / 2) == 0);
    }
    // 4. gangs of one (no worker threads at all) and of more threads than parts
    {
        rb::IoGang one(1), many(16);
        assert(one.size() == 1);
        std::vector<unsigned char> dst(bytes / 2);
        assert(one.pread(fd, 0, dst.data(), dst.size()) && std::memcmp(dst.data(), src.data(), dst.size()) == 0);
        assert(many.pread(fd, 0, dst.data(), dst.size()) && std::memcmp(dst.data(), src.data(), dst.size()) == 0);
        int hits[7] = {0, 0, 0, 0, 0, 0, 0};
        many.run(7, [&](int p) { hits[p] += 1; });  // (each part exactly once; TSan: distinct elements)
        for (int h : hits) assert(h == 1);
        many.run(0, [&](int) { assert(false); });
    }
    // 5. an error from the descriptor itself
    {
        rb::IoGang gang(3);
        std::vector<unsigned char> dst((size_t)9 << 20);
        assert(!gang.pread(-1, 0, dst.data(), dst.size()));
    }
    ::close(fd);
    ::unlink(path.c_str());
    std::puts("rb_io: ok");
    return 0;
}
