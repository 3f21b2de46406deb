// hbm_peak.hip -- what this MI355X delivers for the access patterns of the IBF path, with no compute attached:
//   (1) streaming read of an 8 GiB buffer (16 B per lane),
//   (2) random gathers of ROW-byte rows (128 B = config 2's blocks, 1 KiB = config 3's, 4 KiB ~ GRCh38 at F=100000)
//       from the same 8 GiB buffer, 24 row loads in flight per wave.
// The numbers are the ceilings the roofline fractions in DESIGN.md are compared with (next to the 8 TB/s spec).
// build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 profiles/hbm_peak.hip -o /tmp/hbm_peak && /tmp/hbm_peak
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// U independent 16-byte loads in flight per lane; consecutive waves read consecutive 1 KiB pieces
template <int U, bool NT>
__global__ void stream_read(const u64x2 *__restrict__ p, size_t n, uint64_t *out)
{
    uint64_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u64x2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y;
    }
    for (; i < n; i += stride) {
        const u64x2 v = p[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x123456789ULL) out[0] = acc;
}

// ROW bytes per gathered row; LANES = ROW/16 lanes cover one row, 64/LANES rows per wave instruction
template <int ROW, bool NT>
__global__ void gather_rows(const uint8_t *__restrict__ base, uint64_t n_rows, uint32_t iters, uint64_t *out)
{
    constexpr int LANES = ROW / 16 > 64 ? 64 : ROW / 16;
    constexpr int PER_INSTR = 64 / LANES;
    constexpr int CHUNKS = ROW / 16 > 64 ? ROW / 1024 : 1;  // rows above 1 KiB: several instructions per row
    constexpr int BATCH = 24 / CHUNKS;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int g = lane / LANES, c = lane % LANES;
    uint64_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        u64x2 v[BATCH][CHUNKS];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const uint64_t row = mix((wave * iters + it) * (uint64_t)(BATCH * PER_INSTR) + b * PER_INSTR + g) % n_rows;
#pragma unroll
            for (int k = 0; k < CHUNKS; ++k) {
                const u64x2 *src = reinterpret_cast<const u64x2 *>(base + row * ROW + k * 1024 + c * 16);
                v[b][k] = NT ? __builtin_nontemporal_load(src) : *src;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < BATCH; ++b)
#pragma unroll
            for (int k = 0; k < CHUNKS; ++k) acc ^= v[b][k].x ^ v[b][k].y;
    }
    if (acc == 0x123456789ULL) out[0] = acc;
}

template <int ROW, bool NT>
static void run_gather(const uint8_t *d, size_t bytes, uint64_t *out, const char *label)
{
    constexpr int LANES = ROW / 16 > 64 ? 64 : ROW / 16;
    constexpr int CHUNKS = ROW / 16 > 64 ? ROW / 1024 : 1;
    constexpr int BATCH = 24 / CHUNKS;
    const uint64_t n_rows = bytes / ROW;
    const uint32_t iters = 64;
    const int waves = 256 * 16 * 8;  // 32 k waves
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((gather_rows<ROW, NT>), dim3(waves / 4), dim3(256), 0, 0, d, n_rows, iters, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double moved = (double)waves * iters * BATCH * (64 / LANES) * (double)ROW;
    printf("gather %4d-B rows %s over %.1f GiB: %7.0f GB/s\n", ROW, label, bytes / 1073741824.0, moved / ms / 1e6);
}

template <int U, bool NT>
static void run_stream(const uint8_t *d, size_t bytes, uint64_t *out, int blocks, const char *label)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((stream_read<U, NT>), dim3(blocks), dim3(256), 0, 0, (const u64x2 *)d, bytes / 16, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("stream read of 8 GiB (%s): %7.0f GB/s\n", label, bytes / ms / 1e6);
}

int main()
{
    const size_t bytes = (size_t)8 << 30;
    uint8_t *d;
    uint64_t *out;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 0x5a, bytes);
    run_stream<1, true>(d, bytes, out, 256 * 16, "1 load in flight per lane, nt");
    run_stream<8, true>(d, bytes, out, 256 * 16, "8 loads in flight per lane, nt");
    run_stream<8, false>(d, bytes, out, 256 * 16, "8 loads in flight per lane, default");
    run_stream<16, true>(d, bytes, out, 256 * 8, "16 loads in flight per lane, nt");
    run_gather<128, false>(d, bytes, out, "(default)");
    run_gather<128, true>(d, bytes, out, "(nt)     ");
    run_gather<128, false>(d, (size_t)390 << 20, out, "(default)");  // config 2's table size: Infinity Cache helps
    run_gather<1024, false>(d, bytes, out, "(default)");
    run_gather<1024, true>(d, bytes, out, "(nt)     ");
    run_gather<4096, true>(d, bytes, out, "(nt)     ");
    return 0;
}
