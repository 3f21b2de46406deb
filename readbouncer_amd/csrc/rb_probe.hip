// rb_probe.hip -- measurement aid, not part of the classify path: what THIS device delivers for the access pattern of the wide
// count kernels (random gathers of whole blocks from a filter's table in HBM) with no compute attached.  bench.py runs it
// on the filter it has just measured, in the same process, so that the roofline line carries a reference point from the
// same box and the same minute (`roofline.read_peak_probe`) next to the 8 TB/s spec figure.
//
// One 64-lane wave gathers rows of ROW bytes with 16 bytes per lane: a 1 KiB row is one wave instruction (config 3's
// blocks, K1 <6,2,...>), a 4 KiB row four, a 128-byte row an eighth (eight rows per instruction).  B instructions are issued
// back to back with a schedule fence before the first use -- at least what K1 keeps in flight (12 per wave at 3 waves per
// SIMD = 144 KiB per CU; the probe runs at 7-8 waves per SIMD with 12 or 24).  Row numbers come from a counter hash with a
// multiply-high range reduction: a handful of integer instructions per KiB, so that the memory system is what is timed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "rb_device.h"

namespace {

typedef unsigned long long probe_u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t probe_row(uint64_t x, uint32_t n_rows)
{
    x *= 0x9E3779B97F4A7C15ULL;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ULL;
    return __umulhi((uint32_t)(x >> 32), n_rows);
}

template <int ROW, bool NT, int B>
__global__ __launch_bounds__(256) void probe_gather_rows(const uint8_t *__restrict__ base, uint32_t n_rows, uint32_t iters,
                                                          uint64_t *__restrict__ sink)
{
    constexpr int LANES = ROW >= 1024 ? 64 : ROW / 16;   // lanes that cover one row (or one KiB of it)
    constexpr int RPI = 64 / LANES;                      // rows per wave instruction
    constexpr int IPR = ROW >= 1024 ? ROW / 1024 : 1;    // wave instructions per row
    constexpr int NROW = B / IPR;                        // rows (or row groups) per batch
    static_assert(B % IPR == 0 && NROW >= 1, "batch holds whole rows");
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int g = lane / LANES, c = lane % LANES;
    uint64_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        probe_u64x2 v[NROW][IPR];
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            const uint32_t row = probe_row((wave * iters + it) * (uint64_t)(NROW * RPI) + (uint64_t)(r * RPI + g), n_rows);
            const uint8_t *p = base + (uint64_t)row * ROW + (uint32_t)c * 16u;
#pragma unroll
            for (int k = 0; k < IPR; ++k) {
                const probe_u64x2 *src = reinterpret_cast<const probe_u64x2 *>(p + k * 1024);
                if constexpr (NT) v[r][k] = __builtin_nontemporal_load(src);
                else v[r][k] = *src;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < NROW; ++r)
#pragma unroll
            for (int k = 0; k < IPR; ++k) acc ^= v[r][k].x ^ v[r][k].y;
    }
    if (acc == 0x0123456789ABCDEFULL) sink[0] = acc;  // never true in practice; keeps the loads alive
}

template <int ROW, bool NT, int B>
hipError_t probe_launch(const uint8_t *base, uint32_t n_rows, uint32_t iters, uint32_t blocks, uint64_t *sink, hipStream_t st)
{
    hipLaunchKernelGGL((probe_gather_rows<ROW, NT, B>), dim3(blocks), dim3(256), 0, st, base, n_rows, iters, sink);
    return hipGetLastError();
}

template <int ROW>
hipError_t probe_dispatch(bool nt, int b, const uint8_t *base, uint32_t n_rows, uint32_t iters, uint32_t blocks, uint64_t *sink,
                          hipStream_t st)
{
    if (nt) return b >= 24 ? probe_launch<ROW, true, 24>(base, n_rows, iters, blocks, sink, st) : probe_launch<ROW, true, 12>(base, n_rows, iters, blocks, sink, st);
    return b >= 24 ? probe_launch<ROW, false, 24>(base, n_rows, iters, blocks, sink, st) : probe_launch<ROW, false, 12>(base, n_rows, iters, blocks, sink, st);
}

}  // namespace

// the probe on a bare block of device memory (the current device): rb_dibf_probe_read_peak below, and the placement trials of
// rb_engine.hip (dibf_alloc), which probe a table before it belongs to a filter
int rb::probe_read_peak_raw(const void *table, uint64_t table_bytes, uint32_t row_bytes, int nontemporal, uint32_t loads_in_flight,
                            double target_ms, double *gbps_out, double *ms_out)
{
    if (!table || !gbps_out) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    if (row_bytes != 128 && row_bytes != 1024 && row_bytes != 4096) return rb::fail(RB_ERR_INVALID_ARG, "probe rows are 128, 1024 or 4096 bytes");
    if (table_bytes < row_bytes || table_bytes / row_bytes >= (1ull << 32)) return rb::fail(RB_ERR_INVALID_ARG, "probe table size");
    hipError_t e = hipSuccess;
    const uint8_t *base = (const uint8_t *)table;
    const uint32_t n_rows = (uint32_t)(table_bytes / row_bytes);
    const int b = loads_in_flight >= 24 ? 24 : 12;
    const uint32_t blocks = 256u * 32u;  // 32 768 waves: 32 per SIMD, several rounds of residency
    uint64_t *sink = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t st = nullptr;
    e = hipMalloc((void **)&sink, 8);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ev0);
    if (e == hipSuccess) e = hipEventCreate(&ev1);
    auto run = [&](uint32_t iters, float *ms) -> hipError_t {
        hipError_t r = hipEventRecord(ev0, st);
        if (r == hipSuccess) {
            r = row_bytes == 128    ? probe_dispatch<128>(nontemporal != 0, b, base, n_rows, iters, blocks, sink, st)
                : row_bytes == 1024 ? probe_dispatch<1024>(nontemporal != 0, b, base, n_rows, iters, blocks, sink, st)
                                    : probe_dispatch<4096>(nontemporal != 0, b, base, n_rows, iters, blocks, sink, st);
        }
        if (r == hipSuccess) r = hipEventRecord(ev1, st);
        if (r == hipSuccess) r = hipEventSynchronize(ev1);
        if (r == hipSuccess) r = hipEventElapsedTime(ms, ev0, ev1);
        return r;
    };
    double best_gbps = 0.0, best_ms = 0.0;
    if (e == hipSuccess) {
        float ms = 0.f;
        e = run(8, &ms);  // code object, clocks
        if (e == hipSuccess) e = run(32, &ms);
        // length of the timed runs from the calibration: short kernels under-read (ramp-up and the tail are a few per cent of 10 ms)
        const double want = target_ms > 0 ? target_ms : 200.0;
        uint32_t iters = 32;
        if (e == hipSuccess && ms > 0.f) iters = (uint32_t)std::min<double>(1 << 20, std::max<double>(32.0, 32.0 * want / ms));
        const double bytes_per_iter = (double)blocks * 4.0 * (double)b * 1024.0;  // every wave instruction moves 64 x 16 bytes
        for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {
            e = run(iters, &ms);
            if (e == hipSuccess && ms > 0.f) {
                const double gbps = bytes_per_iter * iters / (ms * 1e6);
                if (gbps > best_gbps) { best_gbps = gbps; best_ms = ms; }
            }
        }
    }
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (st) (void)hipStreamDestroy(st);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return rb::fail(RB_ERR_HIP, std::string("read-peak probe: ") + hipGetErrorString(e));
    *gbps_out = best_gbps;
    if (ms_out) *ms_out = best_ms;
    return RB_OK;
}

extern "C" int rb_dibf_probe_read_peak(rb_dibf *f, uint64_t table_bytes, uint32_t row_bytes, int nontemporal, uint32_t loads_in_flight,
                                       double target_ms, double *gbps_out, double *ms_out)
{
    if (!f || !gbps_out) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    rb_ibf_info g;
    int rc = rb_dibf_get_info(f, &g);
    if (rc != RB_OK) return rc;
    const uint64_t whole = g.n_blocks * rb_dibf_device_stride(f) * 8;  // the table as it lies in HBM
    if (table_bytes == 0) table_bytes = whole;
    if (table_bytes > whole) return rb::fail(RB_ERR_INVALID_ARG, "probe table_bytes beyond the filter's table");  // (would gather out of bounds)
    hipError_t e = hipSetDevice(rb_dibf_device(f));
    if (e != hipSuccess) return rb::fail(e == hipErrorNoDevice ? RB_ERR_NO_DEVICE : RB_ERR_HIP, hipGetErrorString(e));
    return rb::probe_read_peak_raw(rb_dibf_device_words(f), table_bytes, row_bytes, nontemporal, loads_in_flight, target_ms, gbps_out, ms_out);
}
