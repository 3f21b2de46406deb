// gather_probe.hip -- what limits random 8-byte gathers from a 10-20 MB table (the narrow-filter geometry: one- and
// two-word IBF blocks) on this MI355X, and what the candidate ways around it would deliver.  No compute attached.
//   full   : every lane gathers from the whole table (the round-1 kernel's pattern; L2 holds 4 MiB / table of it)
//   xcd    : every lane gathers only from the eighth of the table that belongs to ITS XCD (HW_REG_XCC_ID): the ceiling
//            of a scheme that routes each lookup to the XCD whose L2 owns the slice
//   lds    : every workgroup copies a 128 KiB slice into LDS and gathers from there (ds_read_b64): the ceiling of a
//            table-in-LDS scheme
//   stream : coalesced 4-byte read + 8-byte write per item (the routing traffic such schemes pay per lookup)
// Variants: element 8 or 16 bytes; load policy plain / nt / sc1 (agent scope) / sc0 sc1 (system scope); allocation
// hipMalloc / hipDeviceMallocUncached / hipDeviceMallocFinegrained.
// build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 profiles/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe all
// single case for a counter pass: /tmp/gather_probe one <mode> <table MiB> <esz> <policy> <alloc>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

template <int POLICY>
__device__ __forceinline__ uint64_t ld8(const uint64_t *p)
{
    if constexpr (POLICY == 1) return __builtin_nontemporal_load(p);
    else if constexpr (POLICY == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (POLICY == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else return *p;
}

__device__ __forceinline__ uint32_t xcc_id()
{
    // s_getreg_b32 hwreg(HW_REG_XCC_ID (20), offset 0, size 4)
    return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
}

// 24 gathers in flight per lane (8 steps x 3 "hash functions"), like the count kernel's one-lane-per-block form
template <int ESZ, int POLICY, bool XCD>
__global__ __launch_bounds__(256) void gather(const uint64_t *__restrict__ table, uint32_t n_elems, uint32_t iters,
                                              uint64_t *out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lo = 0, span = n_elems;
    if constexpr (XCD) {
        span = n_elems / 8;
        lo = xcc_id() * span;
    }
    uint64_t acc = 0;
    uint64_t s = mix(tid + 1);
    for (uint32_t it = 0; it < iters; ++it) {
        uint64_t v[8][3][ESZ / 8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                s = s * 6364136223846793005ULL + 1442695040888963407ULL;
                const uint32_t idx = lo + (uint32_t)(((s >> 32) * (uint64_t)span) >> 32);
                const uint64_t *p = table + (size_t)idx * (ESZ / 8);
                if constexpr (ESZ == 8) {
                    v[u][h][0] = ld8<POLICY>(p);
                } else {
                    u64x2 q;
                    if constexpr (POLICY == 1) q = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(p));
                    else q = *reinterpret_cast<const u64x2 *>(p);
                    v[u][h][0] = q.x;
                    v[u][h][1] = q.y;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < ESZ / 8; ++e) acc += v[u][0][e] & v[u][1][e] & v[u][2][e];
    }
    if (acc == 0x123456789ULL) out[0] = acc;
}

// "phased": the table is cut into S slices and the whole chip walks them in step with the 100 MHz wall clock -- during
// window w (dt ticks long) every wave gathers only those of its 24 buffered lookups that fall into slice w % S, so each
// XCD's L2 only has to hold one slice (plus stragglers) at a time.  A wave that is ahead of the clock sleeps until its
// next window opens, a wave that is behind never waits.  No data moves anywhere new: block numbers and the AND
// accumulators stay in registers, exactly like the one-lane-per-block count kernel holds them.
// PF > 0: on entering a window every wave also touches PF x 64 lines of the NEXT slice (one dword per lane, 128 bytes
// apart), a different piece per wave of the XCD, so that the slice is already in L2 when its window opens and the fabric
// streams the table in the background instead of serving demand misses.
template <int S, int NBUF, int PF>
__global__ __launch_bounds__(256) void gather_phased(const uint64_t *__restrict__ table, uint32_t n_elems, uint32_t iters,
                                                     uint32_t dt, uint64_t *out, uint32_t slack = 0)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per_slice = (n_elems + S - 1) / S;
    uint64_t acc = 0;
    uint64_t s = mix(tid + 1);
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t idx[NBUF];
        uint32_t sl[NBUF];
        uint64_t a[NBUF / 3];
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            s = s * 6364136223846793005ULL + 1442695040888963407ULL;
            idx[u] = (uint32_t)(((s >> 32) * (uint64_t)n_elems) >> 32);
            sl[u] = idx[u] / per_slice;
        }
#pragma unroll
        for (int u = 0; u < NBUF / 3; ++u) a[u] = ~0ULL;
        const uint64_t w0 = wall_clock64() / dt;
#pragma unroll 1
        for (uint32_t q = 0; q < (uint32_t)S; ++q) {
            const uint64_t w = w0 + q;
            // slack > 0 (round 3): window w opens `slack` ticks early, i.e. a wave may run ahead of the clock by that much and
            // two slices are live in L2 at a time -- the bursts at the window boundaries overlap instead of queueing
            while (wall_clock64() + slack < w * dt) __builtin_amdgcn_s_sleep(4);
            const uint32_t p = (uint32_t)(w % S);
            uint32_t pf[PF > 0 ? PF : 1];
            if constexpr (PF > 0) {
                const uint32_t lines = per_slice / 16;              // 128-byte lines per slice
                const uint32_t pieces = (lines + 63) / 64;
                const uint32_t me = (blockIdx.x / 8) * 4 + (threadIdx.x >> 6);  // wave number within "its" XCD (round-robin placement)
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    const uint32_t piece = (me * PF + k + (uint32_t)w * 7u) % pieces;
                    uint32_t line = piece * 64 + (threadIdx.x & 63);
                    line = line < lines ? line : lines - 1;
                    const uint64_t e = (uint64_t)((p + 1) % S) * per_slice + (uint64_t)line * 16;
                    pf[k] = reinterpret_cast<const uint32_t *>(table + (e < n_elems ? e : 0))[0];
                }
            }
            uint64_t v[NBUF];
#pragma unroll
            for (int u = 0; u < NBUF; ++u) {
                v[u] = ~0ULL;
                if (sl[u] == p) v[u] = table[idx[u]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NBUF; ++u) a[u / 3] &= v[u];
            if constexpr (PF > 0) {
#pragma unroll
                for (int k = 0; k < PF; ++k) acc ^= pf[k];
            }
        }
#pragma unroll
        for (int u = 0; u < NBUF / 3; ++u) acc += a[u];
    }
    if (acc == 0x123456789ULL) out[0] = acc;
}

// "phased pump" (round 3): the prefetch of the next slice done by DEDICATED waves instead of by the gathering waves (whose
// prefetch loads queue in the same in-order vmcnt as their gathers: the PF variants above lose 30-50 %).  The first
// `n_pump` workgroups of the grid -- dealt round-robin over the XCDs, so n_pump / 8 of them per XCD -- never gather: during
// window w they touch every 128-byte line of slice w+1 (their XCD's L2 then holds it when its window opens) and leave when
// the last gathering workgroup has signed off.  The gathering workgroups are those of "phased" with PF = 0.
template <int S, int NBUF>
__global__ __launch_bounds__(256) void gather_phased_pump(const uint64_t *__restrict__ table, uint32_t n_elems, uint32_t iters,
                                                          uint32_t dt, uint64_t *out, uint32_t n_pump, uint32_t *done,
                                                          uint32_t lead)
{
    const uint32_t per_slice = (n_elems + S - 1) / S;
    if (blockIdx.x < n_pump) {
        const uint32_t lines = per_slice / 16;
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t me = (blockIdx.x / 8) * 4 + (threadIdx.x >> 6), n_me = (n_pump / 8) * 4;
        const uint32_t n_workers = gridDim.x - n_pump;
        uint32_t acc = 0;
        uint64_t w = wall_clock64() / dt;
        for (uint32_t guard = 0; guard < (1u << 22); ++guard) {
            // `lead` = which slice to pull during window w: w + 1 (the next one), pulled from the start of window w
            const uint32_t p1 = (uint32_t)((w + lead) % S);
            for (uint32_t line = me * 64 + lane; line < lines; line += n_me * 64) {
                const uint64_t e = (uint64_t)p1 * per_slice + (uint64_t)line * 16;
                acc ^= reinterpret_cast<const uint32_t *>(table + (e < n_elems ? e : 0))[0];
            }
            if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_workers) break;
            while (wall_clock64() / dt <= w) __builtin_amdgcn_s_sleep(8);
            w = wall_clock64() / dt;
        }
        if (acc == 0x12345u) out[1] = acc;
        return;
    }
    const uint64_t tid = (uint64_t)(blockIdx.x - n_pump) * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    uint64_t s = mix(tid + 1);
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t idx[NBUF];
        uint32_t sl[NBUF];
        uint64_t a[NBUF / 3];
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            s = s * 6364136223846793005ULL + 1442695040888963407ULL;
            idx[u] = (uint32_t)(((s >> 32) * (uint64_t)n_elems) >> 32);
            sl[u] = idx[u] / per_slice;
        }
#pragma unroll
        for (int u = 0; u < NBUF / 3; ++u) a[u] = ~0ULL;
        const uint64_t w0 = wall_clock64() / dt;
#pragma unroll 1
        for (uint32_t q = 0; q < (uint32_t)S; ++q) {
            const uint64_t w = w0 + q;
            while (wall_clock64() / dt < w) __builtin_amdgcn_s_sleep(4);
            const uint32_t p = (uint32_t)(w % S);
            uint64_t v[NBUF];
#pragma unroll
            for (int u = 0; u < NBUF; ++u) {
                v[u] = ~0ULL;
                if (sl[u] == p) v[u] = table[idx[u]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NBUF; ++u) a[u / 3] &= v[u];
        }
#pragma unroll
        for (int u = 0; u < NBUF / 3; ++u) acc += a[u];
    }
    if (acc == 0x123456789ULL) out[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// "phased buf" (round 3): the predication of the phased gathers done by the BOUNDS CHECK of a buffer resource instead of by
// exec masks -- per window the wave points a raw buffer descriptor at the slice of the moment (base = slice start,
// num_records = slice bytes) and issues every lookup as buffer_load_dwordx2 with offset (lookup - slice start): lanes whose
// lookup lies in another slice are out of range, make no memory access and get 0 back.  No compare, no saveexec, no branch
// around a load: one subtract and one load per lookup and window, and an OR to accumulate (the table would be stored
// complemented, so that the AND over the h words becomes an OR and the zeros of out-of-range lanes are neutral).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int S, int NBUF>
__global__ __launch_bounds__(256) void gather_phased_buf(const uint64_t *__restrict__ table, uint32_t n_elems, uint32_t iters,
                                                         uint32_t dt, uint64_t *out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per_slice = (n_elems + S - 1) / S;
    const uint32_t slice_bytes = per_slice * 8;
    uint64_t acc = 0;
    uint64_t s = mix(tid + 1);
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t off[NBUF];
        uint64_t a[NBUF / 3];
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            s = s * 6364136223846793005ULL + 1442695040888963407ULL;
            off[u] = (uint32_t)(((s >> 32) * (uint64_t)n_elems) >> 32) * 8u;
        }
#pragma unroll
        for (int u = 0; u < NBUF / 3; ++u) a[u] = 0;
        const uint64_t w0 = wall_clock64() / dt;
#pragma unroll 1
        for (uint32_t q = 0; q < (uint32_t)S; ++q) {
            const uint64_t w = w0 + q;
            while (wall_clock64() / dt < w) __builtin_amdgcn_s_sleep(4);
            const uint32_t p = (uint32_t)(w % S);
            const uint32_t start = p * slice_bytes;
            const uint32_t recs = min(slice_bytes, n_elems * 8u - start);
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)table + start), 0, (int)recs, 0x00020000);
            u32x2_t v[NBUF];
#pragma unroll
            for (int u = 0; u < NBUF; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, off[u] - start, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NBUF; ++u) a[u / 3] |= ((uint64_t)v[u].y << 32) | v[u].x;
        }
#pragma unroll
        for (int u = 0; u < NBUF / 3; ++u) acc += a[u];
    }
    if (acc == 0x123456789ULL) out[0] = acc;
}

static void *alloc_kind(size_t bytes, int kind);
template <int S, int NBUF>
static void run_phased_buf(int mib, uint32_t dt, uint64_t *out)
{
    const size_t bytes = (size_t)mib << 20;
    uint64_t *t = (uint64_t *)alloc_kind(bytes, 0);
    if (!t) return;
    const uint32_t n = (uint32_t)(bytes / 8);
    const uint32_t iters = 16 * 24 / NBUF;
    const int blocks = 256 * 64;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((gather_phased_buf<S, NBUF>), dim3(blocks), dim3(256), 0, 0, t, n, iters, dt, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double gathers = (double)blocks * 256 * iters * NBUF;
    printf("phased BUF (bounds-checked buffer loads) table %4d MiB  S=%2d  %2d lookups per lane  window %5.2f us : %7.1f G gathers/s\n", mib, S,
           NBUF, dt / 100.0, gathers / best / 1e6);
    fflush(stdout);
    (void)hipFree(t);
}

// "phased dense": what the phased scheme would deliver if the lookups of a window were COMPACTED -- the same 24 lookups
// per lane and round, but as NBUF/S full-width load instructions per window instead of NBUF predicated ones with 1/S of
// the lanes active each (is the texture-address path, one vector-memory instruction per ~25 cycles and CU, the limit?)
template <int S, int NBUF>
__global__ __launch_bounds__(256) void gather_phased_dense(const uint64_t *__restrict__ table, uint32_t n_elems, uint32_t iters,
                                                           uint32_t dt, uint64_t *out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per_slice = (n_elems + S - 1) / S;
    uint64_t acc = 0;
    uint64_t s = mix(tid + 1);
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t w0 = wall_clock64() / dt;
#pragma unroll 1
        for (uint32_t q = 0; q < (uint32_t)S; ++q) {
            const uint64_t w = w0 + q;
            while (wall_clock64() / dt < w) __builtin_amdgcn_s_sleep(4);
            const uint32_t p = (uint32_t)(w % S);
            uint64_t v[NBUF / S];
#pragma unroll
            for (int u = 0; u < NBUF / S; ++u) {
                s = s * 6364136223846793005ULL + 1442695040888963407ULL;
                uint32_t idx = p * per_slice + (uint32_t)(((s >> 32) * (uint64_t)per_slice) >> 32);
                idx = idx < n_elems ? idx : n_elems - 1;
                v[u] = table[idx];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NBUF / S; ++u) acc += v[u];
        }
    }
    if (acc == 0x123456789ULL) out[0] = acc;
}

// "phased sorted": the compaction done for real -- per round every wave counting-sorts its 24 x 64 lookups by slice into
// LDS (block number + owner slot in 32 bits), then each window is ceil(count / 64) full-width loads whose results are
// ANDed into the owners' words with ds_and_b64; the words come back to the owners' registers at the end of the round.
template <int S>
__global__ __launch_bounds__(256) void gather_phased_sorted(const uint64_t *__restrict__ table, uint32_t n_elems, uint32_t iters,
                                                            uint32_t dt, uint64_t *out)
{
    constexpr int NBUF = 24;
    __shared__ uint32_t s_ent[4][NBUF * 64];
    __shared__ uint64_t s_acc[4][8 * 64];
    __shared__ uint32_t s_cnt[4][S + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per_slice = (n_elems + S - 1) / S;
    uint64_t acc = 0;
    uint64_t s = mix(tid + 1);
    uint32_t *ent = s_ent[wave];
    uint64_t *wacc = s_acc[wave];
    uint32_t *cnt = s_cnt[wave];
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t idx[NBUF];
        uint32_t mycnt[S];
#pragma unroll
        for (int p = 0; p < S; ++p) mycnt[p] = 0;
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            s = s * 6364136223846793005ULL + 1442695040888963407ULL;
            idx[u] = (uint32_t)(((s >> 32) * (uint64_t)n_elems) >> 32);
            const uint32_t sl = idx[u] / per_slice;
#pragma unroll
            for (int p = 0; p < S; ++p) mycnt[p] += (sl == (uint32_t)p);
        }
        // exclusive prefix over the lanes, per slice (inclusive scan by shuffles), and the slice totals
        uint32_t base[S];
        uint32_t run = 0;
#pragma unroll
        for (int p = 0; p < S; ++p) {
            uint32_t v = mycnt[p];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane - d) & 63) << 2, (int)v);
                if (lane >= d) v += o;
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
            base[p] = run + v - mycnt[p];
            if (lane == 0) cnt[p] = run;
            run += total;
        }
        if (lane == 0) cnt[S] = run;
        for (int u = 0; u < 8; ++u) wacc[u * 64 + lane] = ~0ULL;
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            const uint32_t sl = idx[u] / per_slice;
            uint32_t pos = 0;
#pragma unroll
            for (int p = 0; p < S; ++p)
                if (sl == (uint32_t)p) { pos = base[p]; base[p] += 1; }
            ent[pos] = (idx[u] << 9) | (uint32_t)((u / 3) * 64 + lane);  // block number (<= 23 bits) | owner word
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint64_t w0 = wall_clock64() / dt;
#pragma unroll 1
        for (uint32_t q = 0; q < (uint32_t)S; ++q) {
            const uint64_t w = w0 + q;
            while (wall_clock64() / dt < w) __builtin_amdgcn_s_sleep(4);
            const uint32_t p = (uint32_t)(w % S);
            const uint32_t lo = cnt[p], hi = cnt[p + 1];
            const uint32_t n_loop = (hi - lo + 255) / 256;  // wave-uniform
            for (uint32_t tl = 0; tl < n_loop; ++tl) {
                const uint32_t i = lo + tl * 256 + lane;
                uint32_t e[4];
                uint64_t v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t j = i + 64 * k;
                    e[k] = j < hi ? ent[j] : 0xFFFFFFFFu;
                    v[k] = ~0ULL;
                    if (j < hi) v[k] = table[e[k] >> 9];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e[k] != 0xFFFFFFFFu) __hip_atomic_fetch_and(&wacc[e[k] & 511u], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int u = 0; u < 8; ++u) acc += wacc[u * 64 + lane];
    }
    if (acc == 0x123456789ULL) out[0] = acc;
}

static void *alloc_kind(size_t bytes, int kind);
static void one(const char *mode, int mib, int esz, int policy, int kind, uint64_t *out);

template <int S, int NBUF, int PF = 0>
static void run_phased(int mib, uint32_t dt, uint64_t *out, uint32_t slack = 0)
{
    const size_t bytes = (size_t)mib << 20;
    uint64_t *t = (uint64_t *)alloc_kind(bytes, 0);
    if (!t) return;
    const uint32_t n = (uint32_t)(bytes / 8);
    const uint32_t iters = 16 * 24 / NBUF;
    const int blocks = 256 * 64;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((gather_phased<S, NBUF, PF>), dim3(blocks), dim3(256), 0, 0, t, n, iters, dt, out, slack);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double gathers = (double)blocks * 256 * iters * NBUF;
    printf("phased table %4d MiB  S=%2d  %2d lookups buffered per lane  prefetch %d  window %5.2f us  slack %5.2f us : %7.1f G gathers/s\n", mib, S,
           NBUF, PF, dt / 100.0, slack / 100.0, gathers / best / 1e6);
    fflush(stdout);
    (void)hipFree(t);
}

template <int S, int NBUF>
static void run_phased_pump(int mib, uint32_t dt, uint64_t *out, int pumps_per_xcd, uint32_t lead = 1)
{
    const size_t bytes = (size_t)mib << 20;
    uint64_t *t = (uint64_t *)alloc_kind(bytes, 0);
    if (!t) return;
    uint32_t *done = nullptr;
    (void)hipMalloc(&done, 4);
    const uint32_t n = (uint32_t)(bytes / 8);
    const uint32_t iters = 16 * 24 / NBUF;
    const int blocks = 256 * 64;
    const uint32_t n_pump = 8u * pumps_per_xcd;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(done, 0, 4);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((gather_phased_pump<S, NBUF>), dim3(blocks + n_pump), dim3(256), 0, 0, t, n, iters, dt, out, n_pump, done, lead);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double gathers = (double)blocks * 256 * iters * NBUF;
    printf("phased PUMP table %4d MiB  S=%2d  %2d lookups per lane  %d pump workgroups per XCD (lead %u)  window %5.2f us : %7.1f G gathers/s\n",
           mib, S, NBUF, pumps_per_xcd, lead, dt / 100.0, gathers / best / 1e6);
    fflush(stdout);
    (void)hipFree(t);
    (void)hipFree(done);
}

template <int S, int NBUF>
static void run_phased_dense(int mib, uint32_t dt, uint64_t *out)
{
    const size_t bytes = (size_t)mib << 20;
    uint64_t *t = (uint64_t *)alloc_kind(bytes, 0);
    if (!t) return;
    const uint32_t n = (uint32_t)(bytes / 8);
    const uint32_t iters = 16 * 24 / NBUF;
    const int blocks = 256 * 64;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((gather_phased_dense<S, NBUF>), dim3(blocks), dim3(256), 0, 0, t, n, iters, dt, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double gathers = (double)blocks * 256 * iters * NBUF;
    printf("phased DENSE table %4d MiB  S=%2d  %2d lookups per lane and round (%d full-width loads per window)  window %5.2f us : %7.1f G gathers/s\n",
           mib, S, NBUF, NBUF / S, dt / 100.0, gathers / best / 1e6);
    fflush(stdout);
    (void)hipFree(t);
}

template <int S>
static void run_phased_sorted(int mib, uint32_t dt, uint64_t *out)
{
    const size_t bytes = (size_t)mib << 20;
    uint64_t *t = (uint64_t *)alloc_kind(bytes, 0);
    if (!t) return;
    const uint32_t n = (uint32_t)(bytes / 8);
    const uint32_t iters = 16;
    const int blocks = 256 * 64;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((gather_phased_sorted<S>), dim3(blocks), dim3(256), 0, 0, t, n, iters, dt, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double gathers = (double)blocks * 256 * iters * 24;
    printf("phased SORTED (LDS compaction + ds_and) table %4d MiB  S=%2d  window %5.2f us : %7.1f G gathers/s\n", mib, S, dt / 100.0,
           gathers / best / 1e6);
    fflush(stdout);
    (void)hipFree(t);
}

__global__ __launch_bounds__(1024) void gather_lds(const uint64_t *__restrict__ table, uint32_t n_elems, uint32_t iters,
                                                    uint64_t *out)
{
    extern __shared__ uint64_t s_tab[];  // 16384 words = 128 KiB
    const uint32_t slice = (blockIdx.x * 16384u) % (n_elems - 16384u);
    for (uint32_t i = threadIdx.x; i < 16384u; i += blockDim.x) s_tab[i] = table[slice + i];
    __syncthreads();
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    uint64_t s = mix(tid + 1);
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint64_t a = ~0ULL;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                s = s * 6364136223846793005ULL + 1442695040888963407ULL;
                a &= s_tab[(uint32_t)(s >> 50)];
            }
            acc += a;
        }
    }
    if (acc == 0x123456789ULL) out[0] = acc;
}

// routing traffic: read a 4-byte query, write an 8-byte answer, both coalesced
__global__ __launch_bounds__(256) void stream_qa(const uint32_t *__restrict__ q, uint64_t *__restrict__ a, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = (uint64_t)q[i] * 0x9E3779B97F4A7C15ULL;
}

static void *alloc_kind(size_t bytes, int kind)
{
    void *p = nullptr;
    hipError_t e;
    if (kind == 1) e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    else if (kind == 2) e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    else e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { printf("alloc kind %d failed: %s\n", kind, hipGetErrorString(e)); return nullptr; }
    (void)hipMemset(p, 0x5a, bytes);
    (void)hipDeviceSynchronize();
    return p;
}

static const char *kPolicy[] = {"plain", "nt", "sc1", "sc0sc1"};
static const char *kAlloc[] = {"hipMalloc", "uncached", "finegrained"};

template <int ESZ, int POLICY, bool XCD>
static double run_gather(const uint64_t *t, size_t bytes, uint64_t *out)
{
    const uint32_t n = (uint32_t)(bytes / ESZ);
    const uint32_t iters = 16;
    const int blocks = 256 * 64;  // 16 waves per CU x 16 rounds
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((gather<ESZ, POLICY, XCD>), dim3(blocks), dim3(256), 0, 0, t, n, iters, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double gathers = (double)blocks * 256 * iters * 24;
    return gathers / best / 1e6;  // G gathers/s
}

static double dispatch(const char *mode, const uint64_t *t, size_t bytes, int esz, int policy, uint64_t *out)
{
    const bool xcd = !strcmp(mode, "xcd");
#define CASE(E, P)                                                                                   \
    if (esz == E && policy == P) return xcd ? run_gather<E, P, true>(t, bytes, out) : run_gather<E, P, false>(t, bytes, out);
    CASE(8, 0) CASE(8, 1) CASE(8, 2) CASE(8, 3) CASE(16, 0) CASE(16, 1)
#undef CASE
    return -1.0;
}

static void one(const char *mode, int mib, int esz, int policy, int kind, uint64_t *out)
{
    const size_t bytes = (size_t)mib << 20;
    uint64_t *t = (uint64_t *)alloc_kind(bytes, kind);
    if (!t) return;
    const double g = dispatch(mode, t, bytes, esz, policy, out);
    printf("%-5s table %4d MiB  elem %2d B  %-6s %-11s : %7.1f G gathers/s  (%7.0f GB/s useful, %7.0f GB/s at 128 B/req)\n", mode, mib,
           esz, kPolicy[policy], kAlloc[kind], g, g * esz, g * 128);
    fflush(stdout);
    (void)hipFree(t);
}

int main(int argc, char **argv)
{
    uint64_t *out;
    (void)hipMalloc(&out, 8);
    if (argc >= 2 && !strcmp(argv[1], "phased")) {
        for (int mib : {20, 10}) {
            for (uint32_t dt : {100u, 200u, 300u, 400u, 600u, 1000u}) run_phased<8, 24>(mib, dt, out);
            for (uint32_t dt : {100u, 200u, 400u}) run_phased<16, 24>(mib, dt, out);
            for (uint32_t dt : {200u, 400u, 600u, 1000u}) run_phased<8, 48>(mib, dt, out);
            for (uint32_t dt : {100u, 200u, 400u}) run_phased<16, 48>(mib, dt, out);
        }
        for (uint32_t dt : {200u, 400u}) run_phased<4, 24>(10, dt, out);
        for (uint32_t dt : {400u, 800u}) run_phased<16, 48>(40, dt, out);
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "phased_buf")) {
        for (int mib : {10, 20}) {
            for (uint32_t dt : {200u, 300u, 450u}) run_phased<8, 24>(mib, dt, out);  // exec-masked loads, same box
            for (uint32_t dt : {100u, 150u, 200u, 250u, 300u, 450u, 600u}) run_phased_buf<8, 24>(mib, dt, out);
            for (uint32_t dt : {75u, 100u, 150u, 200u, 300u}) run_phased_buf<16, 24>(mib, dt, out);
            for (uint32_t dt : {150u, 200u, 300u, 450u}) run_phased_buf<8, 48>(mib, dt, out);
            for (uint32_t dt : {200u, 300u, 450u, 600u}) run_phased_buf<4, 24>(mib, dt, out);
        }
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "phased_slack")) {
        for (uint32_t dt : {150u, 200u, 300u, 450u})
            for (uint32_t sl : {0u, dt / 2, dt, 2 * dt}) run_phased<8, 24>(10, dt, out, sl);
        for (uint32_t dt : {100u, 150u, 200u, 300u})
            for (uint32_t sl : {0u, dt / 2, dt, 2 * dt}) run_phased<16, 24>(10, dt, out, sl);
        for (uint32_t dt : {200u, 300u, 450u, 600u})
            for (uint32_t sl : {0u, dt / 2, dt}) run_phased<8, 24>(20, dt, out, sl);
        for (uint32_t dt : {100u, 150u, 200u, 300u})
            for (uint32_t sl : {0u, dt / 2, dt, 2 * dt}) run_phased<16, 24>(20, dt, out, sl);
        for (uint32_t dt : {150u, 200u, 300u})
            for (uint32_t sl : {0u, dt}) run_phased<8, 48>(10, dt, out, sl);
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "phased_pump")) {
        for (int mib : {10, 20}) {
            for (uint32_t dt : {200u, 300u, 450u}) run_phased<8, 24>(mib, dt, out);  // the scheme as shipped, same box
            for (int pumps : {1, 2, 4})
                for (uint32_t dt : {100u, 150u, 200u, 300u, 450u}) run_phased_pump<8, 24>(mib, dt, out, pumps);
            for (int pumps : {2, 4})
                for (uint32_t dt : {75u, 100u, 150u, 200u, 300u}) run_phased_pump<16, 24>(mib, dt, out, pumps);
            for (uint32_t dt : {150u, 200u, 300u}) run_phased_pump<8, 48>(mib, dt, out, 2);
            for (uint32_t dt : {100u, 150u, 200u}) run_phased_pump<16, 48>(mib, dt, out, 2);
            for (uint32_t dt : {150u, 200u, 300u}) run_phased_pump<8, 24>(mib, dt, out, 2, 0);  // control: pumping the CURRENT slice
        }
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "phased_sorted")) {
        for (int mib : {10, 20})
            for (uint32_t dt : {200u, 300u, 450u, 600u, 800u}) run_phased_sorted<8>(mib, dt, out);
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "phased_dense")) {
        for (int mib : {10, 20}) {
            for (uint32_t dt : {50u, 100u, 150u, 200u, 300u, 450u, 600u}) run_phased_dense<8, 24>(mib, dt, out);
            for (uint32_t dt : {100u, 200u, 300u, 450u}) run_phased_dense<8, 48>(mib, dt, out);
        }
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "phased_pf")) {
        for (int mib : {20, 10}) {
            for (uint32_t dt : {150u, 200u, 300u, 400u, 600u}) run_phased<8, 24, 1>(mib, dt, out);
            for (uint32_t dt : {200u, 300u, 400u}) run_phased<8, 24, 2>(mib, dt, out);
            for (uint32_t dt : {200u, 300u, 400u, 600u}) run_phased<8, 48, 1>(mib, dt, out);
            for (uint32_t dt : {200u, 300u, 400u, 600u}) run_phased<8, 48, 2>(mib, dt, out);
            for (uint32_t dt : {100u, 150u, 200u, 300u}) run_phased<16, 24, 1>(mib, dt, out);
            for (uint32_t dt : {100u, 150u, 200u, 300u}) run_phased<16, 48, 1>(mib, dt, out);
        }
        for (uint32_t dt : {200u, 300u, 400u}) run_phased<8, 24, 1>(5, dt, out);
        for (uint32_t dt : {200u, 300u, 400u}) run_phased<8, 24, 0>(5, dt, out);
        one("full", 5, 8, 0, 0, out);
        for (uint32_t dt : {300u, 400u, 600u}) run_phased<16, 48, 2>(40, dt, out);
        for (uint32_t dt : {300u, 400u, 600u}) run_phased<16, 48, 1>(40, dt, out);
        one("full", 40, 8, 0, 0, out);
        return 0;
    }
    if (argc >= 7 && !strcmp(argv[1], "one")) {
        one(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), out);
        return 0;
    }
    // --- the wall: whole-table gathers by table size
    for (int mib : {2, 10, 20, 64, 400}) one("full", mib, 8, 0, 0, out);
    // --- request size / policy / allocation on the 20 MiB table
    for (int p = 1; p < 4; ++p) one("full", 20, 8, p, 0, out);
    for (int k = 1; k < 3; ++k)
        for (int p : {0, 3}) one("full", 20, 8, p, k, out);
    one("full", 20, 16, 0, 0, out);
    one("full", 20, 16, 1, 0, out);
    // --- slice per XCD: L2-resident gathers
    for (int mib : {10, 20, 32}) one("xcd", mib, 8, 0, 0, out);
    one("xcd", 20, 16, 0, 0, out);
    one("xcd", 20, 8, 2, 0, out);
    // --- table slice in LDS
    {
        const size_t bytes = (size_t)20 << 20;
        uint64_t *t = (uint64_t *)alloc_kind(bytes, 0);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gather_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        const uint32_t iters = 256;
        const int blocks = 256 * 4;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(gather_lds, dim3(blocks), dim3(1024), 128 * 1024, 0, t, (uint32_t)(bytes / 8), iters, out);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        printf("lds   128 KiB slice per workgroup, ds_read_b64 : %7.1f G gathers/s (incl. the slice load)\n",
               (double)blocks * 1024 * iters * 24 / ms / 1e6);
        (void)hipFree(t);
    }
    // --- routing traffic
    {
        const size_t n = (size_t)1 << 28;
        uint32_t *q = (uint32_t *)alloc_kind(n * 4, 0);
        uint64_t *a8 = (uint64_t *)alloc_kind(n * 8, 0);
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(stream_qa, dim3(256 * 16), dim3(256), 0, 0, q, a8, n);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        printf("stream 4 B in + 8 B out per item, 2^28 items : %7.1f G items/s (%7.0f GB/s)\n", n / ms / 1e6, n * 12.0 / ms / 1e6);
    }
    return 0;
}
