/*
 * readbouncer_amd_tuning.h -- measurement aids and scheduling knobs of libreadbouncer_amd.so.
 *
 * Nothing in this header has a counterpart in the reference and nothing in it changes a result: these calls pick kernel
 * forms, window lengths and batch cuts for A/B measurements, report what the engine planned, time its kernels, probe what the
 * device delivers, fill filters with synthetic bits and replay arrival processes.  bench.py, profiles/ and the tests use them;
 * a ReadBouncer integration needs only include/readbouncer_amd.h (the reference-mapped calls of INTEGRATION.md section 1).
 * Same shared library, same symbols.
 */
#ifndef READBOUNCER_AMD_TUNING_H_
#define READBOUNCER_AMD_TUNING_H_

#include "readbouncer_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic data ------------------------------------------------------------------------ */
/* synthetic filler for benchmarks: every bin bit ~ Bernoulli(55/256), padding bits clear */
RB_API int rb_dibf_fill_synth(rb_dibf *f, uint64_t seed);

/* ---- pool: statistics and scheduling ---------------------------------------------------------- */
/* Per worker, since creation or the last reset: the device it drives, seconds spent inside its engine's rb_classify_batch, reads
 * and parts of calls served.  busy / wall time = the share of the time that device's engine had work (bench.py --pool). */
RB_API int rb_pool_get_stats(rb_pool *p, size_t n, int *devices, double *busy_seconds, uint64_t *reads, uint64_t *calls, int reset);
RB_API int rb_pool_set_min_split(rb_pool *p, size_t reads_per_device);
/* Calls from several host threads run concurrently: each worker (engine + host thread per device) has a FIFO of its own,
 * an unsplit micro-batch goes to the least loaded worker, and callers only meet while a call's parts are queued -- K calling
 * threads keep K engines busy, like the reference's N classification threads behind one queue
 * (src/main/adaptive_sampling.hpp:745-751).  Per calling thread the calls stay ordered (each returns before the next starts).
 * rb_pool_set_serialize(p, 1) is a diagnostic: one call at a time, whoever makes it. */
RB_API int rb_pool_set_serialize(rb_pool *p, int enabled);
/* Kernel timing of every engine of the pool (rb_engine_set_timing / rb_engine_kernel_time per worker): total_ms[i] = summed K1 time
 * of worker i since the last collection, n_launches[i] = bracketed launches (a host batch crosses PCIe in slices, one launch
 * each).  For bench.py's pool legs: a roofline figure per device from kernel time, beside the PCIe-inclusive rate. */
RB_API int rb_pool_set_timing(rb_pool *p, int enabled);
RB_API int rb_pool_kernel_time(rb_pool *p, size_t n, double *total_ms, uint64_t *n_launches);

/* ---- arrival replay (config 5) ------------------------------------------------------------------ */
/* Replay of an arrival process through the engine (measurement aid for the live scenario; rb_live.cpp): chunk i = read_len
 * bytes at seqs + i*read_len, available arrival_s[i] seconds after the start (ascending); a work-conserving dispatcher
 * takes everything that has arrived (<= max_batch chunks, 0 = 16384) per rb_classify_batch call (check_unblock).
 * out_latency_s[i] = decision - arrival; the first call_cap calls report their size and service time; out_calls = number of
 * calls made.  The reference's counterpart is its classification thread popping one read at a time
 * (src/main/adaptive_sampling.hpp:214-356). */
RB_API int rb_replay_arrivals(rb_engine *e, const char *seqs, uint32_t read_len, size_t n, const double *arrival_s,
                              size_t max_batch, double error_rate, double significance, uint8_t *out_decision,
                              double *out_latency_s, uint32_t *out_call_reads, double *out_call_service_s, size_t call_cap,
                              size_t *out_calls, double *out_elapsed_s);
/* (max_batch is clamped to min(max_batch, n, 2^20); arrival_s that is not ascending is RB_ERR_INVALID_ARG; the dispatcher
 * spins on the steady clock between arrivals -- it owns a core for the length of the replay, like the reference's
 * classification thread polling its queue, adaptive_sampling.hpp:226-228.)
 * The same dispatcher in front of the live step: chunk i belongs to read read_ids[i] and goes through rb_live_process, so
 * an undecided read's next chunk is classified as the concatenation with what once_seen holds (up to the cut-off, i.e.
 * reads of up to ~1.9 kbp) -- adaptive_sampling.hpp:276-338.  out_action as rb_live_process. */
RB_API int rb_live_replay_arrivals(rb_live *lv, const uint32_t *read_ids, const char *seqs, uint32_t read_len, size_t n,
                                   const double *arrival_s, size_t max_batch, uint8_t *out_action, double *out_latency_s,
                                   uint32_t *out_classified_len, uint32_t *out_call_reads, double *out_call_service_s,
                                   size_t call_cap, size_t *out_calls, double *out_elapsed_s);

/* ---- where a large table lies in HBM ---------------------------------------------------------------
 * Filters of 1 GiB and more are placed by trial: the same table gathers 1.7-2.9 % slower or faster from one allocation of a process to
 * the next (where the driver finds the pages -- an 8 GiB power-of-two table always gets the fast kind, a reference-sized 4.7 GB one does
 * or does not), so rb_dibf_create / _upload / _open / _clone_to / _resize_bins allocate up to `tries` candidates (default 5), probe each
 * with random whole-block gathers (~0.1 s), stop early only when one is 3 % faster than the slowest seen, keep the best and free the others.
 * Transient cost: up to (tries - 1) x the table of HBM at load time (never more than half of what is free) and 1-2 s.  tries = 0 or 1: off.
 * Never done on a device where an engine of this process is alive (see rb_dibf_placement_cost).  Process-wide; results never depend on it.  rb_dibf_placement: how many allocations were probed for this filter (0: not placed by
 * trial), what the kept one and the slowest one delivered in GB/s. */
RB_API int rb_set_placement_tries(int tries);
RB_API int rb_dibf_placement(const rb_dibf *f, uint32_t *tries, double *kept_gbps, double *worst_gbps);
/* What the trial cost for this filter: seconds spent allocating and probing the candidates, seconds waited afterwards until the kept
 * table probed as in the trial (bounded by 3 s; the driver clears the freed candidates in the background), the most HBM the candidates
 * held at once, and -- for a table of 1 GiB or more that was NOT placed by trial -- why: 1 = an engine was alive on the device (a
 * process that is already classifying there is not stalled by probe launches and transient copies: the table takes the first
 * allocation), 2 = less than twice the table was free.  bench.py keeps these per filter in bench_detail.json.  Any pointer may be NULL. */
RB_API int rb_dibf_placement_cost(const rb_dibf *f, double *trial_seconds, double *settle_seconds, uint64_t *peak_bytes, uint32_t *skipped);

/* ---- engine: kernel forms, planner, timing, probe ------------------------------------------------ */
/* Filters of one hash geometry in one table.  Every filter the reference builds with one fragment_size has noOfBits =
 * BinSizeBits x 64 x binWidth (src/IBF/IBFBuild.cpp:404-413), i.e. the same noOfBlocks whatever its bin count; with equal k and
 * three hash functions a k-mer then hashes to the same block number in all of them.  For such filters (blocks of at most 8
 * words, at most 16 words together) the engine keeps a merged copy in which their blocks sit side by side, and one gather per
 * (k-mer, hash function) serves all of them -- the narrow filters are bound by requests, not bytes.  mode 1 (default): when it
 * pays -- the members one after the other are estimated to take longer than one pass over the merged table (a merged table of
 * two to four words is served by the clock-phased kernel like a filter of that width, wider ones by plain gathers): the
 * reference's README shape (a two-word deplete filter and three one-word targets), any two filters too large for the
 * clock-phased kernels, small ones whose merged copy still fits an L2, two or three one-word filters of up to 30 MiB --;
 * 2: whenever two or more filters qualify; 0: never.  Large batches only (micro-batches keep the latency kernels); the copy
 * follows changes of its members (rb_dibf_insert ...).  Results are identical. */
RB_API int rb_engine_set_merge(rb_engine *e, int mode);
/* What the engine has merged (or will, at its next large batch): the number of merged tables, the filters they serve and the HBM
 * bytes of the copies, which live beside the members and are SHARED by every engine of the process that merges the same filters in
 * the same order on the same device (N threads with an engine each -- adaptive_sampling.hpp:745-751 -- gather from one copy; it
 * is freed with its last engine).  A copy larger than 16 GiB (RB_MERGE_MAX_BYTES) is not made,
 * and a group whose copy the device has no room for dissolves at its first call: its members are then served one by one.
 * Any out pointer may be NULL. */
RB_API int rb_engine_merge_info(rb_engine *e, uint32_t *n_tables, uint32_t *n_filters, uint64_t *copy_bytes);

/* Micro-batch latency: batches of at most max_reads reads (x column slices) run the latency form of the
 * count kernel (one workgroup per read, its waves share the read's k-mers and strands); larger batches
 * run the throughput form (one wave per read).  Results are identical.  0 disables; default 2048. */
RB_API int rb_engine_set_split_threshold(rb_engine *e, uint32_t max_reads);

/* The count kernels of different filters run concurrently (filter 0 on the call's stream, the others on the engine's
 * auxiliary streams, joined by events before the decision kernel) -- the reference starts one std::async per filter
 * (src/IBF/IBFClassify.cpp:256-260).  0 serialises them on one stream.  Default on. */
RB_API int rb_engine_set_overlap(rb_engine *e, int enabled);

/* Latency kernel on wide filters (blocks of 17+ word columns): up to max_parts workgroups share one read, each wave
 * walking 1/max_shares of a 64-k-mer tile (default 8 and 4); 0 or 1 parts = one workgroup per read.  The partial
 * counters are added by the last workgroup to finish.  Results are identical. */
RB_API int rb_engine_set_split_parts(rb_engine *e, uint32_t max_parts, uint32_t max_shares);

/* Engines with one filter, micro-batches of up to 512 reads in the latency form: the count kernel makes the decisions too -- the
 * workgroup that writes a read's raw maximum runs check_unblock's decision for it (src/main/adaptive_sampling.hpp:35-113) -- instead
 * of a decision kernel launched behind it: one dependent launch less per call (1-2 us).  0 keeps the two launches.  Default on.
 * Results are identical. */
RB_API int rb_engine_set_fold_decide(rb_engine *e, int enabled);

/* Opt-in.  rb_classify_batch with up to 2 048 reads: the kernel that writes the call's last result also stores a sequence number into a
 * word of page-locked host memory (behind a system-scope release), and the calling thread spins on that word instead of waiting for the
 * stream -- the stream's own wait returns about 4 us later than the results are there (median of a call: one read 40.6 -> 36.1 us).  The
 * price is the tail: the runtime retires its commands in the stream's wait, without it in bulk every few hundred calls (p99 of config 5's
 * replay + 0-20 us from box to box), which is why this is off by default.  A word that has not arrived after 20 ms falls back to the
 * stream, which also reports a failed kernel, and every 256th call waits for the stream as well.  0 = off (default), 1 = on, a value
 * above 1 = on with that period instead of 256 (measurements).  Results are identical. */
RB_API int rb_engine_set_completion_word(rb_engine *e, int enabled);

/* Filters larger than table_bytes are gathered with non-temporal loads (default 512 MiB = 2x the Infinity
 * Cache; measured +2.4 % on the 8 GiB filter, -1.9 % on a 0.41 GB one).  Results are identical. */
RB_API int rb_engine_set_nt_threshold(rb_engine *e, uint64_t table_bytes);

/* Narrow filters -- blocks of one to eight words, tables of a few L2 sizes (10-20 MB: a bacterial genome at the reference's
 * default fragment_size) -- are bound by cache and fabric REQUESTS, not bytes: each 8-byte gather that misses the XCD's 4 MiB
 * L2 costs a 128-byte request.  Two measures, both leave the results untouched:
 *  - filters of at most `table_bytes` (default 128 MiB) never run beside another filter of the same call, so each has the
 *    L2 to itself (rb_engine_set_serial_table_bytes; 0 = overlap everything as rb_engine_set_overlap says);
 *  - for tables of one- to four-word blocks of [min_table_bytes, max_table_bytes] (default 1.25-128 MiB; three and four words:
 *    4.5-48 MiB) and batches of at least min_reads (2049: everything above the latency kernel's micro-batches; more for tables
 *    beyond 32 MiB) the throughput kernel gathers in clock-phased slices: the
 *    table is cut into slices of 0.5 to 4 MiB (at most 32) and the 100 MHz wall clock tells every wave which slice to gather
 *    from, in windows of base_ticks + ticks_per_mib * table MiB ticks of 10 ns -- both 0 = the built-in rule, a whole cycle
 *    over the table of 33-60 us by kernel shape -- so an XCD's L2 holds one slice at a time
 *    (rb_engine_set_phased; max_table_bytes = 0 switches it off; all five arguments 0 also takes one-word filters back to
 *    the plain kernel, whose 512-k-mer tiles are half empty on 250 bp reads).  Blocks of five and more words gain nothing
 *    from phases and keep the plain kernel. */
RB_API int rb_engine_set_serial_table_bytes(rb_engine *e, uint64_t table_bytes);
RB_API int rb_engine_set_phased(rb_engine *e, uint64_t min_table_bytes, uint64_t max_table_bytes, uint32_t base_ticks,
                                uint32_t ticks_per_mib, uint32_t min_reads);
/* What the engine would launch for filter `filter_index` (deplete filters first) on a batch of n_reads reads of at most max_len
 * bases, with its current settings: kernel form, geometry, and -- for the phased form -- the row of the planner's table
 * (readbouncer_amd/csrc/rb_phase_plan.h) with the slice size and window length it gives.  For bench.py's roofline line (which
 * kernel was timed), profiles/phase_rule_check.py (rule against measured best) and the tests; no reference counterpart. */
typedef struct rb_plan_info {
    char kernel[48];             /* ibf_count_max_kernel | _phased_kernel | _merged_kernel | _split_kernel */
    uint64_t table_bytes;        /* of the table the lookups go to (the merged copy when merged_members > 0) */
    uint32_t block_words, stride_words;
    uint32_t merged_members;     /* > 0: the filter is served from a merged table of that many filters */
    uint32_t lanes_per_block_log2, words_per_lane, column_slices, counter_planes, nontemporal;
    uint32_t split_waves;        /* latency form: waves per workgroup (0: throughput form) */
    uint32_t phased;             /* 1: clock-phased gathers */
    uint32_t phase_shape;        /* rbplan::PhaseShape */
    char phase_shape_name[64];
    uint32_t phase_slice_log2, phase_slices, phase_window_ticks;  /* window length in effect, in 10 ns ticks */
    uint32_t phase_rule_ticks;   /* what the planner's table alone gives (differs after rb_engine_calibrate) */
    uint32_t reserved0;
    uint64_t phase_slice_bytes;  /* slice length in effect: 2^phase_slice_log2, or -- four-word one-lane builds -- the equal-length slices
                                  * the table is cut into instead (fewer and up to 1.19 x longer; rb_phase_plan.h, phase_equal_slices) */
} rb_plan_info;
RB_API int rb_engine_plan(rb_engine *e, size_t filter_index, size_t n_reads, uint32_t max_len, rb_plan_info *out);
/* Fits the window lengths of the clock-phased gathers to THIS device: the planner's table was measured on one box, and clocks,
 * firmware and compilers move the optima.  For every table the engine would serve with the phased form on batches of n_reads reads
 * of read_len bases, K1 is timed on synthetic reads with the table's window and with 0.7 / 0.85 / 1.2 / 1.45 x that (same slice
 * size); the best point of the curve smoothed along the window length replaces the rule for that table and kernel shape when it
 * wins by 4 % and a second measurement confirms it, until the engine goes away or rb_engine_set_phased /
 * rb_engine_set_phase_slices is called.  Calibrate at the batch size the engine will be given: where the dips and cliffs of the
 * two-word and wide shapes lie moves with it.  Stops trying new windows after max_ms (0 = no limit); a few
 * launches per table, tens of milliseconds in all.  Results never depend on it; call it on an idle engine.  n_tables: phased
 * tables found; n_changed: how many got a new window.  No reference counterpart (profiles/phase_rule_check.py is the
 * offline form of the same sweep, with an exit code). */
RB_API int rb_engine_calibrate(rb_engine *e, size_t n_reads, uint32_t read_len, double max_ms, uint32_t *n_tables, uint32_t *n_changed);

/* How the phased form cuts a table, for tests and experiments: slices of 2^slice_log2 bytes (0 = the built-in rule, 0.5 to 4 MiB
 * by table size and block width; 1-5 = as small as max_slices allows, which puts test-sized tables through many slices), and
 * never more than max_slices (1-32, default 32; a table that would need more gets larger slices).  Results are identical. */
RB_API int rb_engine_set_phase_slices(rb_engine *e, uint32_t slice_log2, uint32_t max_slices);
/* ... or into n_slices slices of EQUAL length, any number of blocks each (1-32; 0 = back to the built-in rule, which itself cuts the
 * four-word tables, the two-word tables of the LDS-offset builds and one-word tables from 50 MiB on this way: slices shorter than an
 * L2 leave the lines of the window before in place while the next slice arrives).  Ignored while rb_engine_set_phase_slices names a
 * slice size.  Results are identical. */
RB_API int rb_engine_set_phase_equal_slices(rb_engine *e, uint32_t n_slices);

/* Two-word tables (65-128 bins, or two to three small targets merged) of up to 2^21 - 1 blocks (32 MiB), reads of up to 256 k-mers, phased
 * form: `reads` = 2 or 3 takes the build that carries that many reads per wave through one pass of the windows -- AND accumulators in
 * registers, block numbers packed into LDS -- so that a pass, which reloads the table once per XCD whatever rides along, serves more
 * reads (40 instead of 28 per CU at two reads per wave); 1 = one read per wave with the offsets in LDS (eight waves per SIMD);
 * 0 = the one-read build that keeps the offsets in registers.  On a merged table these builds gather from a complemented twin of the
 * copy (the bounds check's zero is then neutral and the masking goes away); + 16 keeps the AND form there too (measurements).
 * Results are identical. */
RB_API int rb_engine_set_reads_per_wave(rb_engine *e, uint32_t reads);

/* OPT-IN, off by default; never part of a roofline figure (work is skipped).  check_unblock (src/main/adaptive_sampling.hpp:35-113) looks at a
 * filter's count only through "count >= threshold(r)" and "count >= threshold(r - 0.02)" (the rescan of :55-56); both are settled the moment
 * some bin of the filter reaches the larger of the two thresholds on either strand -- the reference counts on, and counts again for the
 * rescan.  With this mode on, RB_MODE_CHECK_UNBLOCK calls of the throughput form (batches above the micro-batch limit) that do NOT ask for
 * the raw maxima (out_maxcount / d_maxcount NULL) let a wave of the plain count kernel (filters of five and more word columns: the
 * depletion filter) stop there; target filters do so only when best_target is not asked for either (their counts pick it).  Every
 * output the call returns is identical to the mode being off.  In host depletion most reads are positive and stop after a fraction of
 * one strand. */
RB_API int rb_engine_set_early_decision(rb_engine *e, int enabled);

/* How the eight XCDs walk the slices of a phased table (each has an L2 of its own, so each reloads every slice): bit 0 of `mode` -- at
 * any time every XCD works on a different slice (slice = (window + XCD number) mod slices); bit 1 -- the XCDs' windows start an eighth
 * of a window apart, so that they refill their L2s one after the other instead of all in the same instant.  0 (default): one clock, one
 * slice for the whole chip.  Results are identical. */
RB_API int rb_engine_set_phase_xcd_skew(rb_engine *e, uint32_t mode);

/* Host batches above 8 MB of read bytes cross PCIe in slices of about slice_bytes (default 32 MiB): slice i+1 is
 * copied on a copy stream while slice i is counted.  0 = one slice (no overlap).  Results are identical. */
RB_API int rb_engine_set_host_slice_bytes(rb_engine *e, uint64_t slice_bytes);

/* Kernel timing for the roofline figure: when enabled every rb_classify_batch* call brackets its
 * count kernels (K1, all filters) with a hipEvent pair recorded on the launch stream, without
 * synchronising.  rb_engine_kernel_time waits for the recorded pairs, returns their summed elapsed
 * time and the number of calls, and resets the accumulation. */
RB_API int rb_engine_set_timing(rb_engine *e, int enabled);
RB_API int rb_engine_kernel_time(rb_engine *e, double *total_ms, uint64_t *n_calls);

/* Measurement aid, NOT part of the classify path (no reference counterpart): what this device delivers for the access pattern
 * of the wide count kernels -- random gathers of whole rows of row_bytes (128, 1024 or 4096) from the first table_bytes (0 = all)
 * of the filter's own table in HBM, 16 bytes per lane, loads_in_flight (12 or 24) wave instructions issued back to back,
 * nontemporal as rb_engine_set_nt_threshold would choose, no compute attached; runs of about target_ms (0 = 200), best of three.
 * bench.py calls it on the filter it has just timed so that `roofline` carries a same-box, same-run reference point
 * (`read_peak_probe`) beside the 8 TB/s spec figure. */
RB_API int rb_dibf_probe_read_peak(rb_dibf *f, uint64_t table_bytes, uint32_t row_bytes, int nontemporal, uint32_t loads_in_flight,
                                   double target_ms, double *gbps_out, double *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* READBOUNCER_AMD_TUNING_H_ */
