/* The drop-in boundary from plain C99: what a cgo / JNI / ctypes binding of include/readbouncer_amd.h does, without a binding.
 *
 * Builds a depletion and a target filter in HBM (the reference's IBF::create_filter loop, src/IBF/IBFBuild.cpp:165-204),
 * hands both to an engine (the DepletionFilters / TargetFilters of src/main/adaptive_sampling.hpp:214) and asks for the
 * check_unblock decision (adaptive_sampling.hpp:35-113) of a batch of read prefixes: reads cut from the depletion
 * reference must come back 1 (unblock), reads from the target reference 2 (stop_receiving), random reads 0 (wait).
 *
 *   gcc -std=c99 -Iinclude examples/adaptive_sampling_c_abi.c -Lreadbouncer_amd -lreadbouncer_amd \
 *       -Wl,-rpath,$PWD/readbouncer_amd -o adaptive_sampling_c_abi
 *
 * Exit code 0: every read got the decision above.  2: no GPU (the engine has no CPU path).  1: anything else. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "readbouncer_amd.h"

#define CHECK(call)                                                                                          \
    do {                                                                                                     \
        int st_ = (call);                                                                                    \
        if (st_ != RB_OK) {                                                                                  \
            fprintf(stderr, "%s -> %s (%s)\n", #call, rb_status_string(st_), rb_last_error());               \
            return st_ == RB_ERR_NO_DEVICE ? 2 : 1;                                                          \
        }                                                                                                    \
    } while (0)

static uint64_t lcg_state = 0x243F6A8885A308D3ull;
static uint32_t lcg(void)
{
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(lcg_state >> 33);
}

static void random_dna(char *dst, size_t n)
{
    size_t i;
    for (i = 0; i < n; ++i) dst[i] = "ACGT"[lcg() & 3];
}

enum { REF_LEN = 200000, READ_LEN = 360, N_READS = 96, FRAGMENT = 10000, KMER = 13 };

int main(void)
{
    char *dep_ref, *tgt_ref, *reads;
    uint64_t offsets[N_READS], next_bin = 0;
    uint32_t lens[N_READS];
    uint8_t decision[N_READS], status[N_READS], expect[N_READS];
    uint16_t maxcount[N_READS * 2];
    int32_t best_target[N_READS];
    rb_dibf *dep = NULL, *tgt = NULL;
    rb_engine *eng = NULL;
    rb_ibf_info info;
    uint64_t n_bins, n_bits;
    int i, wrong = 0;

    printf("%s, %d device(s)\n", rb_version(), rb_device_count());
    if (rb_device_count() <= 0) {
        fprintf(stderr, "no gfx950 GPU visible: the engine has no CPU fallback\n");
        return 2;
    }

    dep_ref = (char *)malloc(REF_LEN);
    tgt_ref = (char *)malloc(REF_LEN);
    reads = (char *)malloc((size_t)N_READS * READ_LEN);
    if (!dep_ref || !tgt_ref || !reads) return 1;
    random_dna(dep_ref, REF_LEN);
    random_dna(tgt_ref, REF_LEN);

    /* one fragment per bin; filter size the way the reference picks it (three hash functions, 1 % false positives) */
    n_bins = rb_fragment_bounds(REF_LEN, FRAGMENT, KMER, 500, NULL, NULL, 0);
    n_bits = rb_calculate_filter_size_bits(FRAGMENT, KMER, 3, 0.01, n_bins);
    CHECK(rb_dibf_create(0, n_bins, 3, KMER, n_bits, &dep));
    CHECK(rb_dibf_create(0, n_bins, 3, KMER, n_bits, &tgt));
    CHECK(rb_dibf_add_sequence(dep, dep_ref, REF_LEN, FRAGMENT, 500, 0, &next_bin));
    CHECK(rb_dibf_add_sequence(tgt, tgt_ref, REF_LEN, FRAGMENT, 500, 0, &next_bin));
    CHECK(rb_dibf_get_info(dep, &info));
    printf("filters: %llu bins, k = %llu, %llu bits each\n", (unsigned long long)info.n_bins,
           (unsigned long long)info.kmer_size, (unsigned long long)info.n_bits);

    for (i = 0; i < N_READS; ++i) {
        char *r = reads + (size_t)i * READ_LEN;
        int j;
        offsets[i] = (uint64_t)i * READ_LEN;
        lens[i] = READ_LEN;
        expect[i] = (uint8_t)(i % 3 == 0 ? 1 : i % 3 == 1 ? 2 : 0);
        if (expect[i] == 0) {
            random_dna(r, READ_LEN);
            continue;
        }
        memcpy(r, (expect[i] == 1 ? dep_ref : tgt_ref) + lcg() % (REF_LEN - READ_LEN), READ_LEN);
        for (j = 0; j < READ_LEN; ++j) /* 4 % substitutions, well inside the 10 % the threshold is built for */
            if (lcg() % 25 == 0) r[j] = "ACGT"[lcg() & 3];
    }

    CHECK(rb_engine_create(0, &dep, 1, &tgt, 1, &eng));
    CHECK(rb_classify_batch(eng, reads, offsets, lens, N_READS, 0.1, 0.95, RB_MODE_CHECK_UNBLOCK, maxcount, best_target,
                            decision, status));
    for (i = 0; i < N_READS; ++i) {
        if (status[i] != RB_OK || decision[i] != expect[i]) {
            ++wrong;
            printf("read %d: decision %u (expected %u), status %s, counts deplete %u target %u\n", i, decision[i], expect[i],
                   rb_status_string(status[i]), maxcount[2 * i], maxcount[2 * i + 1]);
        }
    }
    printf("threshold at %d bp: %u k-mers; read 0 (depletion): %u / %u, read 1 (target): %u / %u, read 2 (random): %u / %u\n",
           READ_LEN, rb_threshold(READ_LEN, KMER, 0.1, 0.95), maxcount[0], maxcount[1], maxcount[2], maxcount[3], maxcount[4],
           maxcount[5]);
    printf("%d reads, %d unexpected decisions\n", N_READS, wrong);

    rb_engine_destroy(eng);
    rb_dibf_free(dep);
    rb_dibf_free(tgt);
    free(dep_ref);
    free(tgt_ref);
    free(reads);
    return wrong ? 1 : 0;
}
