/*
 * ibf_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of ReadBouncer's IBF classify hot path, used only as
 * the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  Nothing under readbouncer_amd/ or include/ may include, link or call
 * this file.
 *
 * PARITY STATUS: "parity unpinned" at the bit level for the hash function and
 * the .ibf layout.  The arithmetic of seqan::count / insertKmer / store /
 * retrieve lives in an un-vendored dependency that is absent from
 * /root/reference:
 *     https://github.com/JensUweUlrich/seqan.git  GIT_TAG "SeqAn" (a branch)
 *     simongog/sdsl-lite v2.1.1
 *     (reference: src/seqan/CMakeLists.txt.in:20-30)
 * and the reference's committed .ibf fixtures are missing
 * (.MISSING_LARGE_BLOBS).  The published algorithm of
 * seqan/binning_directory/binning_directory_interleaved_bloom_filter.h is
 * restated here (see ORC_SEED etc.).  Everything that IS in the reference
 * (threshold model, max/argmax logic, decisions, chunk driver, build
 * parameters) follows the cited file:line and is pinned by the reference's
 * own known-answer tests (tests/test_oracle_kat.py).
 */
#ifndef IBF_ORACLE_H_
#define IBF_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants of the absent SeqAn InterleavedBloomFilter (restated) ---- */
#define ORC_SEED 0x90b45d39fb6da1faULL /* seedValue                           */
#define ORC_SHIFT 27                   /* shiftValue                          */
#define ORC_INT_SIZE 64                /* intSize                             */
#define ORC_META_BITS 256              /* filterMetadataSize                  */
#define ORC_MAX_HASH 16

/* status codes mirroring the reference's exceptions (IBFExceptions.hpp) */
enum {
    ORC_OK = 0,
    ORC_ERR_NULL_FILTER = 1, /* NullFilterException  IBFExceptions.hpp:178 */
    ORC_ERR_SHORT_READ = 2,  /* ShortReadException   IBFExceptions.hpp:96  */
    ORC_ERR_COUNT_KMER = 3,  /* CountKmerException   IBFExceptions.hpp:123 */
    ORC_ERR_IO = 4,          /* MissingIBFFileException / StoreFilter      */
    ORC_ERR_PARSE = 5,       /* ParseIBFFileException IBFExceptions.hpp:344 */
    ORC_ERR_BAD_CHUNK = 6    /* chunk start beyond read end (undefined in
                                the reference, classify.hpp:264-273)       */
};

typedef struct orc_ibf {
    uint64_t n_bins;     /* noOfBins      */
    uint64_t n_hash;     /* noOfHashFunc  */
    uint64_t kmer_size;  /* kmerSize      */
    uint64_t n_bits;     /* noOfBits (without the 256 metadata bits) */
    uint64_t bin_width;  /* ceil(n_bins / 64) */
    uint64_t block_bits; /* 64 * bin_width */
    uint64_t n_blocks;   /* n_bits / block_bits */
    uint64_t precalc[ORC_MAX_HASH];
    uint64_t n_words; /* (n_bits + 256 + 63) / 64 */
    uint64_t *words;  /* sdsl::bit_vector payload, LSB first */
    int owns_words;
} orc_ibf;

/* a.2  ASCII -> Dna5 ordinal (A0 C1 G2 T3/U3, everything else 4) */
uint8_t orc_dna5_ord(unsigned char c);
void orc_dna5_encode(const char *ascii, size_t len, uint8_t *ord);
/* TSeqRevComp (IBF.hpp:96-97): reverse + ModComplementDna.  An N of the read becomes ordinal orc_get_revcomp_of_n() on
 * the reverse strand: 3 (T) by default -- the four-letter functor sees N as A -- or 4 ("N stays N", the ModComplementDna5
 * reading); process-wide switch for the tests that run both. */
void orc_revcomp(const uint8_t *ord, size_t len, uint8_t *out);
int orc_set_revcomp_of_n(int ordinal); /* 3 or 4; -1 otherwise */
int orc_get_revcomp_of_n(void);

/* a.1  TIbf(bins, h, k, bits)  (IBFBuild.cpp:465) */
orc_ibf *orc_ibf_new(uint64_t n_bins, uint64_t n_hash, uint64_t kmer_size, uint64_t n_bits);
/* non-owning view over an existing word array of (n_bits+256+63)/64 words */
orc_ibf *orc_ibf_wrap(uint64_t n_bins, uint64_t n_hash, uint64_t kmer_size, uint64_t n_bits,
                      uint64_t *words);
void orc_ibf_free(orc_ibf *f);
/* test hook: hash with another seedValue (files "written under different constants") */
void orc_ibf_set_seed_for_tests(orc_ibf *f, uint64_t seed);
uint64_t *orc_ibf_words(orc_ibf *f);
uint64_t orc_ibf_n_words(const orc_ibf *f);
void orc_ibf_info(const orc_ibf *f, uint64_t *n_bins, uint64_t *n_hash, uint64_t *kmer_size,
                  uint64_t *n_bits, uint64_t *n_blocks, uint64_t *bin_width);

/* k-mer value + h-fold block index (a.3) -- exposed for unit tests */
uint64_t orc_kmer_value(const uint8_t *ord, uint64_t k);
uint64_t orc_block_index(const orc_ibf *f, uint64_t kmer_value, uint64_t hash_no);

/* seqan::insertKmer(filter, fragment, bin)  (IBFBuild.cpp:190) */
void orc_ibf_insert(orc_ibf *f, const uint8_t *ord, size_t len, uint64_t bin);
/* seqan::count(filter, seq)  (IBFClassify.cpp:97,149); counts has n_bins entries */
void orc_ibf_count(const orc_ibf *f, const uint8_t *ord, size_t len, uint16_t *counts);

/* seqan::store / seqan::retrieve  (IBFBuild.cpp:505,343) */
int orc_ibf_store(const orc_ibf *f, const char *path);
orc_ibf *orc_ibf_load(const char *path, int *status);

/* a.5  calculateCI (IBF.hpp:320-338); returns low in *lo, high in *hi */
void orc_calculate_ci(double r, uint8_t kmer_size, uint32_t readlen, double confidence,
                      uint16_t *lo, uint16_t *hi);
double orc_normal_cdf_inverse(double p, int *ok);
/* a.6  threshold as the uint16_t that max_matches receives (IBFClassify.cpp:156-162) */
uint16_t orc_threshold(uint64_t readlen, uint64_t kmer_size, double r, double confidence);

/* a.7  max_matches / select_matches (IBFClassify.cpp:48-71, 16-38) */
uint64_t orc_max_matches(const uint16_t *fwd, const uint16_t *rev, uint64_t n_bins, uint16_t threshold);
int orc_select_matches(const uint16_t *fwd, const uint16_t *rev, uint64_t n_bins, uint16_t threshold);
/* raw max over bins and strands, no threshold (what the GPU kernel K1 emits) */
uint16_t orc_raw_max(const orc_ibf *f, const uint8_t *ord, size_t len);
/* Read::count_matches (IBFClassify.cpp:138-171) */
uint64_t orc_count_matches(const orc_ibf *f, const uint8_t *ord, size_t len, double r, double conf);

/* a.8  Read::classify(vector<TIbf>&)  (IBFClassify.cpp:181-226) -> 0/1 */
int orc_classify_any(orc_ibf *const *filters, size_t n, const uint8_t *ord, size_t len,
                     double r, double conf, int *found);
/* a.9  Read::classify(vector<IBFMeta>&) (IBFClassify.cpp:239-297) -> best index or -1 */
int orc_classify_best(orc_ibf *const *filters, size_t n, const uint8_t *ord, size_t len,
                      double r, double conf, int *best);
/* a.10 Read::classify(filt1, filt2) (IBFClassify.cpp:299-365) */
int orc_classify_pair(orc_ibf *const *f1, size_t n1, orc_ibf *const *f2, size_t n2,
                      const uint8_t *ord, size_t len, double r, double conf,
                      uint64_t *first, uint64_t *second);

/* a.11 check_unblock (adaptive_sampling.hpp:35-113): 0 wait, 1 unblock, 2 stop_receiving */
int orc_check_unblock(orc_ibf *const *deplete, size_t nd, orc_ibf *const *target, size_t nt,
                      const uint8_t *ord, size_t len, double r, double conf, uint8_t *decision);

/* a.12 classify_deplete_target + chunk loop (classify.hpp:58-111, 247-301) for ONE read.
 * classified: 0/1; best_target: index credited (-1 none); chunks_used: chunks evaluated.
 * returns ORC_OK, or an error status when the reference would have thrown (failed++). */
int orc_classify_read_chunks(orc_ibf *const *deplete, size_t nd, orc_ibf *const *target, size_t nt,
                             const char *ascii, size_t len, uint32_t chunk_length, uint32_t max_chunks,
                             double r, double conf, int *too_short, int *classified, int *best_target,
                             uint32_t *chunks_used);

/* build side (oracle tooling, IBFBuild.cpp) */
uint64_t orc_calculate_filter_size_bits(uint64_t fragment_length, uint64_t kmer_size,
                                        uint64_t hash_functions, double max_fp, uint64_t n_bins);
/* cutOutNNNs + concatenation (IBFBuild.cpp:81-88,112-132); out must hold len bytes; returns new len */
size_t orc_cut_out_nnns(const char *seq, size_t len, char *out);
/* fragment loop (IBFBuild.cpp:165-204): inserts one sequence, returns next bin id */
uint64_t orc_add_sequence(orc_ibf *f, const uint8_t *ord, size_t len, uint64_t fragment_length,
                          uint64_t kmer_size, uint64_t overlap_length, uint64_t first_bin);

/* resizeBins of IBF::update_filter (IBFBuild.cpp:274); new filter, caller frees */
orc_ibf *orc_ibf_resize_bins(const orc_ibf *f, uint64_t new_bins);

/* deterministic synthetic filler shared with the GPU fill kernel's definition:
 * bit j of word w (w < n_blocks*bin_width) is set with probability 55/256, bins >= n_bins clear */
uint64_t orc_synth_word(uint64_t seed, uint64_t word_index);
void orc_ibf_fill_synth(orc_ibf *f, uint64_t seed);

/* multi-threaded batch helper for the cpu_baseline leg: raw max per read over one filter,
 * reads given as concatenated ASCII + offsets; n_threads pthreads, read-parallel
 * (the reference's scaling model: N classify threads, adaptive_sampling.hpp:745-751) */
void orc_batch_raw_max(const orc_ibf *f, const char *ascii, const uint64_t *offsets,
                       const uint32_t *lens, size_t n_reads, int n_threads, uint16_t *out_max);
/* full check_unblock over a batch, read-parallel */
void orc_batch_check_unblock(orc_ibf *const *deplete, size_t nd, orc_ibf *const *target, size_t nt,
                             const char *ascii, const uint64_t *offsets, const uint32_t *lens,
                             size_t n_reads, double r, double conf, int n_threads,
                             uint8_t *decision, uint8_t *status);

#ifdef __cplusplus
}
#endif
#endif
