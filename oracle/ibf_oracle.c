/*
 * ibf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See ibf_oracle.h for scope and the "parity unpinned" statement.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the ReadBouncer tree).  Parts marked [SeqAn] restate the published algorithm
 * of the absent dependency (JensUweUlrich/seqan branch "SeqAn",
 * include/seqan/binning_directory/binning_directory_interleaved_bloom_filter.h
 * and bitvector_uncompressed.h; sdsl-lite v2.1.1 int_vector serialisation).
 *
 * Built with strict IEEE double arithmetic (no -ffast-math).
 */
#include "ibf_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ a.2 -- */
/* [SeqAn] TranslateTableCharToDna5_: A/a 0, C/c 1, G/g 2, T/t/U/u 3, rest N=4.
 * Used by (seqan::Dna5String) fragment, src/main/classify.hpp:272 and
 * src/main/adaptive_sampling.hpp:232. */
uint8_t orc_dna5_ord(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}

void orc_dna5_encode(const char *ascii, size_t len, uint8_t *ord)
{
    for (size_t i = 0; i < len; ++i) ord[i] = orc_dna5_ord((unsigned char)ascii[i]);
}

/* TSeqRevComp = ModifiedString<ModifiedString<Dna5String, ModComplementDna>, ModReverse>, src/IBF/IBF.hpp:96-97.
 * [SeqAn, RECALLED] ModComplementDna = ModView<FunctorComplement<Dna>>: the FOUR-letter functor.  Its argument type is
 * Dna, and the Dna5 -> Dna assignment is `value & 0x03` (alphabet_residue.h), so an N of the read (ordinal 4) becomes A
 * before it is complemented: the reverse strand holds T (3) there.  ("N stays N" would be ModComplementDna5, which the
 * reference does not name.)  One constant, switchable for the tests that run both candidates; the forward strand is not
 * touched -- it hashes N as ordinal 4. */
#define ORC_REVCOMP_OF_N 3
static int g_revcomp_of_n = ORC_REVCOMP_OF_N;

int orc_set_revcomp_of_n(int ordinal)
{
    if (ordinal != 3 && ordinal != 4) return -1;
    g_revcomp_of_n = ordinal;
    return 0;
}

int orc_get_revcomp_of_n(void) { return g_revcomp_of_n; }

void orc_revcomp(const uint8_t *ord, size_t len, uint8_t *out)
{
    for (size_t i = 0; i < len; ++i) {
        uint8_t o = ord[len - 1 - i];
        out[i] = (o < 4) ? (uint8_t)(3 - o) : (uint8_t)g_revcomp_of_n;
    }
}

/* ------------------------------------------------------------------ a.1 -- */
/* [SeqAn] BinningDirectory<InterleavedBloomFilter,...>::init() */
static void orc_ibf_init(orc_ibf *f)
{
    f->bin_width = (f->n_bins + ORC_INT_SIZE - 1) / ORC_INT_SIZE;
    f->block_bits = f->bin_width * ORC_INT_SIZE;
    f->n_blocks = f->block_bits ? f->n_bits / f->block_bits : 0;
    for (uint64_t i = 0; i < f->n_hash && i < ORC_MAX_HASH; ++i)
        f->precalc[i] = i ^ (f->kmer_size * ORC_SEED); /* u64 wrap */
    f->n_words = (f->n_bits + ORC_META_BITS + 63) / 64;
}

orc_ibf *orc_ibf_new(uint64_t n_bins, uint64_t n_hash, uint64_t kmer_size, uint64_t n_bits)
{
    if (n_hash > ORC_MAX_HASH || n_bins == 0) return NULL;
    orc_ibf *f = (orc_ibf *)calloc(1, sizeof(orc_ibf));
    if (!f) return NULL;
    f->n_bins = n_bins;
    f->n_hash = n_hash;
    f->kmer_size = kmer_size;
    f->n_bits = n_bits;
    orc_ibf_init(f);
    f->words = (uint64_t *)calloc(f->n_words ? f->n_words : 1, sizeof(uint64_t));
    f->owns_words = 1;
    if (!f->words) { free(f); return NULL; }
    return f;
}

orc_ibf *orc_ibf_wrap(uint64_t n_bins, uint64_t n_hash, uint64_t kmer_size, uint64_t n_bits,
                      uint64_t *words)
{
    if (n_hash > ORC_MAX_HASH || n_bins == 0) return NULL;
    orc_ibf *f = (orc_ibf *)calloc(1, sizeof(orc_ibf));
    if (!f) return NULL;
    f->n_bins = n_bins;
    f->n_hash = n_hash;
    f->kmer_size = kmer_size;
    f->n_bits = n_bits;
    orc_ibf_init(f);
    f->words = words;
    f->owns_words = 0;
    return f;
}

/* TEST HOOK, not part of the restatement: re-derive the per-hash multipliers from another seedValue, to make filter files
 * "written under different constants" for the first-contact check of the product (rb_dibf_compare / --verify-ibf). */
void orc_ibf_set_seed_for_tests(orc_ibf *f, uint64_t seed)
{
    for (uint64_t i = 0; i < f->n_hash && i < ORC_MAX_HASH; ++i) f->precalc[i] = i ^ (f->kmer_size * seed);
}

void orc_ibf_free(orc_ibf *f)
{
    if (!f) return;
    if (f->owns_words) free(f->words);
    free(f);
}

uint64_t *orc_ibf_words(orc_ibf *f) { return f->words; }
uint64_t orc_ibf_n_words(const orc_ibf *f) { return f->n_words; }

void orc_ibf_info(const orc_ibf *f, uint64_t *n_bins, uint64_t *n_hash, uint64_t *kmer_size,
                  uint64_t *n_bits, uint64_t *n_blocks, uint64_t *bin_width)
{
    if (n_bins) *n_bins = f->n_bins;
    if (n_hash) *n_hash = f->n_hash;
    if (kmer_size) *kmer_size = f->kmer_size;
    if (n_bits) *n_bits = f->n_bits;
    if (n_blocks) *n_blocks = f->n_blocks;
    if (bin_width) *bin_width = f->bin_width;
}

/* ------------------------------------------------------------------ a.3 -- */
/* [SeqAn] Shape<Dna5,SimpleShape> hash: base-5 big-endian polynomial (u64 wrap) */
uint64_t orc_kmer_value(const uint8_t *ord, uint64_t k)
{
    uint64_t v = 0;
    for (uint64_t i = 0; i < k; ++i) v = v * 5u + ord[i];
    return v;
}

/* [SeqAn] vecIndex = preCalcValues[i] * kmerHash; hashToIndex(): ^= >>27, %= noOfBlocks.
 * Returned value is the BLOCK number; the bit index is block * block_bits. */
uint64_t orc_block_index(const orc_ibf *f, uint64_t kmer_value, uint64_t hash_no)
{
    uint64_t x = f->precalc[hash_no] * kmer_value;
    x ^= x >> ORC_SHIFT;
    return x % f->n_blocks;
}

static inline void orc_set_bit(uint64_t *w, uint64_t bit) { w[bit >> 6] |= 1ULL << (bit & 63); }

/* [SeqAn] insertKmer(text, binNo): for every k-mer, every hash: set bit idx + binNo.
 * Called from src/IBF/IBFBuild.cpp:190. Texts shorter than k contribute nothing. */
void orc_ibf_insert(orc_ibf *f, const uint8_t *ord, size_t len, uint64_t bin)
{
    uint64_t k = f->kmer_size;
    if (len < k || k == 0 || f->n_blocks == 0) return;
    size_t possible = len - k + 1;
    for (size_t p = 0; p < possible; ++p) {
        uint64_t v = orc_kmer_value(ord + p, k);
        for (uint64_t i = 0; i < f->n_hash; ++i) {
            uint64_t idx = orc_block_index(f, v, i) * f->block_bits + bin;
            orc_set_bit(f->words, idx);
        }
    }
}

/* [SeqAn] count(counts, text): per k-mer AND the h words of each of bin_width columns,
 * ++counts[bin] per set bit.  Call sites: src/IBF/IBFClassify.cpp:97-98,149-150.
 * Note: the reference indexes counts[] (size noOfBins) with any set bit of the last
 * column; bits of bins >= noOfBins are never set by insertKmer, so they are ignored
 * here (the reference would write out of bounds on such a corrupt filter). */
void orc_ibf_count(const orc_ibf *f, const uint8_t *ord, size_t len, uint16_t *counts)
{
    memset(counts, 0, (size_t)f->n_bins * sizeof(uint16_t));
    uint64_t k = f->kmer_size;
    if (len < k || k == 0 || f->n_blocks == 0) return; /* getHash(): k > len -> no k-mers */
    size_t possible = len - k + 1;
    /* faithful to the reference's structure: a fresh hash vector per call */
    uint64_t *hashes = (uint64_t *)malloc(possible * sizeof(uint64_t));
    /* rolling evaluation, hashInit/hashNext; identical to direct evaluation mod 2^64 */
    uint64_t lead = 1;
    for (uint64_t i = 1; i < k; ++i) lead *= 5u;
    uint64_t v = orc_kmer_value(ord, k);
    hashes[0] = v;
    for (size_t p = 1; p < possible; ++p) {
        v = (v - ord[p - 1] * lead) * 5u + ord[p + k - 1];
        hashes[p] = v;
    }
    uint64_t idx[ORC_MAX_HASH];
    for (size_t p = 0; p < possible; ++p) {
        for (uint64_t i = 0; i < f->n_hash; ++i)
            idx[i] = orc_block_index(f, hashes[p], i) * f->bin_width; /* word index */
        for (uint64_t col = 0; col < f->bin_width; ++col) {
            uint64_t tmp = f->words[idx[0] + col];
            for (uint64_t i = 1; i < f->n_hash; ++i) tmp &= f->words[idx[i] + col];
            uint64_t base = col * 64;
            while (tmp) {
                uint64_t b = base + (uint64_t)__builtin_ctzll(tmp);
                tmp &= tmp - 1;
                if (b < f->n_bins) ++counts[b]; /* uint16_t wrap like the reference */
            }
        }
    }
    free(hashes);
}

/* ----------------------------------------------------------------- a.13 -- */
/* [SeqAn]/[sdsl] store(): metadata {noOfBins, noOfHashFunc, kmerSize, 0} as 64-bit ints at
 * bit n_bits, then sdsl int_vector<1>::serialize = u64 bit size + ceil(size/64) LE words.
 * Call sites: src/IBF/IBFBuild.cpp:505,307. */
int orc_ibf_store(const orc_ibf *f, const char *path)
{
    FILE *fp = fopen(path, "wb");
    if (!fp) return ORC_ERR_IO;
    uint64_t bit_size = f->n_bits + ORC_META_BITS;
    uint64_t n_words = (bit_size + 63) / 64;
    /* write metadata through a temporary tail so the in-memory filter stays untouched */
    uint64_t meta[4] = {f->n_bins, f->n_hash, f->kmer_size, 0};
    int ok = fwrite(&bit_size, 8, 1, fp) == 1;
    uint64_t mw = f->n_bits >> 6, ms = f->n_bits & 63;
    uint64_t tail_words = n_words - mw; /* 4 or 5 */
    uint64_t tail[6] = {0, 0, 0, 0, 0, 0};
    for (uint64_t i = 0; i < tail_words; ++i) tail[i] = f->words[mw + i];
    /* clear then set the 256 metadata bits starting at bit offset ms of tail[0] */
    for (int j = 0; j < 4; ++j) {
        if (ms == 0) {
            tail[j] = meta[j];
        } else {
            tail[j] = (tail[j] & ((1ULL << ms) - 1)) | (meta[j] << ms);
            tail[j + 1] = (tail[j + 1] & ~((1ULL << ms) - 1)) | (meta[j] >> (64 - ms));
        }
    }
    if (ok && mw) ok = fwrite(f->words, 8, mw, fp) == mw;
    if (ok) ok = fwrite(tail, 8, tail_words, fp) == tail_words;
    if (fclose(fp) != 0) ok = 0;
    return ok ? ORC_OK : ORC_ERR_IO;
}

/* [SeqAn] retrieve(): load bit_vector, read metadata from the last 256 bits, init().
 * Call sites: src/IBF/IBFBuild.cpp:343,360; src/config/configReader.cpp:216 (non-IBF
 * input must fail -> ORC_ERR_PARSE). */
orc_ibf *orc_ibf_load(const char *path, int *status)
{
    int st = ORC_OK;
    orc_ibf *f = NULL;
    FILE *fp = fopen(path, "rb");
    if (!fp) { if (status) *status = ORC_ERR_IO; return NULL; }
    uint64_t bit_size = 0;
    if (fread(&bit_size, 8, 1, fp) != 1 || bit_size < ORC_META_BITS) { st = ORC_ERR_PARSE; goto done; }
    {
        fseek(fp, 0, SEEK_END);
        long fsz = ftell(fp);
        uint64_t n_words = (bit_size + 63) / 64;
        if (fsz < 0 || (uint64_t)fsz != 8 + 8 * n_words) { st = ORC_ERR_PARSE; goto done; }
        fseek(fp, 8, SEEK_SET);
        uint64_t *words = (uint64_t *)malloc(n_words * 8);
        if (!words) { st = ORC_ERR_IO; goto done; }
        if (fread(words, 8, n_words, fp) != n_words) { free(words); st = ORC_ERR_PARSE; goto done; }
        uint64_t n_bits = bit_size - ORC_META_BITS;
        uint64_t mw = n_bits >> 6, ms = n_bits & 63, meta[4];
        for (int j = 0; j < 4; ++j) {
            meta[j] = words[mw + j] >> ms;
            if (ms) meta[j] |= words[mw + j + 1] << (64 - ms);
        }
        if (meta[0] == 0 || meta[1] == 0 || meta[1] > ORC_MAX_HASH || meta[2] == 0 || meta[2] > 255 ||
            meta[0] > n_bits) {
            free(words); st = ORC_ERR_PARSE; goto done;
        }
        f = orc_ibf_wrap(meta[0], meta[1], meta[2], n_bits, words);
        if (!f) { free(words); st = ORC_ERR_PARSE; goto done; }
        f->owns_words = 1;
        if (f->n_blocks == 0) { orc_ibf_free(f); f = NULL; st = ORC_ERR_PARSE; }
    }
done:
    fclose(fp);
    if (status) *status = st;
    return f;
}

/* ------------------------------------------------------------------ a.5 -- */
/* RationalApproximation, src/IBF/IBF.hpp:268-277 (A&S 26.2.23) */
static double orc_rational_approximation(double t)
{
    const double c[] = {2.515517, 0.802853, 0.010328};
    const double d[] = {1.432788, 0.189269, 0.001308};
    return t - ((c[2] * t + c[1]) * t + c[0]) / (((d[2] * t + d[1]) * t + d[0]) * t + 1.0);
}

/* NormalCDFInverse, src/IBF/IBF.hpp:284-308; *ok=0 where the reference throws invalid_argument */
double orc_normal_cdf_inverse(double p, int *ok)
{
    if (p <= 0.0 || p >= 1.0) { if (ok) *ok = 0; return 0.0; }
    if (ok) *ok = 1;
    if (p < 0.5) return -orc_rational_approximation(sqrt(-2.0 * log(p)));
    return orc_rational_approximation(sqrt(-2.0 * log(1.0 - p)));
}

/* (uint16_t)double as the reference's x86-64 build performs it: cvttsd2si to int32
 * (NaN / out of range -> 0x80000000), then the low 16 bits.  Negative values wrap. */
static uint16_t orc_to_u16(double x)
{
    int32_t i;
    if (isnan(x) || x >= 2147483648.0 || x <= -2147483649.0) i = (int32_t)0x80000000u;
    else i = (int32_t)x;
    return (uint16_t)(uint32_t)i;
}

/* calculateCI, src/IBF/IBF.hpp:320-338 */
void orc_calculate_ci(double r, uint8_t kmer_size, uint32_t readlen, double confidence,
                      uint16_t *lo, uint16_t *hi)
{
    double q = 1.0 - pow(1.0 - r, kmer_size);
    double L = ((double)readlen - (double)kmer_size + 1.0);
    double varN = L * (1.0 - q) * (q * (2.0 * (double)kmer_size + (2.0 / r) - 1.0) - 2.0 * (double)kmer_size)
                  + (double)kmer_size * ((double)kmer_size - 1.0) * pow((1.0 - q), 2.0)
                  + (2.0 * (1.0 - q) / (pow(r, 2.0))) * ((1.0 + ((double)kmer_size - 1.0) * (1.0 - q)) * r - q);
    double alpha = 1 - confidence;
    int ok = 1;
    double z = orc_normal_cdf_inverse(1.0 - alpha / 2.0, &ok);
    uint16_t low = orc_to_u16(floor(L * q - z * sqrt(varN)));
    uint16_t high = orc_to_u16(ceil(L * q + z * sqrt(varN)));
    if (lo) *lo = low;
    if (hi) *hi = high;
}

/* threshold, src/IBF/IBFClassify.cpp:154-162: uint16_t readlen; int16_t threshold =
 * readlen - k + 1 - ci.second; passed to max_matches(uint16_t). */
uint16_t orc_threshold(uint64_t readlen, uint64_t kmer_size, double r, double confidence)
{
    uint16_t lo, hi;
    orc_calculate_ci(r, (uint8_t)kmer_size, (uint32_t)readlen, confidence, &lo, &hi);
    uint16_t readlen16 = (uint16_t)readlen;
    int64_t t = (int64_t)readlen16 - (int64_t)kmer_size + 1 - (int64_t)hi;
    int16_t threshold = (int16_t)(uint16_t)(uint64_t)t;
    return (uint16_t)threshold;
}

/* ------------------------------------------------------------------ a.7 -- */
/* Read::max_matches, src/IBF/IBFClassify.cpp:48-71 */
uint64_t orc_max_matches(const uint16_t *fwd, const uint16_t *rev, uint64_t n_bins, uint16_t threshold)
{
    uint64_t max_kmer_count = 0;
    for (uint64_t b = 0; b < n_bins; ++b) {
        if (fwd[b] >= threshold || rev[b] >= threshold) {
            if (fwd[b] > max_kmer_count) max_kmer_count = fwd[b];
            if (rev[b] > max_kmer_count) max_kmer_count = rev[b];
        }
    }
    return max_kmer_count;
}

/* Read::select_matches, src/IBF/IBFClassify.cpp:16-38 */
int orc_select_matches(const uint16_t *fwd, const uint16_t *rev, uint64_t n_bins, uint16_t threshold)
{
    for (uint64_t b = 0; b < n_bins; ++b)
        if (fwd[b] >= threshold || rev[b] >= threshold) return 1;
    return 0;
}

static void orc_count_both(const orc_ibf *f, const uint8_t *ord, size_t len, uint16_t *fwd, uint16_t *rev)
{
    uint8_t *rc = (uint8_t *)malloc(len ? len : 1);
    orc_revcomp(ord, len, rc);
    orc_ibf_count(f, ord, len, fwd); /* IBFClassify.cpp:149 */
    orc_ibf_count(f, rc, len, rev);  /* IBFClassify.cpp:150 */
    free(rc);
}

uint16_t orc_raw_max(const orc_ibf *f, const uint8_t *ord, size_t len)
{
    uint16_t *fwd = (uint16_t *)malloc(2 * (size_t)f->n_bins * sizeof(uint16_t));
    uint16_t *rev = fwd + f->n_bins;
    orc_count_both(f, ord, len, fwd, rev);
    uint16_t m = 0;
    for (uint64_t b = 0; b < f->n_bins; ++b) {
        if (fwd[b] > m) m = fwd[b];
        if (rev[b] > m) m = rev[b];
    }
    free(fwd);
    return m;
}

/* Read::count_matches, src/IBF/IBFClassify.cpp:138-171 */
uint64_t orc_count_matches(const orc_ibf *f, const uint8_t *ord, size_t len, double r, double conf)
{
    uint16_t *fwd = (uint16_t *)malloc(2 * (size_t)f->n_bins * sizeof(uint16_t));
    uint16_t *rev = fwd + f->n_bins;
    orc_count_both(f, ord, len, fwd, rev);
    uint16_t threshold = orc_threshold(len, f->kmer_size, r, conf);
    uint64_t m = orc_max_matches(fwd, rev, f->n_bins, threshold);
    free(fwd);
    return m;
}

/* ------------------------------------------------------------------ a.8 -- */
/* Read::classify(vector<TIbf>&) + find_matches, src/IBF/IBFClassify.cpp:181-226, 81-128 */
int orc_classify_any(orc_ibf *const *filters, size_t n, const uint8_t *ord, size_t len,
                     double r, double conf, int *found)
{
    *found = 0;
    if (n == 0) return ORC_ERR_NULL_FILTER;
    if (len < filters[0]->kmer_size) return ORC_ERR_SHORT_READ;
    for (size_t i = 0; i < n; ++i) {
        const orc_ibf *f = filters[i];
        uint16_t *fwd = (uint16_t *)malloc(2 * (size_t)f->n_bins * sizeof(uint16_t));
        uint16_t *rev = fwd + f->n_bins;
        orc_count_both(f, ord, len, fwd, rev);
        uint16_t threshold = orc_threshold(len, f->kmer_size, r, conf);
        int hit = orc_select_matches(fwd, rev, f->n_bins, threshold);
        free(fwd);
        if (hit) { *found = 1; break; }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ a.9 -- */
/* Read::classify(vector<IBFMeta>&), src/IBF/IBFClassify.cpp:239-297.
 * One std::async per filter in the reference; results are order-independent. */
int orc_classify_best(orc_ibf *const *filters, size_t n, const uint8_t *ord, size_t len,
                      double r, double conf, int *best)
{
    *best = -1;
    if (n == 0) return ORC_ERR_NULL_FILTER;
    if (len < filters[0]->kmer_size) return ORC_ERR_SHORT_READ;
    uint64_t max_kmer_count = 0;
    int best_index = -1;
    for (size_t i = 0; i < n; ++i) {
        uint64_t count = orc_count_matches(filters[i], ord, len, r, conf);
        if (count > max_kmer_count) { best_index = (int)i; max_kmer_count = count; }
    }
    *best = best_index;
    return ORC_OK;
}

static uint64_t orc_group_max(orc_ibf *const *fl, size_t n, const uint8_t *ord, size_t len,
                              double r, double conf)
{
    uint64_t max_kmer_count = 0;
    for (size_t i = 0; i < n; ++i) {
        if (len >= fl[i]->kmer_size) { /* IBFClassify.cpp:318,340: shorter reads skip the filter */
            uint64_t count = orc_count_matches(fl[i], ord, len, r, conf);
            if (count > max_kmer_count) max_kmer_count = count;
        }
    }
    return max_kmer_count;
}

/* ----------------------------------------------------------------- a.10 -- */
/* Read::classify(filt1, filt2), src/IBF/IBFClassify.cpp:299-365 */
int orc_classify_pair(orc_ibf *const *f1, size_t n1, orc_ibf *const *f2, size_t n2,
                      const uint8_t *ord, size_t len, double r, double conf,
                      uint64_t *first, uint64_t *second)
{
    *first = 0;
    *second = 0;
    if (n1 == 0 || n2 == 0) return ORC_ERR_NULL_FILTER;
    *first = orc_group_max(f1, n1, ord, len, r, conf);
    *second = orc_group_max(f2, n2, ord, len, r, conf);
    return ORC_OK;
}

/* ----------------------------------------------------------------- a.11 -- */
/* check_unblock, src/main/adaptive_sampling.hpp:35-113 */
int orc_check_unblock(orc_ibf *const *deplete, size_t nd, orc_ibf *const *target, size_t nt,
                      const uint8_t *ord, size_t len, double r, double conf, uint8_t *decision)
{
    int withTarget = nt != 0, withDepletion = nd != 0;
    *decision = 0;
    if (withDepletion && withTarget) {
        uint64_t d, t;
        orc_classify_pair(deplete, nd, target, nt, ord, len, r, conf, &d, &t);
        if (d > 0) {
            if (t > 0) {
                double r2 = r - 0.02; /* conf.error_rate -= 0.02, :55 */
                orc_classify_pair(deplete, nd, target, nt, ord, len, r2, conf, &d, &t);
                *decision = (d > 0 && t == 0) ? 1 : 0;
            } else {
                *decision = 1;
            }
        } else {
            *decision = (t > 0) ? 2 : 0;
        }
        return ORC_OK;
    } else if (withDepletion) {
        int best, st = orc_classify_best(deplete, nd, ord, len, r, conf, &best);
        if (st != ORC_OK) return st;
        *decision = (best > -1) ? 1 : 0;
        return ORC_OK;
    } else {
        int best, st = orc_classify_best(target, nt, ord, len, r, conf, &best);
        if (st != ORC_OK) return st; /* NullFilterException when both sets are empty */
        *decision = (best < 0) ? 1 : 2;
        return ORC_OK;
    }
}

/* ----------------------------------------------------------------- a.12 -- */
/* classify_deplete_target, src/main/classify.hpp:58-111.  NOTE the argument order:
 * p = r.classify(TargetFilters, DepletionFilters, Conf)  =>  first = target, second = deplete. */
static int orc_classify_deplete_target(orc_ibf *const *deplete, size_t nd, orc_ibf *const *target, size_t nt,
                                       const uint8_t *ord, size_t len, double r, double conf,
                                       int *classified, int *best_target)
{
    uint64_t t, d;
    int st, best = -1;
    *classified = 0;
    orc_classify_pair(target, nt, deplete, nd, ord, len, r, conf, &t, &d);
    if (t > 0) {
        if (d > 0) {
            double r2 = r - 0.02;
            orc_classify_pair(target, nt, deplete, nd, ord, len, r2, conf, &t, &d);
            if (t > 0 && d > 0) return ORC_OK;
            if (t > 0) {
                st = orc_classify_best(target, nt, ord, len, r, conf, &best); /* at the restored r */
                if (st != ORC_OK) return st;
                if (best != -1) { *classified = 1; *best_target = best; }
                return ORC_OK;
            }
            return ORC_OK;
        }
        st = orc_classify_best(target, nt, ord, len, r, conf, &best);
        if (st != ORC_OK) return st;
        if (best != -1) { *classified = 1; *best_target = best; }
    }
    return ORC_OK;
}

/* chunk loop of classify_reads, src/main/classify.hpp:247-301 (one read) */
int orc_classify_read_chunks(orc_ibf *const *deplete, size_t nd, orc_ibf *const *target, size_t nt,
                             const char *ascii, size_t len, uint32_t chunk_length, uint32_t max_chunks,
                             double r, double conf, int *too_short, int *classified, int *best_target,
                             uint32_t *chunks_used)
{
    *too_short = 0;
    *classified = 0;
    *best_target = -1;
    *chunks_used = 0;
    if (len < chunk_length) { *too_short = 1; return ORC_OK; } /* :247-250 */
    int status = ORC_OK;
    uint8_t *ord = (uint8_t *)malloc(chunk_length ? chunk_length : 1);
    uint8_t i = 0; /* uint8_t like the reference (:260) */
    while (i < max_chunks) {
        uint64_t fragend = (uint64_t)(i + 1) * chunk_length;
        uint64_t fragstart = (uint64_t)i * chunk_length;
        if (fragend > len) fragend = len;
        if (fragstart > fragend) { status = ORC_ERR_BAD_CHUNK; break; } /* infix(begin>end): undefined */
        size_t flen = (size_t)(fragend - fragstart);
        orc_dna5_encode(ascii + fragstart, flen, ord);
        ++*chunks_used;
        int cls = 0, st = ORC_OK;
        if (nd && nt) {
            st = orc_classify_deplete_target(deplete, nd, target, nt, ord, flen, r, conf, &cls, best_target);
        } else if (nd) {
            int best;
            st = orc_classify_best(deplete, nd, ord, flen, r, conf, &best);
            cls = best > -1;
        } else {
            int best;
            st = orc_classify_best(target, nt, ord, flen, r, conf, &best);
            if (st == ORC_OK && best != -1) { cls = 1; *best_target = best; }
        }
        if (st != ORC_OK) { status = st; break; } /* exception -> failed++ (:306-316) */
        if (cls) { *classified = 1; break; }
        i++;
    }
    free(ord);
    return status;
}

/* ------------------------------------------------------------ build side -- */
/* calculate_filter_size_bits, src/IBF/IBFBuild.cpp:404-413 */
uint64_t orc_calculate_filter_size_bits(uint64_t fragment_length, uint64_t kmer_size,
                                        uint64_t hash_functions, double max_fp, uint64_t n_bins)
{
    uint64_t max_kmer_count = fragment_length - kmer_size + 1;
    uint64_t optimalNumberOfBins = (uint64_t)(floor(((double)n_bins / 64.0) + 1) * 64);
    uint64_t BinSizeBits = (uint64_t)ceil(-1 / (pow(1 - pow((double)max_fp, 1.0 / (double)hash_functions),
                                                    1.0 / ((double)(hash_functions * max_kmer_count))) - 1));
    return BinSizeBits * optimalNumberOfBins;
}

/* cutOutNNNs + concatenation, src/IBF/IBFBuild.cpp:112-132 and :81-88.
 * Quirk kept: the piece that runs to the end of the sequence loses its last base (:121-125). */
size_t orc_cut_out_nnns(const char *seq, size_t len, char *out)
{
    size_t n = 0, end = 0;
    for (;;) {
        size_t start = end;
        while (start < len && seq[start] == 'N') ++start; /* find_first_not_of("N", end) */
        if (start >= len) break;
        end = start;
        while (end < len && seq[end] != 'N') ++end; /* find("N", start) */
        if (end >= len) { /* npos > seqlen */
            size_t cnt = len - start - 1;
            memcpy(out + n, seq + start, cnt);
            n += cnt;
            break;
        }
        memcpy(out + n, seq + start, end - start);
        n += end - start;
    }
    return n;
}

/* fragment loop of add_sequences_to_filter, src/IBF/IBFBuild.cpp:165-204 */
uint64_t orc_add_sequence(orc_ibf *f, const uint8_t *ord, size_t len, uint64_t fragment_length,
                          uint64_t kmer_size, uint64_t overlap_length, uint64_t first_bin)
{
    uint64_t binid = first_bin;
    int64_t fragIdx = 0;
    int64_t fragstart = fragIdx * (int64_t)fragment_length - (int64_t)overlap_length + 1;
    if (fragstart < 0) fragstart = 0;
    int64_t seqlen = (int64_t)len;
    while (fragstart < (seqlen - 1)) {
        uint64_t fragend = (uint64_t)(fragIdx + 1) * fragment_length;
        if (fragend > len) fragend = len;
        orc_ibf_insert(f, ord + fragstart, (size_t)(fragend - (uint64_t)fragstart), binid++);
        fragIdx++;
        fragstart = fragIdx * (int64_t)fragment_length - (int64_t)kmer_size + 1;
    }
    return binid;
}

/* [SeqAn] resizeBins(bins) as used by IBF::update_filter (src/IBF/IBFBuild.cpp:274): the number of blocks is
 * kept (hash positions stay valid), every block is widened to ceil(bins/64) words, old words stay at the start
 * of their block, new columns are empty, noOfBits = noOfBlocks * newBlockBitSize.  Returns a new filter. */
orc_ibf *orc_ibf_resize_bins(const orc_ibf *f, uint64_t new_bins)
{
    if (new_bins < f->n_bins) return NULL;
    uint64_t new_width = (new_bins + 63) / 64;
    orc_ibf *g = orc_ibf_new(new_bins, f->n_hash, f->kmer_size, f->n_blocks * new_width * 64);
    if (!g) return NULL;
    for (uint64_t b = 0; b < f->n_blocks; ++b)
        memcpy(g->words + b * new_width, f->words + b * f->bin_width, (size_t)f->bin_width * 8);
    return g;
}

/* ------------------------------------------------------- synthetic filler -- */
static inline uint64_t orc_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* each bit ~ Bernoulli(55/256 = 0.2148), the design load 0.01^(1/3) = 0.2154 of
 * calculate_filter_size_bits; binary digits of 55/256 = .00110111 folded LSB first */
uint64_t orc_synth_word(uint64_t seed, uint64_t word_index)
{
    uint64_t r[8];
    for (int i = 0; i < 8; ++i)
        r[i] = orc_mix64(seed + (word_index * 8 + (uint64_t)i + 1) * 0x9E3779B97F4A7C15ULL);
    uint64_t acc = r[7];  /* digit 8 = 1 */
    acc |= r[6];          /* digit 7 = 1 */
    acc |= r[5];          /* digit 6 = 1 */
    acc &= r[4];          /* digit 5 = 0 */
    acc |= r[3];          /* digit 4 = 1 */
    acc |= r[2];          /* digit 3 = 1 */
    acc &= r[1];          /* digit 2 = 0 */
    acc &= r[0];          /* digit 1 = 0 */
    return acc;
}

void orc_ibf_fill_synth(orc_ibf *f, uint64_t seed)
{
    uint64_t used = f->n_blocks * f->bin_width;
    uint64_t rem = f->n_bins & 63;
    uint64_t last_mask = rem ? ((1ULL << rem) - 1) : ~0ULL;
    for (uint64_t w = 0; w < f->n_words; ++w) {
        if (w >= used) { f->words[w] = 0; continue; }
        uint64_t x = orc_synth_word(seed, w);
        if ((w % f->bin_width) == f->bin_width - 1) x &= last_mask;
        f->words[w] = x;
    }
}

/* ------------------------------------------------- batch helpers (timing) -- */
typedef struct {
    const orc_ibf *f;
    orc_ibf *const *deplete; size_t nd;
    orc_ibf *const *target; size_t nt;
    const char *ascii; const uint64_t *offsets; const uint32_t *lens;
    size_t begin, end;
    double r, conf;
    uint16_t *out_max; uint8_t *decision; uint8_t *status;
    int mode;
} orc_job;

static void *orc_worker(void *arg)
{
    orc_job *j = (orc_job *)arg;
    size_t cap = 1024;
    uint8_t *ord = (uint8_t *)malloc(cap);
    for (size_t i = j->begin; i < j->end; ++i) { /* one read at a time, like the reference */
        size_t len = j->lens[i];
        if (len > cap) { cap = len * 2; ord = (uint8_t *)realloc(ord, cap); }
        orc_dna5_encode(j->ascii + j->offsets[i], len, ord);
        if (j->mode == 0) {
            j->out_max[i] = orc_raw_max(j->f, ord, len);
        } else {
            uint8_t dec = 0;
            int st = orc_check_unblock(j->deplete, j->nd, j->target, j->nt, ord, len, j->r, j->conf, &dec);
            j->decision[i] = dec;
            j->status[i] = (uint8_t)st;
        }
    }
    free(ord);
    return NULL;
}

static void orc_run_jobs(orc_job *proto, size_t n_reads, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_reads && n_reads > 0) n_threads = (int)n_reads;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    orc_job *jobs = (orc_job *)malloc(sizeof(orc_job) * (size_t)n_threads);
    size_t per = (n_reads + (size_t)n_threads - 1) / (size_t)n_threads;
    for (int t = 0; t < n_threads; ++t) {
        jobs[t] = *proto;
        jobs[t].begin = (size_t)t * per < n_reads ? (size_t)t * per : n_reads;
        jobs[t].end = jobs[t].begin + per < n_reads ? jobs[t].begin + per : n_reads;
        if (n_threads == 1) orc_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, orc_worker, &jobs[t]);
    }
    if (n_threads > 1)
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
}

void orc_batch_raw_max(const orc_ibf *f, const char *ascii, const uint64_t *offsets,
                       const uint32_t *lens, size_t n_reads, int n_threads, uint16_t *out_max)
{
    orc_job p;
    memset(&p, 0, sizeof(p));
    p.f = f; p.ascii = ascii; p.offsets = offsets; p.lens = lens; p.out_max = out_max; p.mode = 0;
    orc_run_jobs(&p, n_reads, n_threads);
}

void orc_batch_check_unblock(orc_ibf *const *deplete, size_t nd, orc_ibf *const *target, size_t nt,
                             const char *ascii, const uint64_t *offsets, const uint32_t *lens,
                             size_t n_reads, double r, double conf, int n_threads,
                             uint8_t *decision, uint8_t *status)
{
    orc_job p;
    memset(&p, 0, sizeof(p));
    p.deplete = deplete; p.nd = nd; p.target = target; p.nt = nt;
    p.ascii = ascii; p.offsets = offsets; p.lens = lens; p.r = r; p.conf = conf;
    p.decision = decision; p.status = status; p.mode = 1;
    orc_run_jobs(&p, n_reads, n_threads);
}
