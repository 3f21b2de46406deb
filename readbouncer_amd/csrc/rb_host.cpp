// rb_host.cpp -- host-side parts of the C ABI: status strings, .ibf file image, error model,
// build-side helpers.  These are host functions in the reference as well (load_filter,
// calculateCI, calculate_filter_size_bits, cutOutNNNs); all counting/decision work is in the
// HIP files.  Written from the behaviour of the cited reference lines, not from their text.
#include "rb_internal.h"
#include "rb_io.h"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace rb {

static thread_local std::string g_last_error;
static thread_local std::string g_last_warning;
void set_warning(const std::string &msg) { g_last_warning = msg; }
const std::string &last_warning() { return g_last_warning; }

void set_error(const std::string &msg) { g_last_error = msg; }
int fail(int status, const std::string &msg)
{
    g_last_error = msg;
    return status;
}

bool geometry_from(uint64_t n_bins, uint64_t n_hash, uint64_t kmer_size, uint64_t n_bits, rb_ibf_info *g)
{
    if (n_bins == 0 || n_hash == 0 || kmer_size == 0) return false;
    g->n_bins = n_bins;
    g->n_hash = n_hash;
    g->kmer_size = kmer_size;
    g->n_bits = n_bits;
    g->bin_width = (n_bins + rbspec::kIntSize - 1) / rbspec::kIntSize;
    uint64_t block_bits = g->bin_width * rbspec::kIntSize;
    g->n_blocks = n_bits / block_bits;
    g->n_words = (n_bits + rbspec::kMetaBits + 63) / 64;
    return true;
}

// metadata block of SeqAn's store(): {noOfBins, noOfHashFunc, kmerSize, 0} as u64 at bit n_bits
void read_metadata(const uint64_t *tail, unsigned shift, uint64_t meta[4])
{
    for (int j = 0; j < 4; ++j) {
        meta[j] = tail[j] >> shift;
        if (shift) meta[j] |= tail[j + 1] << (64 - shift);
    }
}

void write_metadata(uint64_t *tail, unsigned shift, const uint64_t meta[4])
{
    for (int j = 0; j < 4; ++j) {
        if (shift == 0) {
            tail[j] = meta[j];
        } else {
            const uint64_t low = (1ULL << shift) - 1;
            tail[j] = (tail[j] & low) | (meta[j] << shift);
            tail[j + 1] = (tail[j + 1] & ~low) | (meta[j] >> (64 - shift));
        }
    }
}

bool metadata_plausible(const uint64_t meta[4], uint64_t n_bits)
{
    return meta[0] != 0 && meta[0] <= n_bits && meta[1] != 0 && meta[1] <= 64 && meta[2] != 0 && meta[2] <= 255;
}

// Opens a .ibf, validates the sdsl framing (u64 bit size + ceil(size/64) words) and the trailing
// metadata; leaves the stream positioned at the first payload word.
int open_ibf_stream(const char *path, FILE **fp_out, rb_ibf_info *geo)
{
    if (!path) return fail(RB_ERR_INVALID_ARG, "null path");
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return fail(RB_ERR_MISSING_FILE, std::string("cannot open IBF file ") + path + ": " + std::strerror(errno));
    uint64_t bit_size = 0;
    if (std::fread(&bit_size, 8, 1, fp) != 1 || bit_size < rbspec::kMetaBits) {
        std::fclose(fp);
        return fail(RB_ERR_PARSE_IBF, std::string(path) + ": not an IBF (no bit-vector header)");
    }
    if (fseeko(fp, 0, SEEK_END) != 0) { std::fclose(fp); return fail(RB_ERR_PARSE_IBF, "seek failed"); }
    const off_t fsz = ftello(fp);
    const uint64_t n_words = (bit_size + 63) / 64;
    if (fsz < 0 || (uint64_t)fsz != 8 + 8 * n_words) {
        std::fclose(fp);
        return fail(RB_ERR_PARSE_IBF, std::string(path) + ": size does not match the bit-vector header");
    }
    const uint64_t n_bits = bit_size - rbspec::kMetaBits;
    const uint64_t mw = n_bits >> 6;
    const unsigned ms = (unsigned)(n_bits & 63);
    uint64_t tail[5] = {0, 0, 0, 0, 0};
    const size_t tail_words = (size_t)(n_words - mw);
    if (fseeko(fp, (off_t)(8 + 8 * mw), SEEK_SET) != 0 || std::fread(tail, 8, tail_words, fp) != tail_words) {
        std::fclose(fp);
        return fail(RB_ERR_PARSE_IBF, std::string(path) + ": truncated metadata");
    }
    uint64_t meta[4];
    read_metadata(tail, ms, meta);
    if (!metadata_plausible(meta, n_bits) || !geometry_from(meta[0], meta[1], meta[2], n_bits, geo) || geo->n_blocks == 0) {
        std::fclose(fp);
        return fail(RB_ERR_PARSE_IBF, std::string(path) + ": metadata block does not describe an IBF");
    }
    // The file parsed; say what is odd about it all the same.  The metadata order {noOfBins, noOfHashFunc, kmerSize, spare}
    // is recalled from SeqAn, not read from the reference tree (ibf_spec.h): values the reference never writes are the
    // first hint that a file contradicts that recollection.
    {
        std::string w;
        if (meta[3] != 0) w += "spare metadata word is " + std::to_string(meta[3]) + ", expected 0; ";
        if (meta[1] != 3) w += "noOfHashFunc = " + std::to_string(meta[1]) + ", the reference always builds with 3 (IBFConfig.hpp:71); ";
        if (meta[2] < 4 || meta[2] > 32) w += "kmerSize = " + std::to_string(meta[2]) + " is outside 4..32; ";
        if (meta[0] > (1ull << 24)) w += "noOfBins = " + std::to_string(meta[0]) + " is implausibly large; ";
        if (!w.empty()) w = std::string(path) + ": " + w + "check the layout constants in ibf_spec.h (rb_dibf_compare / --verify-ibf)";
        set_warning(w);
    }
    fseeko(fp, 8, SEEK_SET);
    *fp_out = fp;
    return RB_OK;
}

// ---- error model: src/IBF/IBF.hpp:268-338 ------------------------------------------------
static double rational_approximation(double t)
{
    // Abramowitz & Stegun 26.2.23
    static const double c0 = 2.515517, c1 = 0.802853, c2 = 0.010328;
    static const double d1 = 1.432788, d2 = 0.189269, d3 = 0.001308;
    const double num = (c2 * t + c1) * t + c0;
    const double den = ((d3 * t + d2) * t + d1) * t + 1.0;
    return t - num / den;
}

bool normal_cdf_inverse(double p, double *z)
{
    if (!(p > 0.0 && p < 1.0)) return false;  // the reference throws std::invalid_argument
    *z = (p < 0.5) ? -rational_approximation(std::sqrt(-2.0 * std::log(p)))
                   : rational_approximation(std::sqrt(-2.0 * std::log(1.0 - p)));
    return true;
}

// double -> uint16_t the way the reference's x86-64 binary does it (cvttsd2si to 32 bits, NaN and
// out-of-range give INT_MIN, then truncation to 16 bits)
static uint16_t narrow_u16(double x)
{
    int32_t i;
    if (std::isnan(x) || x >= 2147483648.0 || x <= -2147483649.0) i = INT32_MIN;
    else i = (int32_t)x;
    return (uint16_t)(uint32_t)i;
}

bool calculate_ci(double r, uint8_t k8, uint32_t readlen, double confidence, uint16_t *low, uint16_t *high)
{
    const double k = (double)k8;
    const double q = 1.0 - std::pow(1.0 - r, (int)k8);
    const double L = (double)readlen - k + 1.0;
    const double one_q = 1.0 - q;
    const double term1 = L * one_q * (q * (2.0 * k + (2.0 / r) - 1.0) - 2.0 * k);
    const double term2 = k * (k - 1.0) * std::pow(one_q, 2.0);
    const double term3 = (2.0 * one_q / (std::pow(r, 2.0))) * ((1.0 + (k - 1.0) * one_q) * r - q);
    const double varN = term1 + term2 + term3;
    const double alpha = 1 - confidence;
    double z = 0.0;
    const bool ok = normal_cdf_inverse(1.0 - alpha / 2.0, &z);
    const double sd = std::sqrt(varN);
    *low = narrow_u16(std::floor(L * q - z * sd));
    *high = narrow_u16(std::ceil(L * q + z * sd));
    return ok;
}

uint16_t threshold_u16(uint64_t readlen, uint64_t kmer_size, double r, double confidence)
{
    uint16_t lo = 0, hi = 0;
    calculate_ci(r, (uint8_t)kmer_size, (uint32_t)readlen, confidence, &lo, &hi);
    const uint16_t len16 = (uint16_t)readlen;  // "uint16_t readlen = seqan::length(...)"
    const int64_t t = (int64_t)len16 - (int64_t)kmer_size + 1 - (int64_t)hi;
    return (uint16_t)(int16_t)(uint16_t)(uint64_t)t;  // int16_t threshold, received as uint16_t
}

}  // namespace rb

using namespace rb;

extern "C" {

const char *rb_status_string(int status)
{
    switch (status) {
    case RB_OK: return "ok";
    case RB_ERR_NULL_FILTER: return "no IBF provided to classify the read (NullFilterException)";
    case RB_ERR_SHORT_READ: return "read shorter than k-mer size (ShortReadException)";
    case RB_ERR_COUNT_KMER: return "error counting k-mers (CountKmerException)";
    case RB_ERR_MISSING_FILE: return "IBF file missing or unreadable (MissingIBFFileException)";
    case RB_ERR_PARSE_IBF: return "file is not a valid IBF (ParseIBFFileException)";
    case RB_ERR_BAD_CHUNK: return "chunk starts beyond the end of the read";
    case RB_ERR_STORE: return "could not store IBF (StoreFilterException)";
    case RB_ERR_INVALID_ARG: return "invalid argument";
    case RB_ERR_UNSUPPORTED: return "filter geometry not supported by the GPU kernels";
    case RB_ERR_NO_DEVICE: return "no gfx950 device available (this engine has no CPU fallback)";
    case RB_ERR_HIP: return "HIP runtime error";
    case RB_ERR_NOMEM: return "out of memory";
    default: return "unknown status";
    }
}

const char *rb_last_error(void) { return g_last_error.c_str(); }
const char *rb_version(void) { return "readbouncer_amd 0.1 (gfx950)"; }

int rb_ibf_create(uint64_t n_bins, uint64_t n_hash, uint64_t kmer_size, uint64_t n_bits, rb_ibf **out)
{
    if (!out) return fail(RB_ERR_INVALID_ARG, "null out");
    rb_ibf_info g;
    if (!geometry_from(n_bins, n_hash, kmer_size, n_bits, &g)) return fail(RB_ERR_INVALID_ARG, "bad IBF geometry");
    rb_ibf *f = new (std::nothrow) rb_ibf();
    if (!f) return fail(RB_ERR_NOMEM, "alloc");
    f->geo = g;
    f->words = (uint64_t *)std::calloc(g.n_words ? g.n_words : 1, 8);
    if (!f->words) { delete f; return fail(RB_ERR_NOMEM, "cannot allocate IBF image"); }
    *out = f;
    return RB_OK;
}

int rb_ibf_open(const char *path, rb_ibf **out)
{
    if (!out) return fail(RB_ERR_INVALID_ARG, "null out");
    FILE *fp = nullptr;
    rb_ibf_info g;
    int st = open_ibf_stream(path, &fp, &g);
    if (st != RB_OK) return st;
    rb_ibf *f = new (std::nothrow) rb_ibf();
    if (!f) { std::fclose(fp); return fail(RB_ERR_NOMEM, "alloc"); }
    f->geo = g;
    f->words = (uint64_t *)std::malloc(g.n_words * 8);
    if (!f->words) { std::fclose(fp); delete f; return fail(RB_ERR_NOMEM, "cannot allocate IBF image"); }
    // (the stream is positioned behind the 8-byte header; the words are read past it, by several threads for a large file)
    rb::IoGang gang(rb::io_threads((size_t)g.n_words * 8));
    if (!gang.pread(fileno(fp), (off_t)8, f->words, (size_t)g.n_words * 8)) {
        std::fclose(fp);
        rb_ibf_close(f);
        return fail(RB_ERR_PARSE_IBF, std::string(path) + ": short read");
    }
    std::fclose(fp);
    *out = f;
    return RB_OK;
}

int rb_ibf_store(const rb_ibf *f, const char *path)
{
    if (!f || !path) return fail(RB_ERR_INVALID_ARG, "null argument");
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return fail(RB_ERR_STORE, std::string("cannot create ") + path + ": " + std::strerror(errno));
    const rb_ibf_info &g = f->geo;
    const uint64_t bit_size = g.n_bits + rbspec::kMetaBits;
    const uint64_t mw = g.n_bits >> 6;
    const unsigned ms = (unsigned)(g.n_bits & 63);
    const size_t tail_words = (size_t)(g.n_words - mw);
    uint64_t tail[6] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < tail_words; ++i) tail[i] = f->words[mw + i];
    const uint64_t meta[4] = {g.n_bins, g.n_hash, g.kmer_size, 0};
    write_metadata(tail, ms, meta);
    bool ok = std::fwrite(&bit_size, 8, 1, fp) == 1;
    if (ok && mw) ok = std::fwrite(f->words, 8, mw, fp) == mw;
    if (ok) ok = std::fwrite(tail, 8, tail_words, fp) == tail_words;
    if (std::fclose(fp) != 0) ok = false;
    return ok ? RB_OK : fail(RB_ERR_STORE, std::string("short write to ") + path);
}

int rb_ibf_get_info(const rb_ibf *f, rb_ibf_info *info)
{
    if (!f || !info) return fail(RB_ERR_INVALID_ARG, "null argument");
    *info = f->geo;
    return RB_OK;
}

uint64_t *rb_ibf_words(rb_ibf *f) { return f ? f->words : nullptr; }

void rb_ibf_close(rb_ibf *f)
{
    if (!f) return;
    std::free(f->words);
    delete f;
}

int rb_is_ibf_file(const char *path)
{
    FILE *fp = nullptr;
    rb_ibf_info g;
    if (open_ibf_stream(path, &fp, &g) != RB_OK) return 0;
    std::fclose(fp);
    return 1;
}

const char *rb_last_warning(void) { return rb::last_warning().c_str(); }

int rb_calculate_ci(double error_rate, uint8_t kmer_size, uint32_t readlen, double significance, uint16_t *low,
                    uint16_t *high)
{
    uint16_t lo = 0, hi = 0;
    const bool ok = calculate_ci(error_rate, kmer_size, readlen, significance, &lo, &hi);
    if (low) *low = lo;
    if (high) *high = hi;
    return ok ? RB_OK : fail(RB_ERR_INVALID_ARG, "significance outside (0,1): NormalCDFInverse would throw");
}

uint16_t rb_threshold(uint64_t readlen, uint64_t kmer_size, double error_rate, double significance)
{
    return threshold_u16(readlen, kmer_size, error_rate, significance);
}

uint64_t rb_calculate_filter_size_bits(uint64_t fragment_length, uint64_t kmer_size, uint64_t hash_functions,
                                       double max_fp, uint64_t n_bins)
{
    const uint64_t kmers_per_bin = fragment_length - kmer_size + 1;
    const uint64_t padded_bins = (uint64_t)(std::floor(((double)n_bins / 64.0) + 1) * 64);
    const double per_hash_fp = std::pow((double)max_fp, 1.0 / (double)hash_functions);
    const double root = std::pow(1 - per_hash_fp, 1.0 / ((double)(hash_functions * kmers_per_bin)));
    const uint64_t bin_size_bits = (uint64_t)std::ceil(-1 / (root - 1));
    return bin_size_bits * padded_bins;
}

size_t rb_cut_out_nnns(const char *seq, size_t len, char *out)
{
    size_t n = 0, pos = 0;
    while (pos < len) {
        while (pos < len && seq[pos] == 'N') ++pos;  // skip the N stretch
        if (pos >= len) break;
        size_t stop = pos;
        while (stop < len && seq[stop] != 'N') ++stop;
        size_t take = stop - pos;
        if (stop >= len) take -= 1;  // the reference drops the final base of a piece that reaches the end
        std::memcpy(out + n, seq + pos, take);
        n += take;
        pos = stop;
    }
    return n;
}

// 2-bit payload + N bitmap of a batch; every read starts at a byte boundary in both arrays
int rb_pack_reads(const char *seqs, const uint64_t *offsets, const uint32_t *lens, size_t n_reads, uint8_t *packed,
                  uint64_t *packed_offsets, uint8_t *nmask, uint64_t *nmask_offsets, uint64_t *packed_bytes,
                  uint64_t *nmask_bytes)
{
    if (n_reads && (!seqs || !offsets || !lens)) return fail(RB_ERR_INVALID_ARG, "null input buffer");
    uint64_t pp = 0, np = 0;
    for (size_t r = 0; r < n_reads; ++r) {
        const uint64_t pb = ((uint64_t)lens[r] + 3) / 4, nb = ((uint64_t)lens[r] + 7) / 8;
        if (packed_offsets) packed_offsets[r] = pp;
        if (nmask_offsets) nmask_offsets[r] = np;
        if (packed && nmask) {
            std::memset(packed + pp, 0, pb);
            std::memset(nmask + np, 0, nb);
            const unsigned char *s = (const unsigned char *)seqs + offsets[r];
            for (uint32_t i = 0; i < lens[r]; ++i) {
                const uint32_t o = rbspec::dna5_ord(s[i]);
                if (o < 4) packed[pp + (i >> 2)] |= (uint8_t)(o << ((i & 3u) << 1));
                else nmask[np + (i >> 3)] |= (uint8_t)(1u << (i & 7u));
            }
        }
        pp += pb;
        np += nb;
    }
    if (packed_bytes) *packed_bytes = pp;
    if (nmask_bytes) *nmask_bytes = np;
    return RB_OK;
}

size_t rb_fragment_bounds(uint64_t len, uint64_t fragment_length, uint64_t kmer_size, uint64_t overlap_length,
                          uint64_t *starts, uint64_t *ends, size_t cap)
{
    size_t n = 0;
    if (fragment_length == 0) return 0;
    int64_t idx = 0;
    int64_t start = 1 - (int64_t)overlap_length;  // first fragment uses overlap_length (IBFBuild.cpp:166)
    if (start < 0) start = 0;
    while (start < (int64_t)len - 1) {
        uint64_t stop = (uint64_t)(idx + 1) * fragment_length;
        if (stop > len) stop = len;
        if (n < cap) {
            if (starts) starts[n] = (uint64_t)start;
            if (ends) ends[n] = stop;
        }
        ++n;
        ++idx;
        start = idx * (int64_t)fragment_length - (int64_t)kmer_size + 1;  // later ones overlap by k-1
    }
    return n;
}

}  // extern "C"
