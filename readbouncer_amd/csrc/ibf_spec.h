// ibf_spec.h -- the single place holding the layout/hash constants of the IBF as SeqAn's
// BinningDirectory<InterleavedBloomFilter, BDConfig<Dna5,Normal,Uncompressed>> defines them.
//
// The SeqAn fork the reference fetches at configure time (src/seqan/CMakeLists.txt.in:28-30,
// JensUweUlrich/seqan branch "SeqAn") is not in the reference tree; these constants restate the
// published binning_directory_interleaved_bloom_filter.h.  If a real .ibf ever contradicts
// them, this header is the only edit.  Reference call sites that depend on them:
//   seqan::count      src/IBF/IBFClassify.cpp:97-98,149-150
//   seqan::insertKmer src/IBF/IBFBuild.cpp:190
//   seqan::store      src/IBF/IBFBuild.cpp:505      seqan::retrieve  src/IBF/IBFBuild.cpp:343,360
//
// Where each constant was recalled from (upstream seqan/seqan, branch develop at the time the binning directory was
// added, 2017-2018; the fork's branch is not pinned to a commit, so these are names to look for, not line numbers):
//   include/seqan/binning_directory/binning_directory_interleaved_bloom_filter.h
//     struct BinningDirectory<InterleavedBloomFilter, TConfig>: members noOfBins, noOfHashFunc, kmerSize, noOfBits,
//     noOfBlocks, binWidth, blockBitSize, preCalcValues, `shiftValue = 27`, `seedValue = 0x90b45d39fb6da1fa`,
//     `intSize = 0x40`, `filterMetadataSize{256}`;  init(): binWidth = ceil(noOfBins / intSize), blockBitSize =
//     binWidth * intSize, noOfBlocks = noOfBits / blockBitSize, preCalcValues[i] = i ^ (kmerSize * seedValue);
//     hashToIndex(): `hash ^= hash >> shiftValue; hash %= noOfBlocks; hash *= blockBitSize`;
//     insertKmer(): vecIndex = preCalcValues[i] * kmerHash; hashToIndex; `vecIndex += binNo; set_pos(vecIndex)`;
//     select()/count(): for every 64-bit batch of a block AND the words at the h positions, then ++counts[binNo] per set bit;
//     getMetadata()/setMetadata(): noOfBins at bit noOfBits, noOfHashFunc at +64, kmerSize at +128 (64-bit fields).
//   include/seqan/binning_directory/bitvector_uncompressed.h -- Bitvector<Uncompressed>: an sdsl::bit_vector of
//     noOfBits + filterMetadataSize bits; store()/retrieve() = sdsl::store_to_file / load_from_file, i.e. the
//     int_vector<1> serialisation of sdsl-lite v2.1.1 (u64 bit count, then the 64-bit words, LSB first).
//   include/seqan/index/shape_base.h -- Shape<Dna5, SimpleShape>: hash()/hashNext() = base-|alphabet| polynomial of the
//     ordinal values, most significant base first (the k-mer value fed to hashToIndex).
//   include/seqan/basic/alphabet_residue.h + alphabet_residue_tabs.h -- Dna5 ordinals A0 C1 G2 T3 N4, the char ->
//     Dna5 translation table (everything but ACGTacgt[Uu] -> N).
// `rb_dibf_compare` / `readbouncer_amd_cli --verify-ibf` check all of them at once against a filter file and the FASTA
// it was built from.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RB_HD __host__ __device__ __forceinline__
#else
#define RB_HD inline
#endif

namespace rbspec {

constexpr uint64_t kSeed = 0x90b45d39fb6da1faULL;  // seedValue
constexpr unsigned kShift = 27;                     // shiftValue
constexpr unsigned kIntSize = 64;                   // intSize: bits per bin-column word
constexpr unsigned kMetaBits = 256;                 // filterMetadataSize (4 x u64 at bit n_bits)
constexpr unsigned kMaxHash = 8;                    // engine limit on noOfHashFunc
constexpr unsigned kMaxKmer = 32;                   // engine limit on kmerSize (u64 base-5 value wraps above 27)

// preCalcValues[i] = i ^ (kmerSize * seedValue)
RB_HD uint64_t precalc(uint64_t kmer_size, uint64_t i) { return i ^ (kmer_size * kSeed); }

// (seqan::Dna5String) conversion: A/a 0, C/c 1, G/g 2, T/t/U/u 3, everything else N = 4
RB_HD uint32_t dna5_ord(uint32_t c)
{
    c &= 0xDFu;  // fold case: 'a'..'z' -> 'A'..'Z' (non letters land on values that match nothing below)
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : (c == 'T' || c == 'U') ? 3u : 4u;
}

// What the reverse strand holds where the read has an N (Dna5 ordinal 4).  The reference's reverse-complement view is
//   ModifiedString<ModifiedString<Dna5String, ModComplementDna>, ModReverse>            (src/IBF/IBF.hpp:96-97)
// i.e. the FOUR-letter functor FunctorComplement<Dna> over a Dna5 host, not ModComplementDna5.  RECALLED from SeqAn2
// (include/seqan/modifier/modifier_functors.h, modifier_alphabet.h; include/seqan/basic/alphabet_residue.h): the functor's
// argument type is Dna, the Dna5 -> Dna assignment is `target.value = source.value & 0x03` (N -> A), the complement of A is
// T, and the Dna result is hashed by Shape<Dna5> with its ordinal unchanged.  So the reverse strand sees T (3) where the
// read has N; the forward strand still hashes N as ordinal 4.  "N stays N" (4) would be ModComplementDna5, which the
// reference does not name.  Like the hash constants this is unverifiable here; tools/make_reference_fixtures.cpp writes
// the two N-containing reads that decide it, and rb_engine_set_revcomp_of_n flips it at run time without a rebuild.
constexpr uint32_t kRevCompOfN = 3;

// complement on ordinals under that view: A<->T, C<->G, N -> comp_of_n
RB_HD uint32_t dna5_comp(uint32_t o, uint32_t comp_of_n = kRevCompOfN) { return o < 4u ? 3u - o : comp_of_n; }

// floor(2^64 / d) for d >= 2 (d = 1 handled by callers: everything maps to block 0)
inline uint64_t fastmod_magic(uint64_t d)
{
    // 2^64 / d = (2^64 - 1) / d + (((2^64 - 1) % d + 1 == d) ? 1 : 0)
    uint64_t q = ~0ULL / d, r = ~0ULL % d;
    return (r + 1 == d) ? q + 1 : q;
}

RB_HD uint64_t mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// x mod d with d < 2^32, magic = floor(2^64/d).  q_est = floor(x*magic / 2^64) is q or q-1
// (x*magic/2^64 > x/d - 1), so one conditional subtraction finishes it.
RB_HD uint32_t fastmod(uint64_t x, uint32_t d, uint64_t magic)
{
    uint64_t q = mulhi64(x, magic);
    uint64_t r = x - q * d;
    if (r >= d) r -= d;
    return (uint32_t)r;
}

// hashToIndex(): idx = preCalc*v; idx ^= idx >> 27; idx %= noOfBlocks  (block NUMBER)
RB_HD uint32_t block_index(uint64_t kmer_value, uint64_t pre, uint32_t n_blocks, uint64_t magic, uint32_t pow2_mask)
{
    uint64_t x = pre * kmer_value;
    x ^= x >> kShift;
    if (pow2_mask != 0xFFFFFFFFu) {  // n_blocks is a power of two <= 2^31: mask = n_blocks-1
        return (uint32_t)x & pow2_mask;
    }
    return fastmod(x, n_blocks, magic);
}

RB_HD uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// synthetic filler: each bit ~ Bernoulli(55/256); 55/256 = .00110111b folded from the last digit
RB_HD uint64_t synth_word(uint64_t seed, uint64_t word_index)
{
    uint64_t r[8];
    for (int i = 0; i < 8; ++i) r[i] = mix64(seed + (word_index * 8 + (uint64_t)i + 1) * 0x9E3779B97F4A7C15ULL);
    uint64_t acc = r[7];
    acc |= r[6];
    acc |= r[5];
    acc &= r[4];
    acc |= r[3];
    acc |= r[2];
    acc &= r[1];
    acc &= r[0];
    return acc;
}

}  // namespace rbspec
