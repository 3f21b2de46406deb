/*
 * readbouncer_amd.h -- C ABI of the MI355X-native IBF read-classification engine.
 *
 * This is the drop-in boundary for ONE path of ReadBouncer: src/IBF classify
 * (k-mer extraction -> h-fold hash -> IBF bulk-contains -> per-bin counting ->
 * error-model threshold -> unblock/keep decision).  The reference exposes that path as
 * a C++ static library whose types leak SeqAn templates (src/IBF/CMakeLists.txt:5,
 * src/IBF/IBF.hpp:92-94), so there is no binary FFI to bind; every entry point below
 * names the reference interface it replaces.  INTEGRATION.md shows the reference-side
 * shim.  include/readbouncer_amd.hpp is the C++ mirror (interleave::Read, IBF, IBFMeta,
 * ClassifyConfig, exceptions) over these calls.
 *
 * All compute entry points need a gfx950 device and fail with RB_ERR_NO_DEVICE /
 * RB_ERR_HIP otherwise -- there is no CPU fallback.  Plain pointers and sizes only.
 */
#ifndef READBOUNCER_AMD_H_
#define READBOUNCER_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RB_API __attribute__((visibility("default")))

/* ---- status codes (the reference throws; src/IBF/IBFExceptions.hpp) ------------------ */
enum rb_status {
    RB_OK = 0,
    RB_ERR_NULL_FILTER = 1,  /* NullFilterException      IBFExceptions.hpp:178 */
    RB_ERR_SHORT_READ = 2,   /* ShortReadException       IBFExceptions.hpp:96  */
    RB_ERR_COUNT_KMER = 3,   /* CountKmerException       IBFExceptions.hpp:123 */
    RB_ERR_MISSING_FILE = 4, /* MissingIBFFileException  IBFExceptions.hpp:317 */
    RB_ERR_PARSE_IBF = 5,    /* ParseIBFFileException    IBFExceptions.hpp:344 */
    RB_ERR_BAD_CHUNK = 6,    /* chunk start beyond read end (classify.hpp:264-273, undefined there) */
    RB_ERR_STORE = 7,        /* StoreFilterException */
    RB_ERR_INVALID_ARG = 8,
    RB_ERR_UNSUPPORTED = 9,  /* filter geometry outside what the kernels handle */
    RB_ERR_NO_DEVICE = 10,   /* no gfx950 GPU visible: the engine has no CPU fallback */
    RB_ERR_HIP = 11,         /* a HIP runtime call failed, see rb_last_error() */
    RB_ERR_NOMEM = 12
};

RB_API const char *rb_status_string(int status);
/* thread-local text of the last failure on the calling thread */
RB_API const char *rb_last_error(void);
RB_API const char *rb_version(void);
/* number of visible HIP devices, or a negative rb_status */
RB_API int rb_device_count(void);

/* ---- filter geometry ------------------------------------------------------------------
 * Mirrors the public members of interleave::TIbf =
 * seqan::BinningDirectory<InterleavedBloomFilter, BDConfig<Dna5,Normal,Uncompressed>>
 * that the reference reads (noOfBins: IBFClassify.cpp:27,58; kmerSize: :102,154,192,248;
 * getNumberOfBins/getKmerSize: IBFBuild.cpp:380-381). */
typedef struct rb_ibf_info {
    uint64_t n_bins;     /* noOfBins */
    uint64_t n_hash;     /* noOfHashFunc */
    uint64_t kmer_size;  /* kmerSize */
    uint64_t n_bits;     /* noOfBits, without the 256 metadata bits */
    uint64_t bin_width;  /* 64-bit words per block = ceil(n_bins/64) */
    uint64_t n_blocks;   /* n_bits / (64*bin_width) */
    uint64_t n_words;    /* payload words incl. metadata = (n_bits+256+63)/64 */
} rb_ibf_info;

/* ---- host-side filter image (.ibf file <-> memory) ------------------------------------ */
typedef struct rb_ibf rb_ibf;

/* IBF::load_filter -> seqan::retrieve (src/IBF/IBFBuild.cpp:329-396).
 * RB_ERR_MISSING_FILE if the file cannot be opened, RB_ERR_PARSE_IBF if it is not an IBF. */
RB_API int rb_ibf_open(const char *path, rb_ibf **out);
/* TIbf(bins, hash_functions, kmer_size, filter_size_bits) (src/IBF/IBFBuild.cpp:465): zeroed */
RB_API int rb_ibf_create(uint64_t n_bins, uint64_t n_hash, uint64_t kmer_size, uint64_t n_bits, rb_ibf **out);
/* seqan::store (src/IBF/IBFBuild.cpp:505,307) */
RB_API int rb_ibf_store(const rb_ibf *f, const char *path);
RB_API int rb_ibf_get_info(const rb_ibf *f, rb_ibf_info *info);
/* payload words (n_words of them), owned by the handle */
RB_API uint64_t *rb_ibf_words(rb_ibf *f);
RB_API void rb_ibf_close(rb_ibf *f);
/* ConfigReader::filterException (src/config/configReader.cpp:210-224): 1 = IBF file, 0 = not */
RB_API int rb_is_ibf_file(const char *path);

/* ---- error model (host, double) -------------------------------------------------------
 * calculateCI / NormalCDFInverse (src/IBF/IBF.hpp:320-338, 284-308) and the threshold
 * expression of count_matches (src/IBF/IBFClassify.cpp:154-162), returned as the
 * uint16_t that max_matches receives (negative int16 thresholds wrap). */
RB_API int rb_calculate_ci(double error_rate, uint8_t kmer_size, uint32_t readlen, double significance,
                           uint16_t *low, uint16_t *high);
RB_API uint16_t rb_threshold(uint64_t readlen, uint64_t kmer_size, double error_rate, double significance);

/* ---- build-side helpers (host) --------------------------------------------------------- */
/* IBF::calculate_filter_size_bits (src/IBF/IBFBuild.cpp:404-413) */
RB_API uint64_t rb_calculate_filter_size_bits(uint64_t fragment_length, uint64_t kmer_size,
                                              uint64_t hash_functions, double max_fp, uint64_t n_bins);
/* IBF::cutOutNNNs + concatenation (src/IBF/IBFBuild.cpp:81-88,112-132); out holds >= len bytes */
RB_API size_t rb_cut_out_nnns(const char *seq, size_t len, char *out);
/* fragment loop of add_sequences_to_filter (src/IBF/IBFBuild.cpp:165-204): fills
 * starts/ends (capacity cap) and returns the number of fragments of a sequence of length len */
RB_API size_t rb_fragment_bounds(uint64_t len, uint64_t fragment_length, uint64_t kmer_size,
                                 uint64_t overlap_length, uint64_t *starts, uint64_t *ends, size_t cap);

/* ---- device-resident filter (the IBF in HBM) -------------------------------------------
 * Layout in HBM: the reference's block-major bit matrix with every block starting on a boundary that suits the
 * memory system -- block b at word b*stride, stride >= bin_width (equal when bin_width is a multiple of 16 words, i.e.
 * the .ibf payload is then uploaded verbatim; otherwise blocks are padded to the next power of two / multiple of
 * 128 bytes).  Bin j of block b is bit b*64*stride + j.  Upload, download, build and resize convert; the .ibf file
 * format never changes. */
typedef struct rb_dibf rb_dibf;

RB_API int rb_dibf_create(int device, uint64_t n_bins, uint64_t n_hash, uint64_t kmer_size, uint64_t n_bits,
                          rb_dibf **out);
RB_API int rb_dibf_upload(int device, const rb_ibf *host, rb_dibf **out);
/* load_filter straight into HBM, streamed through pinned staging (no full host copy) */
RB_API int rb_dibf_open(int device, const char *path, rb_dibf **out);
RB_API int rb_dibf_download(const rb_dibf *f, rb_ibf **out);
/* replica of a resident filter on another GPU (or the same one), copied device to device -- over xGMI when the pair
 * has peer access -- in its HBM layout: no host image, no conversion */
RB_API int rb_dibf_clone_to(const rb_dibf *src, int device, rb_dibf **out);
/* the same, reporting how the copy travelled: *used_peer = 1 when the destination mapped the source (peer access, xGMI
 * between two GPUs of a node), 0 for the runtime's staged path or a same-device copy; *seconds = wall time of the copy */
RB_API int rb_dibf_clone_to_ex(const rb_dibf *src, int device, rb_dibf **out, int *used_peer, double *seconds);
RB_API int rb_dibf_get_info(const rb_dibf *f, rb_ibf_info *info);
/* The device image itself.  Engines that keep a merged copy of several filters (rb_engine_set_merge) notice changes made through
 * rb_dibf_insert / rb_dibf_add_sequence / rb_dibf_fill_synth by themselves; a caller that WRITES through this pointer calls
 * rb_dibf_touch afterwards (with its writes complete), or merged copies keep serving the old bits. */
RB_API void *rb_dibf_device_words(rb_dibf *f);
RB_API int rb_dibf_touch(rb_dibf *f);
/* words between consecutive blocks of the device image (see above) */
RB_API uint64_t rb_dibf_device_stride(const rb_dibf *f);
RB_API int rb_dibf_device(const rb_dibf *f);
RB_API void rb_dibf_free(rb_dibf *f);
/* resizeBins of IBF::update_filter (src/IBF/IBFBuild.cpp:274): same blocks and hash positions, every block widened
 * to ceil(new_bins/64) words, new bins empty, noOfBits = noOfBlocks * new block size.  Returns a new filter. */
RB_API int rb_dibf_resize_bins(const rb_dibf *f, uint64_t new_bins, rb_dibf **out);
/* seqan::insertKmer for a batch of fragments (src/IBF/IBFBuild.cpp:189-190) on the GPU:
 * fragment i = seq[starts[i], ends[i]) goes to bin bins[i]. seq is host ASCII. */
RB_API int rb_dibf_insert(rb_dibf *f, const char *seq, size_t len, const uint64_t *starts,
                          const uint64_t *ends, const uint64_t *bins, size_t n_fragments);
/* one reference sequence through the reference fragmenter; *next_bin = first_bin + #fragments */
RB_API int rb_dibf_add_sequence(rb_dibf *f, const char *seq, size_t len, uint64_t fragment_length,
                                uint64_t overlap_length, uint64_t first_bin, uint64_t *next_bin);

/* First-contact check for a filter that came from somewhere else (a .ibf written by the reference).  The hash constants
 * and the bit layout of the IBF live in a dependency that is not part of the reference tree (ibf_spec.h), so a filter
 * file is the first place where a wrong constant would show: re-insert the reference sequences the file was built from
 * (the reference fragmenter + insertKmer, src/IBF/IBFBuild.cpp:165-204,190) into an EMPTY filter of the same geometry
 * and compare.  With matching constants the rebuilt bits are a subset of the file's bits: new_bits == 0 (SURVEY 7:
 * "re-inserting a known reference must set no new bits") and rebuilt_bits / file_bits says how much of the file the
 * given sequences explain; under a different seed, shift, k-mer encoding or block order about (1 - load) of the
 * rebuilt bits land on clear positions. */
typedef struct rb_ibf_compare {
    uint64_t file_bits;     /* bits set in the block payload of the filter under test */
    uint64_t rebuilt_bits;  /* bits the re-insertion sets */
    uint64_t new_bits;      /* of those, bits that are clear in the filter under test -- must be 0 */
    uint64_t payload_bits;  /* noOfBlocks * noOfBins: the positions insertKmer can reach */
} rb_ibf_compare;
RB_API int rb_dibf_compare(const rb_dibf *file_filter, const rb_dibf *rebuilt, rb_ibf_compare *out);
/* Text of what the last rb_ibf_open / rb_dibf_open / rb_is_ibf_file on this thread found odd about a file that still
 * parsed (non-zero spare metadata word, a hash-function count or k-mer size the reference never writes, tail bits set
 * beyond the last block); empty string when there was nothing. */
RB_API const char *rb_last_warning(void);

/* ---- classification engine --------------------------------------------------------------
 * One engine per GPU.  It borrows the filters (like the reference's
 * std::vector<IBFMeta>& DepletionFilters / TargetFilters, classify.hpp:142,
 * adaptive_sampling.hpp:214) and owns streams, threshold tables and workspaces. */
typedef struct rb_engine rb_engine;

RB_API int rb_engine_create(int device, rb_dibf *const *deplete, size_t n_deplete,
                            rb_dibf *const *target, size_t n_target, rb_engine **out);
RB_API void rb_engine_destroy(rb_engine *e);

enum rb_mode {
    /* check_unblock (src/main/adaptive_sampling.hpp:35-113): decision 0 wait / 1 unblock / 2 stop_receiving */
    RB_MODE_CHECK_UNBLOCK = 0,
    /* one chunk of classify_reads (src/main/classify.hpp:275-292, 58-111): decision 1 = classified */
    RB_MODE_CLASSIFY_CHUNK = 1,
    /* Read::classify(std::vector<TIbf>&) -> find_matches / select_matches (src/IBF/IBFClassify.cpp:181-226, 81-128,
     * 16-38): decision 1 = some filter of the list (deplete entries first, then target entries) holds a bin whose
     * forward or reverse count is >= the uint16_t threshold -- with a threshold of 0 that is every read, matches or
     * not; status RB_ERR_SHORT_READ when the read is shorter than the first filter's k */
    RB_MODE_CLASSIFY_ANY = 2
};

/* Batch form of Read::classify x3 + the decision, inputs and outputs in HOST memory.
 *   seqs/offsets/lens : read i is the ASCII bytes seqs[offsets[i] .. offsets[i]+lens[i])
 *                       ((seqan::Dna5String) conversion is done on the device)
 *   error_rate        : ClassifyConfig::error_rate, by value (the reference mutates a shared
 *                       config, adaptive_sampling.hpp:55-59); significance: 0.95 in the reference
 *   out_maxcount      : [n_reads x (n_deplete+n_target)] raw max k-mer count over bins and both
 *                       strands per filter, deplete filters first (before thresholding); may be NULL
 *   out_best_target   : [n_reads] Read::classify(TargetFilters) -> index or -1; may be NULL
 *   out_decision      : [n_reads] per rb_mode
 *   out_status        : [n_reads] RB_OK / RB_ERR_SHORT_READ / RB_ERR_NULL_FILTER per read --
 *                       one bad read never aborts the batch (classify.hpp:306-316)          */
RB_API int rb_classify_batch(rb_engine *e, const char *seqs, const uint64_t *offsets, const uint32_t *lens,
                             size_t n_reads, double error_rate, double significance, int mode,
                             uint16_t *out_maxcount, int32_t *out_best_target, uint8_t *out_decision,
                             uint8_t *out_status);

/* Pointer-array form of rb_classify_batch: read i = seq_ptrs[i][0 .. lens[i]) (one buffer per read, as a basecaller
 * hands them over; the reference's RTPair carries one std::string per read, src/interfaces/ont_read.hpp:24-61). */
RB_API int rb_classify_batch_ptrs(rb_engine *e, const char *const *seq_ptrs, const uint32_t *lens, size_t n_reads,
                                  double error_rate, double significance, int mode, uint16_t *out_maxcount,
                                  int32_t *out_best_target, uint8_t *out_decision, uint8_t *out_status);

/* Page-locked host memory for read buffers handed to rb_classify_batch: the copy to the GPU is then a plain DMA
 * (pageable buffers are pinned on the fly by the runtime, which costs more than the copy itself and serialises with
 * threads that page-fault on a memory-mapped read file).  Optional: any host pointer works. */
RB_API int rb_host_alloc(size_t bytes, void **out);
RB_API void rb_host_free(void *p);

/* Same with every buffer already resident in HBM (device pointers) and asynchronous on
 * `stream` (a hipStream_t, NULL = the engine's own stream, which is then synchronised
 * before returning).  max_len = an upper bound of lens[]: a longer read gets status RB_ERR_INVALID_ARG, and its row of
 * d_maxcount is then UNDEFINED (the narrow-filter kernels are built per max_len and write 0 for it) -- callers that consume raw
 * maxima without the decision stage (the bin-sharded layout) must not understate it.  The inputs must be complete on `stream`
 * (work that produces them on another stream has to be ordered before this call by the caller).  An engine's
 * workspaces (partial maxima, threshold tables, arrival counters) are per engine: calls on one engine may come from
 * several threads, but their GPU work must be ordered -- use one stream per engine, or order the streams with events;
 * for independent streams create one engine per stream (filters are shared, not copied). */
RB_API int rb_classify_batch_device(rb_engine *e, const void *d_seqs, const void *d_offsets, const void *d_lens,
                                    size_t n_reads, uint32_t max_len, double error_rate, double significance,
                                    int mode, void *d_maxcount, void *d_best_target, void *d_decision,
                                    void *d_status, void *stream);

/* Descriptor form of the device entry point: packed reads and on-GPU chunking.
 *   d_nmask != NULL : d_seqs holds 2 bits per base (A0 C1 G2 T3; base i of a read in bits 2*(i&3) of byte i>>2 of the
 *                     read's payload, d_offsets[] = byte offset of that payload) and d_nmask an N bitmap (bit i&7 of
 *                     byte i>>3, d_nmask_offsets[]); a flagged base is hashed as Dna5 ordinal 4 exactly like an
 *                     'N' in ASCII input.  rb_pack_reads builds both arrays on the host.
 *   chunk_start / chunk_length : classify bases [chunk_start, min(chunk_start+chunk_length, len)) of every read
 *                     (chunk_length 0 = to the end) -- chunk i of classify_reads is chunk_start = i*chunk_length
 *                     (src/main/classify.hpp:264-271); reads uploaded once serve every chunk iteration.  A chunk that
 *                     starts beyond the read's end gets status RB_ERR_BAD_CHUNK.
 *   d_read_ids      : optional u32[n_items]: work item j classifies read d_read_ids[j] (the reads still unclassified
 *                     after the previous chunk); outputs are indexed by work item.  d_lens/d_offsets stay per read. */
typedef struct rb_batch_desc {
    const void *d_seqs;
    const void *d_offsets;        /* u64 per read */
    const void *d_lens;           /* u32 per read: full read lengths */
    size_t n_items;               /* work items (= reads unless d_read_ids is given) */
    uint32_t max_len;             /* upper bound of the full read lengths */
    const void *d_nmask;          /* NULL = ASCII input */
    const void *d_nmask_offsets;  /* u64 per read */
    uint32_t chunk_start;
    uint32_t chunk_length;
    const void *d_read_ids;       /* NULL = identity */
} rb_batch_desc;
RB_API int rb_classify_batch_device_ex(rb_engine *e, const rb_batch_desc *desc, double error_rate, double significance,
                                       int mode, void *d_maxcount, void *d_best_target, void *d_decision,
                                       void *d_status, void *stream);
/* host helper for the packed form; call once with packed == NULL to obtain the sizes and offsets */
RB_API int rb_pack_reads(const char *seqs, const uint64_t *offsets, const uint32_t *lens, size_t n_reads, uint8_t *packed,
                         uint64_t *packed_offsets, uint8_t *nmask, uint64_t *nmask_offsets, uint64_t *packed_bytes,
                         uint64_t *nmask_bytes);

/* bin-sharded operation: restrict the engine to word columns
 * [rank*ceil(W/world) , ...) of every block; out_maxcount then holds PARTIAL maxima that the
 * caller combines with an all-reduce(max) before rb_decide_device. world=1 restores the default. */
RB_API int rb_engine_set_column_shard(rb_engine *e, int rank, int world);
/* decision stage alone (K2) on device-resident raw maxima */
RB_API int rb_decide_device(rb_engine *e, const void *d_maxcount, const void *d_lens, size_t n_reads,
                            uint32_t max_len, double error_rate, double significance, int mode,
                            void *d_best_target, void *d_decision, void *d_status, void *stream);

/* Same for a bin-sharded node: d_maxcount holds n_parts partial tables (the all-gathered outputs of the ranks, each
 * [n_reads x filters] u16, part_stride elements apart); the raw maximum of a (read, filter) is the max over them, taken
 * inside the decision kernel -- no separate reduction pass and no widening of the u16 values for a collective. */
RB_API int rb_decide_device_parts(rb_engine *e, const void *d_maxcount, uint32_t n_parts, uint64_t part_stride,
                                  const void *d_lens, size_t n_reads, uint32_t max_len, double error_rate,
                                  double significance, int mode, void *d_best_target, void *d_decision, void *d_status,
                                  void *stream);

/* ---- single-process multi-GPU pool -----------------------------------------------------------
 * One engine + one host thread per entry of devices[] (an entry may repeat), every filter replicated into each
 * device's HBM from its host image, batches cut into contiguous slices of ceil(n/parts) reads, no collective
 * (SURVEY 8e).  Batches with fewer than min_split reads per device are not split: they go to one device,
 * round-robin.  Outputs as rb_classify_batch. */
typedef struct rb_pool rb_pool;
RB_API int rb_pool_create(const int *devices, size_t n_devices, const rb_ibf *const *deplete, size_t n_deplete,
                          const rb_ibf *const *target, size_t n_target, rb_pool **out);
/* The same pool from .ibf FILES, without a host image of any filter (rb_pool_create needs rb_ibf images: 8 GiB on the host
 * and one PCIe upload per device for a GRCh38 filter): every file is streamed once into the HBM of devices[0]
 * (rb_dibf_open) and replicated from there to all other devices AT ONCE, device to device -- xGMI is point to point,
 * devices[0] has a link of its own to each peer, so the N-1 copies run side by side at link speed (no ring, no tree:
 * a tree would only add hops on a fully connected node).  A device that refuses peer access gets its copy by the
 * runtime's staged path; if that fails too it streams the file itself.  replication_seconds (may be NULL): wall time
 * of all device-to-device copies. */
RB_API int rb_pool_create_from_files(const int *devices, size_t n_devices, const char *const *deplete_paths, size_t n_deplete,
                                     const char *const *target_paths, size_t n_target, rb_pool **out,
                                     double *replication_seconds);
/* The same from filters that are already resident on some device (built there, or loaded once): every entry of `devices` gets
 * a replica of its own, copied device to device like above (the source's own device included -- the pool owns what it
 * classifies against; the caller keeps its filters). */
RB_API int rb_pool_create_from_device(const int *devices, size_t n_devices, rb_dibf *const *deplete, size_t n_deplete,
                                      rb_dibf *const *target, size_t n_target, rb_pool **out, double *replication_seconds);
RB_API void rb_pool_destroy(rb_pool *p);
RB_API size_t rb_pool_size(const rb_pool *p);
RB_API int rb_pool_classify_batch(rb_pool *p, const char *seqs, const uint64_t *offsets, const uint32_t *lens,
                                  size_t n_reads, double error_rate, double significance, int mode,
                                  uint16_t *out_maxcount, int32_t *out_best_target, uint8_t *out_decision,
                                  uint8_t *out_status);

/* ---- live micro-batch shim ----------------------------------------------------------------
 * Batch form of classify_live_reads (src/main/adaptive_sampling.hpp:214-356), the step between the reference's
 * classification_queue and action_queue: keeps the once_seen map, concatenates undecided chunks of a read,
 * applies the 1500 bp cut-off (:315) and maps decisions to actions like Data::sendActions
 * (src/minknow/Data.cpp:169-187).  Read ids are opaque byte strings. */
typedef struct rb_live rb_live;
RB_API int rb_live_create(rb_engine *e, double error_rate, double significance, uint32_t max_undecided_len,
                          rb_live **out);
RB_API void rb_live_destroy(rb_live *lv);
/* out_action[i]: 0 = none yet (read kept in once_seen), 1 = unblock_read, 2 = stop_receiving_data;
 * out_status[i] != RB_OK = the reference's caught-exception case (no action, state untouched);
 * out_classified_len[i] = length of the (possibly concatenated) sequence the decision was taken on. */
RB_API int rb_live_process(rb_live *lv, const char *ids, const uint64_t *id_offsets, const uint32_t *id_lens,
                           const char *seqs, const uint64_t *offsets, const uint32_t *lens, size_t n,
                           uint8_t *out_action, uint8_t *out_status, uint32_t *out_classified_len);
/* reads currently waiting for more data (size of once_seen) */
RB_API size_t rb_live_pending(rb_live *lv);
/* drop a read that ended on the sequencer before a decision was reached */
RB_API int rb_live_forget(rb_live *lv, const char *id, uint32_t id_len);

/* What the reverse strand holds where the read has an N.  The reference counts the second strand on
 * ModifiedString<ModifiedString<Dna5String, ModComplementDna>, ModReverse> (src/IBF/IBF.hpp:96-97, used at
 * src/IBF/IBFClassify.cpp:98,150): the four-letter complement functor over a Dna5 host, for which SeqAn converts N to A
 * (value & 3) before complementing -- the reverse strand sees T (ordinal 3), the forward strand hashes N as ordinal 4.
 * That is the default (rbspec::kRevCompOfN in readbouncer_amd/csrc/ibf_spec.h, a recalled SeqAn fact like the hash
 * constants).  ordinal = 4 gives the other candidate, "N stays N" (ModComplementDna5); anything else is refused.  Only
 * k-mers of the reverse strand that cover an N are affected. */
RB_API int rb_engine_set_revcomp_of_n(rb_engine *e, uint32_t ordinal);
/* The same for every engine this process creates FROM NOW ON (the C++ mirror and the CLI create their engines themselves:
 * interleave::set_revcomp_of_n, `readbouncer_amd_cli --revcomp-of-n 4`).  This is the only process-wide switch that changes a
 * result, and it is a call, not an environment variable. */
RB_API int rb_set_default_revcomp_of_n(uint32_t ordinal);

#ifdef __cplusplus
}
#endif
#endif /* READBOUNCER_AMD_H_ */
