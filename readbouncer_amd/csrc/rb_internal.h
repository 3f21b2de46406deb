// rb_internal.h -- private declarations shared by the translation units of libreadbouncer_amd.so
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/readbouncer_amd.h"
#include "../../include/readbouncer_amd_tuning.h"
#include "ibf_spec.h"

struct rb_ibf {
    rb_ibf_info geo{};
    uint64_t *words = nullptr;  // n_words payload words (host)
};

namespace rb {

void set_error(const std::string &msg);
int fail(int status, const std::string &msg);
void set_warning(const std::string &msg);
const std::string &last_warning();
bool geometry_from(uint64_t n_bins, uint64_t n_hash, uint64_t kmer_size, uint64_t n_bits, rb_ibf_info *g);
int open_ibf_stream(const char *path, FILE **fp_out, rb_ibf_info *geo);
void read_metadata(const uint64_t *tail, unsigned shift, uint64_t meta[4]);
void write_metadata(uint64_t *tail, unsigned shift, const uint64_t meta[4]);
bool metadata_plausible(const uint64_t meta[4], uint64_t n_bits);
bool calculate_ci(double r, uint8_t k, uint32_t readlen, double confidence, uint16_t *low, uint16_t *high);
uint16_t threshold_u16(uint64_t readlen, uint64_t kmer_size, double r, double confidence);
// rb_probe.hip: the read-peak probe on a bare block of device memory of the current device
int probe_read_peak_raw(const void *table, uint64_t table_bytes, uint32_t row_bytes, int nontemporal, uint32_t loads_in_flight,
                        double target_ms, double *gbps_out, double *ms_out);

}  // namespace rb

// device-to-device replication in two steps (rb_engine.hip), for rb_pool.cpp: start = allocate on `device` and queue the
// copy on a stream of its own, finish = wait for it.  used_peer = 1 when the pair is mapped for direct xGMI access.
extern "C" int rb_dibf_clone_start(const rb_dibf *src, int device, rb_dibf **out, void **stream, int *used_peer);
extern "C" int rb_dibf_clone_finish(void *stream);
