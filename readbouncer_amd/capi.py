"""ctypes binding of libreadbouncer_amd.so (the C ABI declared in include/readbouncer_amd.h).

This is plumbing for tests and bench.py: every call goes straight to the HIP library.  There is
no Python or CPU implementation of the hot path here -- if the shared library is missing the
import fails, and if no GPU is visible every compute call raises RBError(RB_ERR_NO_DEVICE).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libreadbouncer_amd.so")

(RB_OK, RB_ERR_NULL_FILTER, RB_ERR_SHORT_READ, RB_ERR_COUNT_KMER, RB_ERR_MISSING_FILE, RB_ERR_PARSE_IBF,
 RB_ERR_BAD_CHUNK, RB_ERR_STORE, RB_ERR_INVALID_ARG, RB_ERR_UNSUPPORTED, RB_ERR_NO_DEVICE, RB_ERR_HIP,
 RB_ERR_NOMEM) = range(13)
RB_MODE_CHECK_UNBLOCK, RB_MODE_CLASSIFY_CHUNK, RB_MODE_CLASSIFY_ANY = 0, 1, 2


class BatchDesc(C.Structure):
    _fields_ = [("d_seqs", C.c_void_p), ("d_offsets", C.c_void_p), ("d_lens", C.c_void_p), ("n_items", C.c_size_t),
                ("max_len", C.c_uint32), ("d_nmask", C.c_void_p), ("d_nmask_offsets", C.c_void_p),
                ("chunk_start", C.c_uint32), ("chunk_length", C.c_uint32), ("d_read_ids", C.c_void_p)]


class PlanInfo(C.Structure):
    _fields_ = [("kernel", C.c_char * 48), ("table_bytes", C.c_uint64), ("block_words", C.c_uint32), ("stride_words", C.c_uint32),
                ("merged_members", C.c_uint32), ("lanes_per_block_log2", C.c_uint32), ("words_per_lane", C.c_uint32),
                ("column_slices", C.c_uint32), ("counter_planes", C.c_uint32), ("nontemporal", C.c_uint32), ("split_waves", C.c_uint32),
                ("phased", C.c_uint32), ("phase_shape", C.c_uint32), ("phase_shape_name", C.c_char * 64),
                ("phase_slice_log2", C.c_uint32), ("phase_slices", C.c_uint32), ("phase_window_ticks", C.c_uint32),
                ("phase_rule_ticks", C.c_uint32), ("reserved0", C.c_uint32), ("phase_slice_bytes", C.c_uint64)]


class IbfCompare(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("file_bits", "rebuilt_bits", "new_bits", "payload_bits")]


class IbfInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("n_bins", "n_hash", "kmer_size", "n_bits", "bin_width", "n_blocks", "n_words")]


class RBError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = lib().rb_last_error().decode(errors="replace")
        super().__init__("%s: status %d (%s) %s" % (where, status, lib().rb_status_string(status).decode(), msg))


# every exported symbol with (restype, argtypes); tests check the .so exports exactly these
_vp, _u64, _u32, _sz, _dbl, _int = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t, C.c_double, C.c_int
_pp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "rb_status_string": (C.c_char_p, [_int]),
    "rb_last_error": (C.c_char_p, []),
    "rb_version": (C.c_char_p, []),
    "rb_device_count": (_int, []),
    "rb_ibf_open": (_int, [C.c_char_p, _pp]),
    "rb_ibf_create": (_int, [_u64, _u64, _u64, _u64, _pp]),
    "rb_ibf_store": (_int, [_vp, C.c_char_p]),
    "rb_ibf_get_info": (_int, [_vp, C.POINTER(IbfInfo)]),
    "rb_ibf_words": (_vp, [_vp]),
    "rb_ibf_close": (None, [_vp]),
    "rb_is_ibf_file": (_int, [C.c_char_p]),
    "rb_calculate_ci": (_int, [_dbl, C.c_uint8, _u32, _dbl, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]),
    "rb_threshold": (C.c_uint16, [_u64, _u64, _dbl, _dbl]),
    "rb_calculate_filter_size_bits": (_u64, [_u64, _u64, _u64, _dbl, _u64]),
    "rb_cut_out_nnns": (_sz, [C.c_char_p, _sz, C.c_char_p]),
    "rb_fragment_bounds": (_sz, [_u64, _u64, _u64, _u64, _vp, _vp, _sz]),
    "rb_dibf_create": (_int, [_int, _u64, _u64, _u64, _u64, _pp]),
    "rb_dibf_upload": (_int, [_int, _vp, _pp]),
    "rb_dibf_open": (_int, [_int, C.c_char_p, _pp]),
    "rb_dibf_download": (_int, [_vp, _pp]),
    "rb_dibf_get_info": (_int, [_vp, C.POINTER(IbfInfo)]),
    "rb_dibf_device_words": (_vp, [_vp]),
    "rb_dibf_touch": (_int, [_vp]),
    "rb_dibf_device_stride": (_u64, [_vp]),
    "rb_dibf_device": (_int, [_vp]),
    "rb_dibf_free": (None, [_vp]),
    "rb_dibf_resize_bins": (_int, [_vp, _u64, _pp]),
    "rb_dibf_fill_synth": (_int, [_vp, _u64]),
    "rb_dibf_insert": (_int, [_vp, _vp, _sz, _vp, _vp, _vp, _sz]),
    "rb_dibf_add_sequence": (_int, [_vp, _vp, _sz, _u64, _u64, _u64, C.POINTER(_u64)]),
    "rb_engine_create": (_int, [_int, _pp, _sz, _pp, _sz, _pp]),
    "rb_engine_destroy": (None, [_vp]),
    "rb_classify_batch": (_int, [_vp, _vp, _vp, _vp, _sz, _dbl, _dbl, _int, _vp, _vp, _vp, _vp]),
    "rb_classify_batch_ptrs": (_int, [_vp, _vp, _vp, _sz, _dbl, _dbl, _int, _vp, _vp, _vp, _vp]),
    "rb_classify_batch_device_ex": (_int, [_vp, C.POINTER(BatchDesc), _dbl, _dbl, _int, _vp, _vp, _vp, _vp, _vp]),
    "rb_pack_reads": (_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rb_host_alloc": (_int, [_sz, C.POINTER(_vp)]),
    "rb_host_free": (None, [_vp]),
    "rb_classify_batch_device": (_int, [_vp, _vp, _vp, _vp, _sz, _u32, _dbl, _dbl, _int, _vp, _vp, _vp, _vp, _vp]),
    "rb_engine_set_column_shard": (_int, [_vp, _int, _int]),
    "rb_decide_device": (_int, [_vp, _vp, _vp, _sz, _u32, _dbl, _dbl, _int, _vp, _vp, _vp, _vp]),
    "rb_decide_device_parts": (_int, [_vp, _vp, _u32, _u64, _vp, _sz, _u32, _dbl, _dbl, _int, _vp, _vp, _vp, _vp]),
    "rb_pool_create": (_int, [C.POINTER(_int), _sz, _pp, _sz, _pp, _sz, _pp]),
    "rb_pool_create_from_files": (_int, [C.POINTER(_int), _sz, C.POINTER(C.c_char_p), _sz, C.POINTER(C.c_char_p), _sz, _pp,
                                         C.POINTER(_dbl)]),
    "rb_pool_create_from_device": (_int, [C.POINTER(_int), _sz, _pp, _sz, _pp, _sz, _pp, C.POINTER(_dbl)]),
    "rb_pool_get_stats": (_int, [_vp, _sz, C.POINTER(_int), C.POINTER(_dbl), C.POINTER(_u64), C.POINTER(_u64), _int]),
    "rb_pool_destroy": (None, [_vp]),
    "rb_dibf_clone_to": (_int, [_vp, _int, _pp]),
    "rb_dibf_compare": (_int, [_vp, _vp, C.POINTER(IbfCompare)]),
    "rb_last_warning": (C.c_char_p, []),
    "rb_pool_size": (_sz, [_vp]),
    "rb_pool_set_min_split": (_int, [_vp, _sz]),
    "rb_pool_set_serialize": (_int, [_vp, _int]),
    "rb_pool_set_timing": (_int, [_vp, _int]),
    "rb_pool_kernel_time": (_int, [_vp, _sz, C.POINTER(_dbl), C.POINTER(_u64)]),
    "rb_pool_classify_batch": (_int, [_vp, _vp, _vp, _vp, _sz, _dbl, _dbl, _int, _vp, _vp, _vp, _vp]),
    "rb_live_create": (_int, [_vp, _dbl, _dbl, _u32, _pp]),
    "rb_live_destroy": (None, [_vp]),
    "rb_live_process": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "rb_live_pending": (_sz, [_vp]),
    "rb_live_forget": (_int, [_vp, C.c_char_p, _u32]),
    "rb_replay_arrivals": (_int, [_vp, _vp, _u32, _sz, _vp, _sz, _dbl, _dbl, _vp, _vp, _vp, _vp, _sz, C.POINTER(_sz),
                                  C.POINTER(_dbl)]),
    "rb_live_replay_arrivals": (_int, [_vp, _vp, _vp, _u32, _sz, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(_sz),
                                       C.POINTER(_dbl)]),
    "rb_dibf_clone_to_ex": (_int, [_vp, _int, _pp, C.POINTER(_int), C.POINTER(_dbl)]),
    "rb_engine_set_revcomp_of_n": (_int, [_vp, _u32]),
    "rb_set_default_revcomp_of_n": (_int, [_u32]),
    "rb_set_placement_tries": (_int, [_int]),
    "rb_dibf_placement": (_int, [_vp, C.POINTER(_u32), C.POINTER(_dbl), C.POINTER(_dbl)]),
    "rb_dibf_placement_cost": (_int, [_vp, C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_u64), C.POINTER(_u32)]),
    "rb_engine_set_merge": (_int, [_vp, _int]),
    "rb_engine_merge_info": (_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "rb_engine_set_split_threshold": (_int, [_vp, _u32]),
    "rb_engine_set_overlap": (_int, [_vp, _int]),
    "rb_engine_set_split_parts": (_int, [_vp, _u32, _u32]),
    "rb_engine_set_fold_decide": (_int, [_vp, _int]),
    "rb_engine_set_completion_word": (_int, [_vp, _int]),
    "rb_engine_set_nt_threshold": (_int, [_vp, _u64]),
    "rb_engine_set_host_slice_bytes": (_int, [_vp, _u64]),
    "rb_engine_set_serial_table_bytes": (_int, [_vp, _u64]),
    "rb_engine_set_phase_slices": (_int, [_vp, _u32, _u32]),
    "rb_engine_set_phase_equal_slices": (_int, [_vp, _u32]),
    "rb_engine_set_reads_per_wave": (_int, [_vp, _u32]),
    "rb_engine_set_phase_xcd_skew": (_int, [_vp, _u32]),
    "rb_engine_set_early_decision": (_int, [_vp, _int]),
    "rb_engine_set_phased": (_int, [_vp, _u64, _u64, _u32, _u32, _u32]),
    "rb_engine_set_timing": (_int, [_vp, _int]),
    "rb_engine_kernel_time": (_int, [_vp, C.POINTER(_dbl), C.POINTER(_u64)]),
    "rb_engine_plan": (_int, [_vp, _sz, _sz, _u32, C.POINTER(PlanInfo)]),
    "rb_engine_calibrate": (_int, [_vp, _sz, _u32, _dbl, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rb_dibf_probe_read_peak": (_int, [_vp, _u64, _u32, _int, _u32, _dbl, C.POINTER(_dbl), C.POINTER(_dbl)]),
}

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (same SONAME as
    /opt/rocm's, different file).  If our library pulled in the system copy and torch loaded its own
    later, two runtimes would fight over the device; and loading torch's libraries after a runtime has
    already been initialised can make their code-object registration take minutes.  So when torch is
    installed it is imported BEFORE libreadbouncer_amd.so is loaded: its libraries register lazily and
    our library binds to the already loaded runtime by SONAME.  Without torch the system runtime is
    used.  torch stays plumbing: nothing else of it is touched here."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    except Exception:
        pass


def lib():
    global _lib
    if _lib is None:
        # RB_AMD_LIBRARY: another build of the SAME library (tests load the -DRB_TESTING build in a child process); never a
        # different implementation -- the symbol table below is checked either way
        path = os.environ.get("RB_AMD_LIBRARY") or LIB_PATH
        if not os.path.exists(path):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no fallback implementation" % path)
        _preload_hip_runtime()
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(st, where):
    if st != RB_OK:
        raise RBError(st, where)


def _ptr(a):
    """numpy array / int / None -> void* value"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    return int(a)


def _info(handle, fn):
    i = IbfInfo()
    _check(fn(handle, C.byref(i)), "get_info")
    return {k: getattr(i, k) for k, _ in IbfInfo._fields_}


class HostIBF:
    """rb_ibf: host image of a .ibf file (IBF::load_filter / TIbf ctor / seqan::store)."""

    def __init__(self, handle):
        self.h = handle
        self.info = _info(self.h, lib().rb_ibf_get_info)

    @classmethod
    def open(cls, path):
        h = C.c_void_p()
        _check(lib().rb_ibf_open(os.fsencode(path), C.byref(h)), "rb_ibf_open")
        return cls(h)

    @classmethod
    def create(cls, n_bins, n_hash, kmer_size, n_bits):
        h = C.c_void_p()
        _check(lib().rb_ibf_create(n_bins, n_hash, kmer_size, n_bits, C.byref(h)), "rb_ibf_create")
        return cls(h)

    def words(self):
        p = lib().rb_ibf_words(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(self.info["n_words"],))

    def store(self, path):
        _check(lib().rb_ibf_store(self.h, os.fsencode(path)), "rb_ibf_store")

    def close(self):
        if self.h:
            lib().rb_ibf_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceIBF:
    """rb_dibf: an IBF resident in HBM."""

    def __init__(self, handle):
        self.h = handle
        self.info = _info(self.h, lib().rb_dibf_get_info)

    @classmethod
    def create(cls, device, n_bins, n_hash, kmer_size, n_bits):
        h = C.c_void_p()
        _check(lib().rb_dibf_create(device, n_bins, n_hash, kmer_size, n_bits, C.byref(h)), "rb_dibf_create")
        return cls(h)

    @classmethod
    def upload(cls, device, host):
        h = C.c_void_p()
        _check(lib().rb_dibf_upload(device, host.h, C.byref(h)), "rb_dibf_upload")
        return cls(h)

    @classmethod
    def open(cls, device, path):
        h = C.c_void_p()
        _check(lib().rb_dibf_open(device, os.fsencode(path), C.byref(h)), "rb_dibf_open")
        return cls(h)

    def download(self):
        h = C.c_void_p()
        _check(lib().rb_dibf_download(self.h, C.byref(h)), "rb_dibf_download")
        return HostIBF(h)

    def clone_to(self, device):
        """device-to-device replica (xGMI between peers) in the HBM layout"""
        h = C.c_void_p()
        _check(lib().rb_dibf_clone_to(self.h, device, C.byref(h)), "rb_dibf_clone_to")
        return DeviceIBF(h)

    def clone_to_ex(self, device):
        """-> (replica, used_peer, seconds): how the copy travelled (peer mapping = xGMI between two GPUs) and how long"""
        h, peer, secs = C.c_void_p(), C.c_int(0), C.c_double(0)
        _check(lib().rb_dibf_clone_to_ex(self.h, device, C.byref(h), C.byref(peer), C.byref(secs)), "rb_dibf_clone_to_ex")
        return DeviceIBF(h), bool(peer.value), secs.value

    def compare(self, rebuilt):
        """bit statistics against a filter of the same geometry re-inserted from the reference sequences"""
        c = IbfCompare()
        _check(lib().rb_dibf_compare(self.h, rebuilt.h, C.byref(c)), "rb_dibf_compare")
        return {k: getattr(c, k) for k, _ in IbfCompare._fields_}

    def device_words(self):
        return lib().rb_dibf_device_words(self.h)

    def device_stride(self):
        return lib().rb_dibf_device_stride(self.h)

    def probe_read_peak(self, row_bytes, nontemporal, loads_in_flight=12, table_bytes=0, target_ms=200.0):
        """measurement aid: GB/s of random whole-row gathers from this filter's table with no compute attached -> (GB/s, ms)"""
        g, ms = _dbl(0.0), _dbl(0.0)
        _check(lib().rb_dibf_probe_read_peak(self.h, table_bytes, row_bytes, int(bool(nontemporal)), loads_in_flight, target_ms,
                                             C.byref(g), C.byref(ms)), "rb_dibf_probe_read_peak")
        return g.value, ms.value

    def placement(self):
        """-> (allocations probed for this table: 0 = not placed by trial, GB/s of the kept one, GB/s of the slowest one)"""
        t, g, w = _u32(0), _dbl(0.0), _dbl(0.0)
        _check(lib().rb_dibf_placement(self.h, C.byref(t), C.byref(g), C.byref(w)), "rb_dibf_placement")
        return t.value, g.value, w.value

    def placement_cost(self):
        """-> dict: seconds of the trial, seconds waited after it, most HBM its candidates held, why a large table was not tried (0 / 1 / 2)"""
        ts, ss, pk, sk = _dbl(0.0), _dbl(0.0), _u64(0), _u32(0)
        _check(lib().rb_dibf_placement_cost(self.h, C.byref(ts), C.byref(ss), C.byref(pk), C.byref(sk)), "rb_dibf_placement_cost")
        return {"trial_s": ts.value, "settle_s": ss.value, "peak_bytes": pk.value, "skipped": sk.value}

    def resize_bins(self, new_bins):
        h = C.c_void_p()
        _check(lib().rb_dibf_resize_bins(self.h, new_bins, C.byref(h)), "rb_dibf_resize_bins")
        return DeviceIBF(h)

    def fill_synth(self, seed):
        _check(lib().rb_dibf_fill_synth(self.h, seed), "rb_dibf_fill_synth")

    def insert(self, seq, starts, ends, bins):
        if isinstance(seq, str):
            seq = seq.encode()
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        ends = np.ascontiguousarray(ends, dtype=np.uint64)
        bins = np.ascontiguousarray(bins, dtype=np.uint64)
        buf = np.frombuffer(seq, dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
        _check(lib().rb_dibf_insert(self.h, _ptr(np.ascontiguousarray(buf)), len(buf), _ptr(starts), _ptr(ends),
                                    _ptr(bins), len(starts)), "rb_dibf_insert")

    def add_sequence(self, seq, fragment_length, first_bin=0, overlap_length=1500):
        if isinstance(seq, str):
            seq = seq.encode()
        buf = np.ascontiguousarray(np.frombuffer(seq, dtype=np.uint8))
        nxt = C.c_uint64(0)
        _check(lib().rb_dibf_add_sequence(self.h, _ptr(buf), len(buf), fragment_length, overlap_length, first_bin,
                                          C.byref(nxt)), "rb_dibf_add_sequence")
        return nxt.value

    def free(self):
        if self.h:
            lib().rb_dibf_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _handle_array(filters):
    arr = (C.c_void_p * max(1, len(filters)))()
    for i, f in enumerate(filters):
        arr[i] = f.h
    return C.cast(arr, _pp)


class Engine:
    """rb_engine: per-GPU classifier over borrowed deplete/target filters."""

    def __init__(self, device, deplete, target):
        self._keep = (list(deplete), list(target))
        self.nd, self.nt = len(deplete), len(target)
        h = C.c_void_p()
        _check(lib().rb_engine_create(device, _handle_array(deplete), len(deplete), _handle_array(target), len(target),
                                      C.byref(h)), "rb_engine_create")
        self.h = h

    def classify(self, seqs, offsets, lens, error_rate=0.1, significance=0.95, mode=RB_MODE_CHECK_UNBLOCK):
        """host buffers in, host numpy arrays out: (maxcount[n, nf], best_target[n], decision[n], status[n])"""
        n = len(lens)
        nf = self.nd + self.nt
        maxcount = np.zeros((n, nf), dtype=np.uint16)
        best = np.full(n, -1, dtype=np.int32)
        decision = np.zeros(n, dtype=np.uint8)
        status = np.zeros(n, dtype=np.uint8)
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        _check(lib().rb_classify_batch(self.h, _ptr(seqs), _ptr(offsets), _ptr(lens), n, error_rate, significance, mode,
                                       _ptr(maxcount), _ptr(best), _ptr(decision), _ptr(status)), "rb_classify_batch")
        return maxcount, best, decision, status

    def decide(self, seqs, offsets, lens, error_rate=0.1, significance=0.95, mode=RB_MODE_CHECK_UNBLOCK):
        """host buffers in, (decision[n], status[n]) out; the raw maxima and best_target are NOT asked for (NULL out pointers) -- the form
        of call the opt-in early-decision mode applies to (set_early_decision)"""
        n = len(lens)
        decision = np.zeros(n, dtype=np.uint8)
        status = np.zeros(n, dtype=np.uint8)
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        _check(lib().rb_classify_batch(self.h, _ptr(seqs), _ptr(offsets), _ptr(lens), n, error_rate, significance, mode,
                                       None, None, _ptr(decision), _ptr(status)), "rb_classify_batch")
        return decision, status

    def classify_reads(self, reads, error_rate=0.1, significance=0.95, mode=RB_MODE_CHECK_UNBLOCK):
        """list of bytes objects (one buffer per read) through rb_classify_batch_ptrs"""
        n = len(reads)
        nf = self.nd + self.nt
        keep = [r.encode() if isinstance(r, str) else r for r in reads]
        ptrs = (C.c_char_p * max(1, n))(*keep)
        lens = np.array([len(r) for r in keep], dtype=np.uint32)
        maxcount = np.zeros((n, nf), dtype=np.uint16)
        best = np.full(n, -1, dtype=np.int32)
        decision = np.zeros(n, dtype=np.uint8)
        status = np.zeros(n, dtype=np.uint8)
        _check(lib().rb_classify_batch_ptrs(self.h, C.cast(ptrs, C.c_void_p), _ptr(lens), n, error_rate, significance, mode,
                                            _ptr(maxcount), _ptr(best), _ptr(decision), _ptr(status)),
               "rb_classify_batch_ptrs")
        return maxcount, best, decision, status

    def classify_device(self, d_seqs, d_offsets, d_lens, n_reads, max_len, error_rate=0.1, significance=0.95,
                        mode=RB_MODE_CHECK_UNBLOCK, d_maxcount=None, d_best=None, d_decision=None, d_status=None,
                        stream=None):
        """raw device pointers (ints, e.g. torch.Tensor.data_ptr())"""
        _check(lib().rb_classify_batch_device(self.h, d_seqs, d_offsets, d_lens, n_reads, max_len, error_rate,
                                              significance, mode, d_maxcount, d_best, d_decision, d_status, stream),
               "rb_classify_batch_device")

    def classify_device_ex(self, d_seqs, d_offsets, d_lens, n_items, max_len, d_nmask=None, d_nmask_offsets=None,
                           chunk_start=0, chunk_length=0, d_read_ids=None, error_rate=0.1, significance=0.95,
                           mode=RB_MODE_CHECK_UNBLOCK, d_maxcount=None, d_best=None, d_decision=None, d_status=None,
                           stream=None):
        desc = BatchDesc(d_seqs, d_offsets, d_lens, n_items, max_len, d_nmask, d_nmask_offsets, chunk_start, chunk_length,
                         d_read_ids)
        _check(lib().rb_classify_batch_device_ex(self.h, C.byref(desc), error_rate, significance, mode, d_maxcount, d_best,
                                                 d_decision, d_status, stream), "rb_classify_batch_device_ex")

    def decide_device(self, d_maxcount, d_lens, n_reads, max_len, error_rate=0.1, significance=0.95,
                      mode=RB_MODE_CHECK_UNBLOCK, d_best=None, d_decision=None, d_status=None, stream=None):
        _check(lib().rb_decide_device(self.h, d_maxcount, d_lens, n_reads, max_len, error_rate, significance, mode,
                                      d_best, d_decision, d_status, stream), "rb_decide_device")

    def decide_device_parts(self, d_maxcount, n_parts, part_stride, d_lens, n_reads, max_len, error_rate=0.1,
                            significance=0.95, mode=RB_MODE_CHECK_UNBLOCK, d_best=None, d_decision=None, d_status=None,
                            stream=None):
        """decision over n_parts all-gathered partial maxcount tables (bin-sharded ranks), part_stride elements apart"""
        _check(lib().rb_decide_device_parts(self.h, d_maxcount, n_parts, part_stride, d_lens, n_reads, max_len, error_rate,
                                            significance, mode, d_best, d_decision, d_status, stream),
               "rb_decide_device_parts")

    def replay_arrivals(self, seqs, read_len, arrival_s, max_batch=16384, error_rate=0.1, significance=0.95):
        """work-conserving replay in C++ -> (decision[n], latency_s[n], call_reads[calls], call_service_s[calls], elapsed_s)"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        arrival_s = np.ascontiguousarray(arrival_s, dtype=np.float64)
        n = len(arrival_s)
        assert len(seqs) >= n * read_len
        dec = np.zeros(n, dtype=np.uint8)
        lat = np.zeros(n, dtype=np.float64)
        cr = np.zeros(n, dtype=np.uint32)
        cs = np.zeros(n, dtype=np.float64)
        calls, el = C.c_size_t(0), C.c_double(0)
        _check(lib().rb_replay_arrivals(self.h, _ptr(seqs), read_len, n, _ptr(arrival_s), max_batch, error_rate, significance,
                                        _ptr(dec), _ptr(lat), _ptr(cr), _ptr(cs), n, C.byref(calls), C.byref(el)),
               "rb_replay_arrivals")
        return dec, lat, cr[:calls.value], cs[:calls.value], el.value

    def set_column_shard(self, rank, world):
        _check(lib().rb_engine_set_column_shard(self.h, rank, world), "rb_engine_set_column_shard")

    def set_revcomp_of_n(self, ordinal):
        """3 (default): the reverse strand sees T where the read has N (ModComplementDna on a Dna5String); 4: N stays N"""
        _check(lib().rb_engine_set_revcomp_of_n(self.h, ordinal), "rb_engine_set_revcomp_of_n")

    def set_merge(self, mode):
        """filters of one hash geometry in one merged table: 0 never, 1 when it pays (default), 2 whenever two qualify"""
        _check(lib().rb_engine_set_merge(self.h, mode), "rb_engine_set_merge")

    def merge_info(self):
        """(merged tables, filters they serve, HBM bytes of the copies)"""
        t, f, b = C.c_uint32(), C.c_uint32(), C.c_uint64()
        _check(lib().rb_engine_merge_info(self.h, C.byref(t), C.byref(f), C.byref(b)), "rb_engine_merge_info")
        return t.value, f.value, b.value

    def plan(self, filter_index, n_reads, max_len):
        """what the engine would launch for that filter on such a batch -> dict (rb_plan_info)"""
        p = PlanInfo()
        _check(lib().rb_engine_plan(self.h, filter_index, n_reads, max_len, C.byref(p)), "rb_engine_plan")
        d = {k: getattr(p, k) for k, _ in PlanInfo._fields_}
        d["kernel"], d["phase_shape_name"] = p.kernel.decode(), p.phase_shape_name.decode()
        return d

    def calibrate(self, n_reads=262144, read_len=250, max_ms=0.0):
        """fit the phased windows to this device -> (phased tables found, tables whose window changed)"""
        nt, nc = C.c_uint32(0), C.c_uint32(0)
        _check(lib().rb_engine_calibrate(self.h, n_reads, read_len, max_ms, C.byref(nt), C.byref(nc)), "rb_engine_calibrate")
        return nt.value, nc.value

    def set_split_threshold(self, max_reads):
        _check(lib().rb_engine_set_split_threshold(self.h, max_reads), "rb_engine_set_split_threshold")

    def set_overlap(self, on):
        _check(lib().rb_engine_set_overlap(self.h, int(on)), "rb_engine_set_overlap")

    def set_split_parts(self, max_parts, max_shares=4):
        _check(lib().rb_engine_set_split_parts(self.h, max_parts, max_shares), "rb_engine_set_split_parts")

    def set_fold_decide(self, on):
        _check(lib().rb_engine_set_fold_decide(self.h, int(on)), "rb_engine_set_fold_decide")

    def set_completion_word(self, on):
        _check(lib().rb_engine_set_completion_word(self.h, int(on)), "rb_engine_set_completion_word")

    def set_nt_threshold(self, table_bytes):
        _check(lib().rb_engine_set_nt_threshold(self.h, table_bytes), "rb_engine_set_nt_threshold")

    def set_serial_table_bytes(self, table_bytes):
        _check(lib().rb_engine_set_serial_table_bytes(self.h, table_bytes), "rb_engine_set_serial_table_bytes")

    def set_phased(self, min_table_bytes=5 << 18, max_table_bytes=128 << 20, base_ticks=0, ticks_per_mib=0, min_reads=2049):
        _check(lib().rb_engine_set_phased(self.h, min_table_bytes, max_table_bytes, base_ticks, ticks_per_mib, min_reads),
               "rb_engine_set_phased")

    def set_phase_slices(self, slice_log2=0, max_slices=32):
        """slices of 2^slice_log2 bytes (0: built-in rule; 1-5: as small as max_slices allows), at most max_slices"""
        _check(lib().rb_engine_set_phase_slices(self.h, slice_log2, max_slices), "rb_engine_set_phase_slices")

    def set_phase_equal_slices(self, n_slices=0):
        """n equal-length slices for the phased form (0 = the built-in rule); ignored while set_phase_slices names a slice size"""
        _check(lib().rb_engine_set_phase_equal_slices(self.h, n_slices), "rb_engine_set_phase_equal_slices")

    def set_reads_per_wave(self, reads):
        """two-word phased tables, reads of up to 256 k-mers: reads a wave carries through a pass of the windows (0: the one-read build)"""
        _check(lib().rb_engine_set_reads_per_wave(self.h, reads), "rb_engine_set_reads_per_wave")

    def set_early_decision(self, on):
        """opt-in: check_unblock calls without raw maxima stop counting a read once a bin has reached the larger of its two thresholds"""
        _check(lib().rb_engine_set_early_decision(self.h, int(on)), "rb_engine_set_early_decision")

    def set_phase_xcd_skew(self, mode):
        """bit 0: every XCD on a different slice at any time; bit 1: the XCDs' windows start an eighth of a window apart"""
        _check(lib().rb_engine_set_phase_xcd_skew(self.h, mode), "rb_engine_set_phase_xcd_skew")

    def set_host_slice_bytes(self, slice_bytes):
        _check(lib().rb_engine_set_host_slice_bytes(self.h, slice_bytes), "rb_engine_set_host_slice_bytes")

    def set_timing(self, on):
        _check(lib().rb_engine_set_timing(self.h, int(on)), "rb_engine_set_timing")

    def kernel_time(self):
        """(total_ms, n_calls) of the count kernels since the last query; waits for them"""
        ms, n = C.c_double(0), C.c_uint64(0)
        _check(lib().rb_engine_kernel_time(self.h, C.byref(ms), C.byref(n)), "rb_engine_kernel_time")
        return ms.value, n.value

    def destroy(self):
        if self.h:
            lib().rb_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Pool:
    """rb_pool: one engine per device entry, filters replicated from host images, read-sharded batches."""

    def __init__(self, devices, deplete_images, target_images):
        self.nd, self.nt = len(deplete_images), len(target_images)
        self._keep = (list(deplete_images), list(target_images))
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(lib().rb_pool_create(devs, len(devices), _handle_array(deplete_images), self.nd,
                                    _handle_array(target_images), self.nt, C.byref(h)), "rb_pool_create")
        self.h = h

    @classmethod
    def from_files(cls, devices, deplete_paths, target_paths):
        """filters streamed once into devices[0] and replicated device to device; no host images"""
        self = cls.__new__(cls)
        self.nd, self.nt = len(deplete_paths), len(target_paths)
        self._keep = ()
        devs = (C.c_int * len(devices))(*devices)
        dp = (C.c_char_p * max(1, self.nd))(*[os.fsencode(x) for x in deplete_paths])
        tp = (C.c_char_p * max(1, self.nt))(*[os.fsencode(x) for x in target_paths])
        h = C.c_void_p()
        secs = C.c_double(0)
        _check(lib().rb_pool_create_from_files(devs, len(devices), dp, self.nd, tp, self.nt, C.byref(h), C.byref(secs)),
               "rb_pool_create_from_files")
        self.h = h
        self.replication_seconds = secs.value
        return self

    @classmethod
    def from_device(cls, devices, deplete, target):
        """replicas of filters that are already resident (DeviceIBF), copied device to device onto every entry of `devices`"""
        self = cls.__new__(cls)
        self.nd, self.nt = len(deplete), len(target)
        self._keep = (list(deplete), list(target))
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        secs = C.c_double(0)
        _check(lib().rb_pool_create_from_device(devs, len(devices), _handle_array(deplete), self.nd, _handle_array(target), self.nt,
                                                C.byref(h), C.byref(secs)), "rb_pool_create_from_device")
        self.h = h
        self.replication_seconds = secs.value
        return self

    def stats(self, reset=False):
        """per worker: (device, busy seconds inside its engine, reads served, parts of calls served)"""
        n = self.size()
        dev, busy, reads, calls = (C.c_int * n)(), (C.c_double * n)(), (C.c_uint64 * n)(), (C.c_uint64 * n)()
        _check(lib().rb_pool_get_stats(self.h, n, dev, busy, reads, calls, int(reset)), "rb_pool_get_stats")
        return [(dev[i], busy[i], reads[i], calls[i]) for i in range(n)]

    def set_timing(self, on):
        _check(lib().rb_pool_set_timing(self.h, int(on)), "rb_pool_set_timing")

    def kernel_time(self):
        """per worker: (summed K1 milliseconds since the last collection, bracketed launches) -- rb_engine_kernel_time of each engine"""
        n = self.size()
        ms, calls = (C.c_double * n)(), (C.c_uint64 * n)()
        _check(lib().rb_pool_kernel_time(self.h, n, ms, calls), "rb_pool_kernel_time")
        return [(ms[i], calls[i]) for i in range(n)]

    def classify_into(self, seqs_ptr, offsets_ptr, lens_ptr, n, maxcount_ptr, best_ptr, decision_ptr, status_ptr, error_rate=0.1,
                      significance=0.95, mode=RB_MODE_CHECK_UNBLOCK):
        """the C call on caller-owned buffers (e.g. page-locked ones from host_alloc), no numpy copies around it"""
        _check(lib().rb_pool_classify_batch(self.h, seqs_ptr, offsets_ptr, lens_ptr, n, error_rate, significance, mode, maxcount_ptr,
                                            best_ptr, decision_ptr, status_ptr), "rb_pool_classify_batch")

    def size(self):
        return lib().rb_pool_size(self.h)

    def set_min_split(self, reads_per_device):
        _check(lib().rb_pool_set_min_split(self.h, reads_per_device), "rb_pool_set_min_split")

    def set_serialize(self, on):
        _check(lib().rb_pool_set_serialize(self.h, int(on)), "rb_pool_set_serialize")

    def classify(self, seqs, offsets, lens, error_rate=0.1, significance=0.95, mode=RB_MODE_CHECK_UNBLOCK):
        n = len(lens)
        nf = self.nd + self.nt
        maxcount = np.zeros((n, nf), dtype=np.uint16)
        best = np.full(n, -1, dtype=np.int32)
        decision = np.zeros(n, dtype=np.uint8)
        status = np.zeros(n, dtype=np.uint8)
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        _check(lib().rb_pool_classify_batch(self.h, _ptr(seqs), _ptr(offsets), _ptr(lens), n, error_rate, significance,
                                            mode, _ptr(maxcount), _ptr(best), _ptr(decision), _ptr(status)),
               "rb_pool_classify_batch")
        return maxcount, best, decision, status

    def destroy(self):
        if self.h:
            lib().rb_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Live:
    """rb_live: micro-batch form of classify_live_reads (once_seen bookkeeping + actions)."""

    def __init__(self, engine, error_rate=0.1, significance=0.95, max_undecided_len=1500):
        self._engine = engine
        h = C.c_void_p()
        _check(lib().rb_live_create(engine.h, error_rate, significance, max_undecided_len, C.byref(h)), "rb_live_create")
        self.h = h

    def process(self, ids, seqs):
        """ids, seqs: lists of bytes/str -> (action[n] uint8, status[n] uint8, classified_len[n] uint32)"""
        n = len(ids)
        ids = [i.encode() if isinstance(i, str) else i for i in ids]
        seqs = [s.encode() if isinstance(s, str) else s for s in seqs]

        def pack(items):
            lens = np.array([len(x) for x in items], dtype=np.uint32)
            offs = np.zeros(n, dtype=np.uint64)
            if n:
                offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
            buf = np.frombuffer(b"".join(items) or b"N", dtype=np.uint8).copy()
            return buf, offs, lens
        ib, io, il = pack(ids)
        sb, so, sl = pack(seqs)
        action = np.zeros(n, dtype=np.uint8)
        status = np.zeros(n, dtype=np.uint8)
        clen = np.zeros(n, dtype=np.uint32)
        _check(lib().rb_live_process(self.h, _ptr(ib), _ptr(io), _ptr(il), _ptr(sb), _ptr(so), _ptr(sl), n, _ptr(action),
                                     _ptr(status), _ptr(clen)), "rb_live_process")
        return action, status, clen

    def replay_arrivals(self, read_ids, seqs, read_len, arrival_s, max_batch=16384):
        """work-conserving replay through the live step -> (action[n], latency_s[n], classified_len[n], call_reads, call_service_s,
        elapsed_s)"""
        read_ids = np.ascontiguousarray(read_ids, dtype=np.uint32)
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        arrival_s = np.ascontiguousarray(arrival_s, dtype=np.float64)
        n = len(arrival_s)
        assert len(seqs) >= n * read_len and len(read_ids) == n
        act = np.zeros(n, dtype=np.uint8)
        lat = np.zeros(n, dtype=np.float64)
        clen = np.zeros(n, dtype=np.uint32)
        cr = np.zeros(n, dtype=np.uint32)
        cs = np.zeros(n, dtype=np.float64)
        calls, el = C.c_size_t(0), C.c_double(0)
        _check(lib().rb_live_replay_arrivals(self.h, _ptr(read_ids), _ptr(seqs), read_len, n, _ptr(arrival_s), max_batch,
                                             _ptr(act), _ptr(lat), _ptr(clen), _ptr(cr), _ptr(cs), n, C.byref(calls),
                                             C.byref(el)), "rb_live_replay_arrivals")
        return act, lat, clen, cr[:calls.value], cs[:calls.value], el.value

    def pending(self):
        return lib().rb_live_pending(self.h)

    def forget(self, read_id):
        if isinstance(read_id, str):
            read_id = read_id.encode()
        _check(lib().rb_live_forget(self.h, read_id, len(read_id)), "rb_live_forget")

    def destroy(self):
        if self.h:
            lib().rb_live_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class HostBlock:
    """page-locked host memory from rb_host_alloc, seen as a numpy array of `dtype`"""

    def __init__(self, count, dtype=np.uint8):
        self.ptr = C.c_void_p()
        nbytes = int(count) * np.dtype(dtype).itemsize
        _check(lib().rb_host_alloc(max(1, nbytes), C.byref(self.ptr)), "rb_host_alloc")
        buf = (C.c_uint8 * max(1, nbytes)).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(count))

    def free(self):
        if self.ptr:
            self.array = None
            lib().rb_host_free(self.ptr)
            self.ptr = None


def set_placement_tries(tries):
    """process-wide: candidates a table of >= 1 GiB is allocated and probed as (default 5; 0 / 1 = off)"""
    _check(lib().rb_set_placement_tries(int(tries)), "rb_set_placement_tries")


def pack_reads(seqs, offsets, lens):
    """ASCII batch -> (packed uint8, packed_offsets uint64, nmask uint8, nmask_offsets uint64)"""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    n = len(lens)
    po_, no_ = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    pb, nb = C.c_uint64(0), C.c_uint64(0)
    _check(lib().rb_pack_reads(_ptr(seqs), _ptr(offsets), _ptr(lens), n, None, _ptr(po_), None, _ptr(no_), C.byref(pb),
                               C.byref(nb)), "rb_pack_reads")
    packed = np.zeros(max(1, pb.value), dtype=np.uint8)
    nmask = np.zeros(max(1, nb.value), dtype=np.uint8)
    _check(lib().rb_pack_reads(_ptr(seqs), _ptr(offsets), _ptr(lens), n, _ptr(packed), _ptr(po_), _ptr(nmask), _ptr(no_),
                               None, None), "rb_pack_reads")
    return packed, po_, nmask, no_


def threshold(readlen, k, error_rate=0.1, significance=0.95):
    return lib().rb_threshold(readlen, k, error_rate, significance)


def calculate_ci(error_rate, k, readlen, significance):
    lo, hi = C.c_uint16(), C.c_uint16()
    st = lib().rb_calculate_ci(error_rate, k, readlen, significance, C.byref(lo), C.byref(hi))
    return st, lo.value, hi.value


def calculate_filter_size_bits(fragment_length, k, h, max_fp, n_bins):
    return lib().rb_calculate_filter_size_bits(fragment_length, k, h, max_fp, n_bins)


def cut_out_nnns(seq):
    if isinstance(seq, str):
        seq = seq.encode()
    out = C.create_string_buffer(len(seq) + 1)
    n = lib().rb_cut_out_nnns(seq, len(seq), out)
    return out.raw[:n].decode()


def fragment_bounds(length, fragment_length, k, overlap=1500):
    n = lib().rb_fragment_bounds(length, fragment_length, k, overlap, None, None, 0)
    s = np.zeros(n, dtype=np.uint64)
    e = np.zeros(n, dtype=np.uint64)
    lib().rb_fragment_bounds(length, fragment_length, k, overlap, _ptr(s), _ptr(e), n)
    return s, e


def is_ibf_file(path):
    return bool(lib().rb_is_ibf_file(os.fsencode(path)))


def last_warning():
    return lib().rb_last_warning().decode(errors="replace")


def device_count():
    return lib().rb_device_count()
