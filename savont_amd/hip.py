"""ctypes binding of the C-ABI device layer (include/savont_hip.h -> savont_amd/libsavont_hip.so).

This is plumbing for the Python harness (tests, bench.py).  It fails loudly when the HIP library
is missing or when no gfx950 device is present: there is no CPU fallback and nothing here touches
oracle/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsavont_hip.so")

SVT_OK = 0
SVT_ERR_OVERFLOW = -4
SVT_ERR_NODEVICE = -5
SVT_ERR_TOOWIDE = -6
SVT_ERR_EXCHANGE = -7
COMM_ID_BYTES = 128
VIEW_ALL, VIEW_FILTERED = 0, 1
LIST_COMPATIBLE, LIST_OVERLAP = 0, 1
LSH_TABLES = 20

# every symbol include/savont_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "svt_version", "svt_device_count", "svt_create", "svt_destroy", "svt_last_error", "svt_set_option", "svt_get_option", "svt_fork", "svt_fork_refresh",
    "svt_profile_enable", "svt_profile_reset", "svt_profile_count", "svt_profile_get",
    "svt_hbm_copy_peak", "svt_batch_upload", "svt_batch_free", "svt_batch_repack", "svt_batch_slice",
    "svt_count_partial_device", "svt_count_export_device", "svt_count_merge_begin", "svt_count_merge_device", "svt_batch_size", "svt_batch_fetch_packed",
    "svt_split_kmers_emit", "svt_count_split_kmers", "svt_count_fetch", "svt_count_candidates_sizes", "svt_count_candidates_fetch", "svt_count_partial",
    "svt_count_export", "svt_count_merge", "svt_count_finalize", "svt_set_snpmers",
    "svt_extract_seeds", "svt_seeds_sizes", "svt_seeds_fetch", "svt_twin_order", "svt_twin_gather", "svt_lsh_candidates", "svt_minimizer_shared_counts",
    "svt_snpmer_words", "svt_snpmer_site_order", "svt_snpmer_bits_fetch", "svt_bitset_upload", "svt_bitset_free",
    "svt_snpmer_compat_lists", "svt_snpmer_consensus", "svt_snpmer_best_column", "svt_align_nm", "svt_align_nm_affine", "svt_align_nm_affine_near", "svt_set_shard", "svt_shard_comm_id", "svt_set_shard_comm", "svt_shard_abort", "svt_count_shard_merge", "svt_shard_info", "svt_shard_pause", "svt_shard_allgather_u64", "svt_shard_allgatherv", "svt_host_pin", "svt_host_unpin", "svt_qualbin_mean", "svt_batch_set_tags", "svt_pileup_hp_median", "svt_align_pileup",
    "svt_pileup_create", "svt_pileup_free", "svt_pileup_cells", "svt_pileup_columns", "svt_pileup_fetch", "svt_pileup_stats", "svt_pileup_loglik", "svt_snpmer_compat_lists_seg", "svt_snpmer_compat_rows_seg", "svt_poa_graphs", "svt_poa_graphs_submit", "svt_poa_graphs_submit_reads", "svt_poa_consensus_fetch", "svt_poa_graphs_wait", "svt_poa_graphs_fetch", "svt_read_asv_ties",
]


class SeedsOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "mini_off", "mini_pos", "mini_kmer", "mini_flags", "snp_off", "snp_pos", "snp_kmer", "snp_flags",
        "est_id", "est_valid", "lsh", "lsh_valid", "n_unique", "n_solid", "qualbin_off", "qualbins", "status")]


class SavontHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load libsavont_hip.so; raises if the extension has not been built (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SavontHipError("HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so (same soname as /opt/rocm's) and its libraries ask for it by file name.
    # If this library loaded the system copy first, torch would load a second runtime next to it and find no GPU ("No HIP GPUs are available");
    # loaded after torch, this library binds to the copy torch brought (soname match).  So torch goes first whenever it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.svt_version.restype = C.c_int
    L.svt_device_count.restype = C.c_int
    L.svt_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.svt_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.svt_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.svt_hbm_copy_peak.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    L.svt_batch_repack.argtypes = [vp, vp]
    L.svt_batch_slice.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.svt_count_partial_device.argtypes = [vp, vp, C.c_uint32, C.c_uint8, vp, C.POINTER(C.c_uint64)]
    L.svt_count_export_device.argtypes = [vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.svt_count_merge_begin.argtypes = [vp, C.c_uint64]
    L.svt_count_merge_device.argtypes = [vp, vp, vp, vp, C.c_uint64]
    L.svt_fork.argtypes = [vp, C.POINTER(vp)]
    L.svt_fork_refresh.argtypes = [vp]
    L.svt_destroy.argtypes = [vp]
    L.svt_destroy.restype = None
    L.svt_last_error.argtypes = [vp]
    L.svt_last_error.restype = C.c_char_p
    L.svt_profile_enable.argtypes = [vp, C.c_int]
    L.svt_profile_reset.argtypes = [vp]
    L.svt_profile_reset.restype = None
    L.svt_profile_count.argtypes = [vp]
    L.svt_profile_get.argtypes = [vp, C.c_int, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.svt_batch_upload.argtypes = [vp, vp, vp, vp, C.c_uint32, C.POINTER(vp)]
    L.svt_batch_free.argtypes = [vp, vp]
    L.svt_batch_free.restype = None
    L.svt_batch_size.argtypes = [vp]
    L.svt_batch_size.restype = C.c_uint32
    L.svt_batch_fetch_packed.argtypes = [vp, vp, C.c_uint32, vp, vp]
    L.svt_split_kmers_emit.argtypes = [vp, vp, C.c_uint32, C.c_uint8, vp, vp, vp, vp]
    L.svt_count_split_kmers.argtypes = [vp, vp, C.c_uint32, C.c_uint8, vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.svt_count_fetch.argtypes = [vp, vp, vp, vp]
    L.svt_count_candidates_sizes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.svt_count_candidates_fetch.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.svt_count_partial.argtypes = [vp, vp, C.c_uint32, C.c_uint8, vp, C.POINTER(C.c_uint64)]
    L.svt_count_export.argtypes = [vp, vp, vp, vp]
    L.svt_count_merge.argtypes = [vp, vp, vp, vp, C.c_uint64]
    L.svt_count_finalize.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.svt_set_snpmers.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.c_uint32, vp, C.c_uint32]
    L.svt_snpmer_site_order.argtypes = [vp, vp]
    L.svt_extract_seeds.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint8, C.c_int]
    L.svt_seeds_sizes.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.svt_seeds_fetch.argtypes = [vp, vp, C.POINTER(SeedsOut)]
    L.svt_minimizer_shared_counts.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, vp, vp]
    L.svt_snpmer_words.argtypes = [vp]
    L.svt_snpmer_words.restype = C.c_uint32
    L.svt_snpmer_bits_fetch.argtypes = [vp, vp, vp, vp, vp]
    L.svt_bitset_upload.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(vp)]
    L.svt_bitset_free.argtypes = [vp, vp]
    L.svt_bitset_free.restype = None
    L.svt_snpmer_compat_lists.argtypes = [vp, vp, C.c_int, vp, C.c_uint32, vp, C.c_int, vp, vp, C.c_uint32, C.c_int, C.c_int, C.c_uint32, vp,
                                          vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.svt_snpmer_best_column.argtypes = [vp, vp, C.c_int, vp, C.c_uint32, vp, vp, vp, vp, vp]
    L.svt_snpmer_consensus.argtypes = [vp, vp, vp, vp, C.c_uint32, vp, vp, C.POINTER(vp)]
    L.svt_align_nm.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_uint64, vp]
    L.svt_align_nm_affine.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp]
    L.svt_align_nm_affine_near.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp]
    L.svt_set_shard.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp]
    L.svt_twin_order.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp, vp, vp]
    L.svt_twin_gather.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.svt_lsh_candidates.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    L.svt_shard_comm_id.argtypes = [vp]
    L.svt_set_shard_comm.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.svt_count_shard_merge.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.svt_host_pin.argtypes = [vp, vp, C.c_uint64]
    L.svt_shard_info.argtypes = [vp, vp, vp]
    L.svt_shard_pause.argtypes = [vp, C.c_int]
    L.svt_shard_abort.argtypes = [vp, C.c_char_p]
    L.svt_shard_allgather_u64.argtypes = [vp, C.c_uint64, vp]
    L.svt_shard_allgatherv.argtypes = [vp, vp, vp, vp]
    L.svt_host_unpin.argtypes = [vp, vp]
    L.svt_align_pileup.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp, vp]
    L.svt_pileup_create.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, C.c_uint32, C.POINTER(vp), vp, vp]
    L.svt_pileup_free.argtypes = [vp, vp]
    L.svt_pileup_free.restype = None
    L.svt_pileup_cells.argtypes = [vp]
    L.svt_pileup_cells.restype = C.c_uint64
    L.svt_pileup_columns.argtypes = [vp]
    L.svt_pileup_columns.restype = C.c_uint64
    L.svt_pileup_fetch.argtypes = [vp, vp, vp, vp]
    L.svt_pileup_stats.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.svt_pileup_loglik.argtypes = [vp, vp, vp, C.c_double, C.c_double, vp, vp]
    L.svt_pileup_hp_median.argtypes = [vp, vp, vp]
    L.svt_qualbin_mean.argtypes = [vp, vp, vp, vp]
    L.svt_batch_set_tags.argtypes = [vp, vp, vp, vp]
    L.svt_read_asv_ties.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_double, C.c_double, vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.svt_snpmer_compat_lists_seg.argtypes = [vp, vp, C.c_int, vp, C.c_uint32, vp, vp, vp, C.c_uint32, C.c_int, vp, vp, vp, C.c_uint64, vp]
    L.svt_snpmer_compat_rows_seg.argtypes = [vp, vp, C.c_int, vp, C.c_uint32, vp, vp, vp, C.c_uint32, C.c_int, vp, vp, vp, C.c_uint64, vp]
    L.svt_poa_graphs.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.svt_poa_graphs_submit.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp]
    L.svt_poa_graphs_wait.argtypes = [vp, vp, vp, vp]
    L.svt_poa_graphs_fetch.argtypes = [vp, vp, vp, vp]
    _lib = L
    return L


def shard_comm_id():
    """128 bytes (an ncclUniqueId) for svt_set_shard_comm; call on ONE rank and broadcast"""
    L = load()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = L.svt_shard_comm_id(buf)
    if rc != 0:
        raise SavontHipError("svt_shard_comm_id failed (%d): RCCL could not be loaded or gave no id" % rc)
    return bytes(buf)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


class Batch:
    def __init__(self, dev, handle, n, offsets):
        self.dev, self.h, self.n = dev, handle, n
        self.offsets = offsets
        self.lengths = (offsets[1:] - offsets[:-1]).astype(np.uint32)

    def free(self):
        if self.h:
            self.dev.L.svt_batch_free(self.dev.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Device:
    """One svt_ctx (one GPU).  Raises SavontHipError on any non-zero return code."""

    def __init__(self, device_id=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.svt_create(device_id, C.byref(h))
        if rc != SVT_OK:
            raise SavontHipError("svt_create(%d) failed with %d: no gfx950 device (the HIP path has no CPU fallback)" % (device_id, rc))
        self.h = h

    def close(self):
        if self.h:
            self.L.svt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=()):
        if rc != SVT_OK and rc not in allow:
            raise SavontHipError("libsavont_hip error %d: %s" % (rc, self.L.svt_last_error(self.h).decode()))
        return rc

    def set_option(self, key, value):
        """svt_set_option: kernel / copy-path selection (identical results); see include/savont_hip.h"""
        self._chk(self.L.svt_set_option(self.h, key.encode(), int(value)))

    def hbm_copy_peak(self, nbytes=1 << 30, iters=5):
        """measured HBM streaming-copy rate in GB/s (read + written bytes)"""
        v = C.c_double()
        self._chk(self.L.svt_hbm_copy_peak(self.h, int(nbytes), int(iters), C.byref(v)))
        return v.value

    def set_shard(self, rank, world, hook):
        """svt_set_shard: this context runs only its slice of the tiled calls and completes the results through `hook` (savont_amd/shard.py:
        a ctypes EXCHANGE_FN the caller keeps alive); hook=None switches sharding off"""
        fn = C.cast(hook, C.c_void_p) if hook is not None else None
        self._chk(self.L.svt_set_shard(self.h, int(rank), int(world), fn, None))

    def set_shard_comm(self, rank, world, comm_id):
        """svt_set_shard_comm: the library creates an RCCL communicator from the 128 id bytes of shard_comm_id() (made on one rank, handed to all
        by the caller) and issues every exchange itself as one grouped collective on its stream -- no callback"""
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        self._chk(self.L.svt_set_shard_comm(self.h, int(rank), int(world), buf))

    def get_option(self, key):
        v = C.c_int64()
        self._chk(self.L.svt_get_option(self.h, key.encode(), C.byref(v)))
        return v.value

    # ---- profiling
    def profile(self, on=True):
        self.L.svt_profile_enable(self.h, int(on) if on else 0)                # True / 1: every kernel; 2: the roofline kernels only

    def profile_reset(self):
        self.L.svt_profile_reset(self.h)

    def profile_table(self):
        out = {}
        for i in range(self.L.svt_profile_count(self.h)):
            name = C.create_string_buffer(64)
            n = C.c_uint64(); ms = C.c_double(); by = C.c_double(); un = C.c_double()
            self.L.svt_profile_get(self.h, i, name, C.byref(n), C.byref(ms), C.byref(by), C.byref(un))
            out[name.value.decode()] = dict(launches=n.value, ms=ms.value, algo_bytes=by.value, units=un.value)
        return out

    # ---- batches
    def upload(self, seq, qual, offsets):
        seq = _c(seq, np.uint8); qual = _c(qual, np.uint8); offsets = _c(offsets, np.uint64)
        n = len(offsets) - 1
        h = C.c_void_p()
        self._chk(self.L.svt_batch_upload(self.h, _p(seq), _p(qual), _p(offsets), n, C.byref(h)))
        return Batch(self, h, n, offsets - offsets[0])

    def fetch_packed(self, b, read):
        nw = (int(b.lengths[read]) + 15) // 16
        w = np.zeros(nw, np.uint32); m = np.zeros(nw, np.uint16)
        self._chk(self.L.svt_batch_fetch_packed(self.h, b.h, read, _p(w), _p(m)))
        return w, m

    # ---- stage 1
    def split_kmers_emit(self, b, k, min_bq, rc_flags=None):
        need = np.maximum(b.lengths.astype(np.int64) - k + 1, 0).astype(np.uint64)
        off = np.zeros(b.n + 1, np.uint64); np.cumsum(need, out=off[1:])
        out = np.zeros(int(off[-1]), np.uint64); cnt = np.zeros(b.n, np.uint32)
        rc_flags = _c(rc_flags, np.uint8)
        self._chk(self.L.svt_split_kmers_emit(self.h, b.h, k, min_bq, _p(rc_flags), _p(off), _p(out), _p(cnt)))
        return off, out, cnt

    def _count_fetch(self, n):
        km = np.zeros(n, np.uint64); rev = np.zeros(n, np.uint32); fwd = np.zeros(n, np.uint32)
        self._chk(self.L.svt_count_fetch(self.h, _p(km), _p(rev), _p(fwd)))
        return km, rev, fwd

    def count_split_kmers(self, b, k, min_bq, rc_flags=None, single_strand=False):
        rc_flags = _c(rc_flags, np.uint8)
        nd = C.c_uint64(); nk = C.c_uint64()
        self._chk(self.L.svt_count_split_kmers(self.h, b.h, k, min_bq, _p(rc_flags), int(single_strand), C.byref(nd), C.byref(nk)))
        return (nd.value,) + self._count_fetch(nk.value)

    def count_candidates(self):
        """the two Stage-1b selections of the sorted table -> (n_table, (g_kmer, g_rev, g_fwd), (h_kmer, h_rev, h_fwd))"""
        nt = C.c_uint64(); ng = C.c_uint64(); nh = C.c_uint64()
        self._chk(self.L.svt_count_candidates_sizes(self.h, C.byref(nt), C.byref(ng), C.byref(nh)))
        g = (np.zeros(ng.value, np.uint64), np.zeros(ng.value, np.uint32), np.zeros(ng.value, np.uint32))
        h = (np.zeros(nh.value, np.uint64), np.zeros(nh.value, np.uint32), np.zeros(nh.value, np.uint32))
        self._chk(self.L.svt_count_candidates_fetch(self.h, _p(g[0]), _p(g[1]), _p(g[2]), _p(h[0]), _p(h[1]), _p(h[2])))
        return nt.value, g, h

    def count_partial(self, b, k, min_bq, rc_flags=None):
        rc_flags = _c(rc_flags, np.uint8)
        nd = C.c_uint64()
        self._chk(self.L.svt_count_partial(self.h, b.h, k, min_bq, _p(rc_flags), C.byref(nd)))
        return self._count_fetch(nd.value)

    def count_merge(self, km, rev, fwd):
        km = _c(km, np.uint64); rev = _c(rev, np.uint32); fwd = _c(fwd, np.uint32)
        self._chk(self.L.svt_count_merge(self.h, _p(km), _p(rev), _p(fwd), len(km)))

    def count_finalize(self, k, single_strand=False):
        nd = C.c_uint64(); nk = C.c_uint64()
        self._chk(self.L.svt_count_finalize(self.h, k, int(single_strand), C.byref(nd), C.byref(nk)))
        return (nd.value,) + self._count_fetch(nk.value)

    def set_snpmers(self, k, split, mid0, mid1, high_freq, weight=None):
        split = _c(split, np.uint64); mid0 = _c(mid0, np.uint8); mid1 = _c(mid1, np.uint8); high_freq = _c(high_freq, np.uint64)
        weight = _c(weight, np.uint32)
        self._keep_snp = (split, mid0, mid1, high_freq, weight)
        self._n_sites = len(split)
        self._chk(self.L.svt_set_snpmers(self.h, k, _p(split), _p(mid0), _p(mid1), _p(weight), len(split), _p(high_freq), len(high_freq)))

    def site_order(self):
        o = np.zeros(self._n_sites, np.uint32)
        self._chk(self.L.svt_snpmer_site_order(self.h, _p(o)))
        return o

    # ---- seeds
    def extract_seeds(self, b, k, c, min_bq, use_qual=True):
        self._chk(self.L.svt_extract_seeds(self.h, b.h, k, c, min_bq, int(use_qual)))

    def fetch_seeds(self, b, qualbins=True):
        nm = C.c_uint64(); ns = C.c_uint64(); nq = C.c_uint64()
        self._chk(self.L.svt_seeds_sizes(self.h, b.h, C.byref(nm), C.byref(ns), C.byref(nq)))
        n = b.n
        r = dict(
            mini_off=np.zeros(n + 1, np.uint64), mini_pos=np.zeros(nm.value, np.uint32), mini_kmer=np.zeros(nm.value, np.uint64),
            mini_flags=np.zeros(nm.value, np.uint8), snp_off=np.zeros(n + 1, np.uint64), snp_pos=np.zeros(ns.value, np.uint32),
            snp_kmer=np.zeros(ns.value, np.uint64), snp_flags=np.zeros(ns.value, np.uint8), est_id=np.zeros(n, np.float64),
            est_valid=np.zeros(n, np.uint8), lsh=np.zeros((n, LSH_TABLES), np.uint64), lsh_valid=np.zeros(n, np.uint8),
            n_unique=np.zeros(n, np.uint32), n_solid=np.zeros(n, np.uint32), qualbin_off=np.zeros(n + 1, np.uint64),
            qualbins=np.zeros(nq.value if qualbins else 0, np.uint8), status=np.zeros(n, np.uint8))
        so = SeedsOut()
        for name, _ in SeedsOut._fields_:
            a = r[name]
            setattr(so, name, a.ctypes.data if a.size else None)
        self._chk(self.L.svt_seeds_fetch(self.h, b.h, C.byref(so)))
        return r

    def lsh_candidates(self, B, q_idx, r_idx, ref_limit=None, mode=0, top_n=10, cap=64, capacity=None):
        """Stage-2 candidate lists of a block (svt_lsh_candidates) -> list per query: None when the device flagged it (more than cap, or the flat output full), else [n, 2] entries"""
        q_idx = _c(q_idx, np.uint32); r_idx = _c(r_idx, np.uint32)
        lim = None if ref_limit is None else _c(ref_limit, np.uint32)
        capacity = 16 * len(q_idx) if capacity is None else capacity
        cnt = np.zeros(len(q_idx), np.uint32); off = np.zeros(len(q_idx), np.uint32); out = np.zeros((max(capacity, 1), 2), np.uint32); used = C.c_uint32()
        self._chk(self.L.svt_lsh_candidates(self.h, B.h, _p(q_idx), len(q_idx), _p(r_idx), len(r_idx), None if lim is None else _p(lim), mode, top_n, cap, capacity,
                                            _p(cnt), _p(off), _p(out), C.byref(used)))
        return [None if cnt[i] == 0xFFFFFFFF else out[off[i]:off[i] + cnt[i]] for i in range(len(q_idx))]

    def minimizer_shared_counts(self, A, B, a_idx, b_idx):
        a_idx = _c(a_idx, np.uint32); b_idx = _c(b_idx, np.uint32)
        n = len(a_idx)
        sh = np.zeros(n, np.uint32); sm = np.zeros(n, np.uint32)
        self._chk(self.L.svt_minimizer_shared_counts(self.h, A.h, B.h, _p(a_idx), _p(b_idx), n, _p(sh), _p(sm)))
        return sh, sm

    def read_asv_ties(self, R, row_idx, A, n_asvs, row_max_mismatch, min_frac, c_param, with_mismatches=False):
        """fused Stage-7 candidate scoring -> (tie_row, tie_col, tie_rev, n_candidates[, tie_mismatches]), unordered"""
        row_idx = _c(row_idx, np.uint32); row_max_mismatch = _c(row_max_mismatch, np.uint32)
        cap = max(1024, 4 * len(row_idx))
        while True:
            tr = np.zeros(cap, np.uint32); tc = np.zeros(cap, np.uint32); tv = np.zeros(cap, np.uint8)
            tm = np.zeros(cap, np.uint32) if with_mismatches else None
            nt = C.c_uint64(); nc = C.c_uint64()
            rc = self.L.svt_read_asv_ties(self.h, R.h, _p(row_idx), len(row_idx), A.h, n_asvs, _p(row_max_mismatch), float(min_frac), float(c_param),
                                          _p(tr), _p(tc), _p(tv), _p(tm), cap, C.byref(nt), C.byref(nc))
            if rc == SVT_ERR_OVERFLOW:
                cap = int(nt.value) + 1024
                continue
            self._chk(rc)
            n = int(nt.value)
            return (tr[:n], tc[:n], tv[:n], int(nc.value), tm[:n]) if with_mismatches else (tr[:n], tc[:n], tv[:n], int(nc.value))

    # ---- SNPmer bitsets
    def snpmer_words(self):
        return self.L.svt_snpmer_words(self.h)

    def snpmer_bits(self, b):
        w = self.snpmer_words()
        pa = np.zeros((b.n, w), np.uint64); pf = np.zeros((b.n, w), np.uint64); al = np.zeros((b.n, w), np.uint64)
        self._chk(self.L.svt_snpmer_bits_fetch(self.h, b.h, _p(pa), _p(pf), _p(al)))
        return pa, pf, al

    def bitset_upload(self, presence, allele):
        presence = _c(presence, np.uint64); allele = _c(allele, np.uint64)
        h = C.c_void_p()
        self._chk(self.L.svt_bitset_upload(self.h, _p(presence), _p(allele), presence.shape[0], C.byref(h)))
        return h

    def bitset_free(self, h):
        self.L.svt_bitset_free(self.h, h)

    def compat_lists(self, R, row_view, row_idx, C_batch=None, col_view=VIEW_ALL, S=None, col_idx=None, n_cols=None,
                     filt=LIST_COMPATIBLE, triangular=False, tri_base=0, cap=None, row_max_mismatch=None):
        row_idx = _c(row_idx, np.uint32); col_idx = _c(col_idx, np.uint32); row_max_mismatch = _c(row_max_mismatch, np.uint32)
        if n_cols is None:
            n_cols = len(col_idx)
        cap = cap or max(1024, 8 * len(row_idx))
        while True:
            orow = np.zeros(cap, np.uint32); ocol = np.zeros(cap, np.uint32); omm = np.zeros(cap, np.uint32)
            n = C.c_uint64()
            rc = self.L.svt_snpmer_compat_lists(self.h, R.h, row_view, _p(row_idx), len(row_idx), C_batch.h if C_batch else None, col_view, S,
                                                _p(col_idx), n_cols, filt, int(triangular), tri_base, _p(row_max_mismatch), _p(orow), _p(ocol), _p(omm), cap, C.byref(n))
            if rc == SVT_ERR_OVERFLOW:
                cap = int(n.value) + 1024
                continue
            self._chk(rc)
            k = n.value
            return orow[:k], ocol[:k], omm[:k] >> 16, omm[:k] & 0xFFFF

    def compat_lists_seg(self, R, view, row_idx, seg_row_off, col_idx, seg_col_off, filt=LIST_COMPATIBLE, cap=None, by_rows=False):
        """svt_snpmer_compat_lists_seg -> (row, col, matches, mismatches) triples; by_rows: svt_snpmer_compat_rows_seg -> (offsets, col, matches, mismatches)"""
        row_idx = _c(row_idx, np.uint32); col_idx = _c(col_idx, np.uint32); seg_row_off = _c(seg_row_off, np.uint32); seg_col_off = _c(seg_col_off, np.uint32)
        cap = cap or max(1024, 8 * len(row_idx))
        fn = self.L.svt_snpmer_compat_rows_seg if by_rows else self.L.svt_snpmer_compat_lists_seg
        while True:
            orow = np.zeros(len(row_idx) + 1 if by_rows else cap, np.uint32); ocol = np.zeros(cap, np.uint32); omm = np.zeros(cap, np.uint32)
            n = C.c_uint64()
            rc = fn(self.h, R.h, view, _p(row_idx), len(row_idx), _p(seg_row_off), _p(col_idx), _p(seg_col_off), len(seg_row_off) - 1, filt, _p(orow), _p(ocol), _p(omm), cap, C.byref(n))
            if rc == SVT_ERR_OVERFLOW:
                assert not by_rows or not orow.any()
                cap = int(n.value) + 1024
                continue
            self._chk(rc)
            k = n.value
            return (orow if by_rows else orow[:k]), ocol[:k], omm[:k] >> 16, omm[:k] & 0xFFFF

    def best_column(self, R, row_view, row_idx, S, col_lo=None, col_hi=None):
        row_idx = _c(row_idx, np.uint32); col_lo = _c(col_lo, np.uint32); col_hi = _c(col_hi, np.uint32)
        bc = np.zeros(len(row_idx), np.uint32); bs = np.zeros(len(row_idx), np.uint32)
        self._chk(self.L.svt_snpmer_best_column(self.h, R.h, row_view, _p(row_idx), len(row_idx), S, _p(col_lo), _p(col_hi), _p(bc), _p(bs)))
        return bc, bs >> 16, bs & 0xFFFF

    def consensus(self, R, clusters, keep_set=False):
        """clusters: list of arrays of read indices -> (presence, allele[, svt_bitset handle])"""
        off = np.zeros(len(clusters) + 1, np.uint64)
        np.cumsum([len(x) for x in clusters], out=off[1:])
        mem = np.concatenate(clusters).astype(np.uint32) if clusters else np.zeros(0, np.uint32)
        w = self.snpmer_words()
        p = np.zeros((len(clusters), w), np.uint64); a = np.zeros((len(clusters), w), np.uint64)
        h = C.c_void_p()
        self._chk(self.L.svt_snpmer_consensus(self.h, R.h, _p(off), _p(mem), len(clusters), _p(p), _p(a), C.byref(h) if keep_set else None))
        return (p, a, h) if keep_set else (p, a)

    def align_nm(self, Q, T, q_idx, t_idx, reverse, band):
        q_idx = _c(q_idx, np.uint32); t_idx = _c(t_idx, np.uint32); reverse = _c(reverse, np.uint8); band = _c(band, np.uint32)
        nm = np.zeros(len(q_idx), np.int32)
        self._chk(self.L.svt_align_nm(self.h, Q.h, T.h, _p(q_idx), _p(t_idx), _p(reverse), _p(band), len(q_idx), _p(nm)))
        return nm

    def align_nm_affine(self, Q, T, q_idx, t_idx, reverse, band):
        """K8a (minimap2-style nm of the best local two-piece-affine alignment) -> (nm i32[n], score i32[n])"""
        q_idx = _c(q_idx, np.uint32); t_idx = _c(t_idx, np.uint32); reverse = _c(reverse, np.uint8); band = _c(band, np.uint32)
        nm = np.zeros(len(q_idx), np.int32); score = np.zeros(len(q_idx), np.int32)
        self._chk(self.L.svt_align_nm_affine(self.h, Q.h, T.h, _p(q_idx), _p(t_idx), _p(reverse), _p(band), len(q_idx), _p(nm), _p(score)))
        return nm, score

    def align_nm_affine_near(self, Q, T, q_idx, t_idx, reverse, band):
        """K8a inside the band around the unit-cost optimum (Stage 7's default nm) -> (nm i32[n], score i32[n], band_used u32[n])"""
        q_idx = _c(q_idx, np.uint32); t_idx = _c(t_idx, np.uint32); reverse = _c(reverse, np.uint8); band = _c(band, np.uint32)
        nm = np.zeros(len(q_idx), np.int32); score = np.zeros(len(q_idx), np.int32); used = np.zeros(len(q_idx), np.uint32)
        self._chk(self.L.svt_align_nm_affine_near(self.h, Q.h, T.h, _p(q_idx), _p(t_idx), _p(reverse), _p(band), len(q_idx), _p(nm), _p(score), _p(used)))
        return nm, score, used

    def pileup_create(self, Q, T, q_idx, t_idx, reverse, band, grp_off):
        """K9 with device-resident rows -> (handle, span u32[n,4], nm i32[n]); free with pileup_free"""
        q_idx = _c(q_idx, np.uint32); t_idx = _c(t_idx, np.uint32); reverse = _c(reverse, np.uint8); band = _c(band, np.uint32)
        grp_off = _c(grp_off, np.uint64)
        n = len(q_idx)
        span = np.zeros((max(n, 1), 4), np.uint32); nm = np.zeros(max(n, 1), np.int32)
        h = C.c_void_p()
        self._chk(self.L.svt_pileup_create(self.h, Q.h, T.h, _p(q_idx), _p(t_idx), _p(reverse), _p(band), n, _p(grp_off), len(grp_off) - 1, C.byref(h), _p(span), _p(nm)))
        return h, span[:n], nm[:n]

    def pileup_free(self, h):
        self.L.svt_pileup_free(self.h, h)

    def pileup_fetch(self, h, n_pairs):
        cells = np.zeros(max(1, self.L.svt_pileup_cells(h)), np.uint64); off = np.zeros(n_pairs + 1, np.uint64)
        self._chk(self.L.svt_pileup_fetch(self.h, h, _p(cells), _p(off)))
        return cells[:int(off[-1])], off

    def pileup_stats(self, h, grp_selected):
        n = self.L.svt_pileup_columns(h)
        depth = np.zeros(max(1, n), np.uint32); err = np.zeros(max(1, n), np.uint32); qt = np.zeros(256, np.uint64); qe = np.zeros(256, np.uint64)
        self._chk(self.L.svt_pileup_stats(self.h, h, _p(_c(grp_selected, np.uint8)), _p(depth), _p(err), _p(qt), _p(qe)))
        return depth[:n], err[:n], qt, qe

    def qualbin_mean(self, B, table16):
        """per read: mean of table16[bin] over its 4-bit quality bins, in bin order (f64)"""
        table16 = _c(table16, np.float64); out = np.zeros(max(1, B.n), np.float64)
        self._chk(self.L.svt_qualbin_mean(self.h, B.h, _p(table16), _p(out)))
        return out[:B.n]

    def pileup_hp_median(self, h):
        n = self.L.svt_pileup_columns(h)
        med = np.zeros(max(1, n), np.uint8)
        self._chk(self.L.svt_pileup_hp_median(self.h, h, _p(med)))
        return med[:n]

    def batch_set_tags(self, B, qual, hp_len):
        """per-base tags of a batch of homopolymer-compressed reads (K9 then reports them in the pile-up cells)"""
        qual = _c(qual, np.uint8); hp_len = _c(hp_len, np.uint8)
        self._chk(self.L.svt_batch_set_tags(self.h, B.h, _p(qual), _p(hp_len)))

    def pileup_loglik(self, h, ln_table, ln_indel_err, ln_indel_acc):
        n = self.L.svt_pileup_columns(h)
        lr = np.zeros(max(1, n), np.float64); ln = np.zeros(max(1, n), np.float64)
        self._chk(self.L.svt_pileup_loglik(self.h, h, _p(_c(ln_table, np.float64)), float(ln_indel_err), float(ln_indel_acc), _p(lr), _p(ln)))
        return lr[:n], ln[:n]

    def align_pileup(self, Q, T, q_idx, t_idx, reverse, band):
        """-> (cell_off u64[n+1], cells u64[total], span u32[n,4], nm i32[n])"""
        q_idx = _c(q_idx, np.uint32); t_idx = _c(t_idx, np.uint32); reverse = _c(reverse, np.uint8); band = _c(band, np.uint32)
        n = len(q_idx)
        off = np.zeros(n + 1, np.uint64); np.cumsum(Q.lengths[q_idx].astype(np.uint64), out=off[1:])
        cells = np.zeros(int(off[-1]), np.uint64); span = np.zeros((n, 4), np.uint32); nm = np.zeros(n, np.int32)
        self._chk(self.L.svt_align_pileup(self.h, Q.h, T.h, _p(q_idx), _p(t_idx), _p(reverse), _p(band), n, _p(off), _p(cells), _p(span), _p(nm)))
        return off, cells, span, nm
