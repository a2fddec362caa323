"""ctypes binding of the C++ host pipeline (savont_amd/libsavont_asv.so), which sits above the C-ABI
device layer and mirrors the reference's stage functions (src/main.rs:49-201 call order):

    read_to_split_kmers -> get_snpmers_inplace_sort -> twin_reads_from_snpmers ->
    cluster_reads_by_kmers -> cluster_reads_by_snpmers -> refine_asv_depths_with_em
    [-> compute_per_sample_depths]

Python here is harness plumbing only (tests, bench.py); the product logic is C++/HIP.
"""
import ctypes as C
import os

import numpy as np

from . import hip

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsavont_asv.so")


class Args(C.Structure):
    _fields_ = [
        ("kmer_size", C.c_uint32), ("c", C.c_uint32), ("min_read_length", C.c_uint32), ("max_read_length", C.c_uint32),
        ("quality_value_cutoff", C.c_double), ("minimum_base_quality", C.c_uint32), ("single_strand", C.c_uint32),
        ("min_cluster_size", C.c_uint32), ("max_iterations_recluster", C.c_uint32), ("primary_clustering_threshold", C.c_double),
        ("low_polymorphism", C.c_uint32), ("align_band", C.c_uint32),
        ("n_depth_cutoff", C.c_uint32), ("mask_low_quality", C.c_uint32), ("posterior_threshold_ln", C.c_double),
        ("chimera_allowable_errors", C.c_uint32), ("chimera_detect_length", C.c_uint32), ("skip_chimera_detection", C.c_uint32), ("use_hpc", C.c_uint32), ("no_snpmers", C.c_uint32), ("no_band", C.c_uint32),
    ]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    hip.load()   # libsavont_asv.so links libsavont_hip.so; fail loudly if the device layer is missing
    if not os.path.exists(LIB_PATH):
        raise hip.SavontHipError("host pipeline library missing: %s (run __graft_entry__.build())" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.svh_default_args.argtypes = [C.POINTER(Args)]
    L.svh_create.argtypes = [C.c_int, C.POINTER(Args), C.POINTER(vp)]
    L.svh_destroy.argtypes = [vp]
    L.svh_destroy.restype = None
    L.svh_last_error.argtypes = [vp]
    L.svh_last_error.restype = C.c_char_p
    L.svh_ctx.argtypes = [vp]
    L.svh_ctx.restype = vp
    L.svh_stage_seconds.argtypes = [vp, C.c_char_p]
    L.svh_stage_seconds.restype = C.c_double
    L.svh_set_reads.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_char_p, vp]
    L.svh_repack.argtypes = [vp]
    L.svh_set_temp_dir.argtypes = [vp, C.c_char_p]
    for n in ("svh_read_to_split_kmers", "svh_get_snpmers", "svh_twin_reads", "svh_cluster_reads_by_kmers", "svh_cluster_reads_by_snpmers",
              "svh_refine_asv_depths_with_em", "svh_auto_low_polymorphism"):
        getattr(L, n).argtypes = [vp]
    for n in ("svh_count_distinct", "svh_count_size"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = C.c_uint64
    for n in ("svh_snpmer_count", "svh_high_freq_thresh", "svh_high_freq_count", "svh_twin_count"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = C.c_uint32
    L.svh_count_fetch.argtypes = [vp, vp, vp, vp]
    L.svh_consensus.argtypes = [vp, C.c_int]
    L.svh_consensus_count.argtypes = [vp, C.c_int]
    L.svh_consensus_count.restype = C.c_uint32
    L.svh_consensus_bases.argtypes = [vp, C.c_int]
    L.svh_consensus_bases.restype = C.c_uint64
    L.svh_consensus_fetch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.svh_consensus_fetch.restype = None
    L.svh_quality_map.argtypes = [vp, vp, vp]
    L.svh_quality_map.restype = C.c_uint32
    L.svh_poa_consensus.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]
    L.svh_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.svh_consensus_to_asvs.argtypes = [vp]
    L.svh_merge_similar_consensuses.argtypes = [vp]
    L.svh_detect_chimeras.argtypes = [vp]
    L.svh_chimera_count.argtypes = [vp]
    L.svh_chimera_count.restype = C.c_uint32
    L.svh_chimera_fetch.argtypes = [vp, vp]
    L.svh_chimera_fetch.restype = None
    L.svh_minimizer_seeds.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, C.c_uint64]
    L.svh_minimizer_seeds.restype = C.c_uint64
    L.svh_poa_consensus_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint64, vp]
    L.svh_fastx_digest.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64]
    L.svh_load_fastx.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint32)]
    L.svh_write_outputs.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int]
    L.svh_keep_pileups.argtypes = [vp, C.c_int]
    L.svh_keep_pileups.restype = None
    L.svh_pileup_entries.argtypes = [vp, C.c_uint32]
    L.svh_pileup_entries.restype = C.c_uint64
    L.svh_pileup_fetch.argtypes = [vp, C.c_uint32, vp, vp, vp, vp]
    L.svh_pileup_fetch.restype = None
    L.svh_pileup_fetch_hp.argtypes = [vp, C.c_uint32, vp]
    L.svh_pileup_fetch_hp.restype = None
    L.svh_raw_consensus_count.argtypes = [vp]
    L.svh_raw_consensus_count.restype = C.c_uint32
    L.svh_raw_consensus_len.argtypes = [vp, C.c_uint32]
    L.svh_raw_consensus_len.restype = C.c_uint64
    L.svh_raw_consensus_fetch.argtypes = [vp, C.c_uint32, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.svh_raw_consensus_fetch.restype = None
    L.svh_set_count_table.argtypes = [vp, vp, vp, vp, C.c_uint64]
    L.svh_snpmer_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
    L.svh_high_freq_fetch.argtypes = [vp, vp]
    L.svh_set_snpmers.argtypes = [vp, vp, vp, vp, C.c_uint32, vp, C.c_uint32]
    L.svh_twin_meta.argtypes = [vp] + [vp] * 9
    L.svh_cluster_count.argtypes = [vp, C.c_int]
    L.svh_cluster_count.restype = C.c_uint32
    L.svh_cluster_total.argtypes = [vp, C.c_int]
    L.svh_cluster_total.restype = C.c_uint64
    L.svh_clusters_fetch.argtypes = [vp, C.c_int, vp, vp, vp]
    L.svh_set_asvs.argtypes = [vp, vp, vp, C.c_uint32]
    L.svh_em_fetch.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.svh_em_read_assignments.argtypes = [vp, vp, vp, vp]
    L.svh_compute_per_sample_depths.argtypes = [vp, C.c_uint32, vp]
    L.svh_synth_reads.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp]
    L.svh_synth_reads.restype = C.c_uint64
    L.svh_binomial_test.argtypes = [C.c_uint64, C.c_uint64, C.c_double]
    L.svh_binomial_test.restype = C.c_double
    L.svh_fisher_two_tail.argtypes = [C.c_uint32] * 4
    L.svh_fisher_two_tail.restype = C.c_double
    L.svh_snpmers_from_table.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    _bind_pooled(L)                     # the sharded halves of stages 1a / 4a / 7 (savont_amd/pooled.py drives them; binding them needs no torch)
    _lib = L
    return L


def _bind_pooled(L):
    """ctypes signatures of the pooled-mode entry points of libsavont_asv.so (include/savont_asv.h, last group)"""
    vp = C.c_void_p
    L.svh_count_partial_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.svh_count_export_device.argtypes = [vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.svh_count_merge_begin.argtypes = [vp, C.c_uint64]
    L.svh_count_merge_device.argtypes = [vp, vp, vp, vp, C.c_uint64]
    L.svh_count_finalize.argtypes = [vp]
    L.svh_consensus_poa.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32]
    L.svh_consensus_raw_count.argtypes = [vp]; L.svh_consensus_raw_count.restype = C.c_uint32
    L.svh_consensus_raw_bytes.argtypes = [vp]; L.svh_consensus_raw_bytes.restype = C.c_uint64
    L.svh_consensus_raw_export.argtypes = [vp, vp, vp]; L.svh_consensus_raw_export.restype = None
    L.svh_consensus_raw_import.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint64]
    L.svh_consensus_polish.argtypes = [vp]
    L.svh_em_begin.argtypes = [vp]
    L.svh_em_classes.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.svh_em_classes_members.argtypes = [vp, C.c_uint32, C.c_uint32]; L.svh_em_classes_members.restype = C.c_uint64
    L.svh_em_classes_export.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp]; L.svh_em_classes_export.restype = None
    L.svh_em_classes_import.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_uint64]
    L.svh_em_finish.argtypes = [vp]
    L.svh_set_shard_comm.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.svh_run_asv.argtypes = [vp]
    L.svh_count_shard_merge.argtypes = [vp]
    L.svh_snpmers_check_ranks.argtypes = [vp]
    L.svh_consensus_gather.argtypes = [vp]
    L.svh_em_classes_gather.argtypes = [vp]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_args(**kw):
    a = Args()
    load().svh_default_args(C.byref(a))
    known = {f[0] for f in Args._fields_}
    for k, v in kw.items():
        if k not in known:
            raise TypeError("unknown pipeline parameter %r (known: %s)" % (k, ", ".join(sorted(known))))
        setattr(a, k, v)
    return a


class AsvPipeline:
    def __init__(self, device_id=0, **args):
        self.L = load()
        self.args = default_args(**args)
        h = C.c_void_p()
        rc = self.L.svh_create(device_id, C.byref(self.args), C.byref(h))
        if rc != 0:
            raise hip.SavontHipError("svh_create failed with %d: no gfx950 device (no CPU fallback exists)" % rc)
        self.h = h
        self.n_asvs = 0

    def close(self):
        if self.h:
            self.L.svh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise hip.SavontHipError("%s failed (%d): %s" % (what, rc, self.L.svh_last_error(self.h).decode()))

    def device(self):
        """a hip.Device view sharing this pipeline's svt_ctx (for profiling)"""
        d = hip.Device.__new__(hip.Device)
        d.L = hip.load(); d.h = C.c_void_p(self.L.svh_ctx(self.h))
        d.close = lambda: None
        return d

    def set_option(self, key, value):
        """implementation choices with identical results (block schedules, kernel variants, POA engine): svh_set_option / svt_set_option"""
        self._chk(self.L.svh_set_option(self.h, key.encode(), int(value)), "set_option(%s)" % key)

    def set_temp_dir(self, path):
        """the reference's `<out>/temp/` directory: when set, every stage writes its intermediate file(s) there in the reference's formats"""
        if path:
            os.makedirs(path, exist_ok=True)
        self.L.svh_set_temp_dir(self.h, path.encode() if path else None)

    def trace_dump(self):
        """SAVONT_TRACE=1: print and clear the host-side timers"""
        self.L.svh_trace_dump()

    def seconds(self, name):
        return self.L.svh_stage_seconds(self.h, name.encode())

    # ---- stages
    def set_reads(self, seq, qual, offsets, ids=None, file_idx=None):
        self._keep = (seq, qual, offsets, file_idx)
        idb = None if ids is None else ("\n".join(ids)).encode()
        self.n_reads = len(offsets) - 1
        self._chk(self.L.svh_set_reads(self.h, _p(seq), _p(qual), _p(offsets), self.n_reads, idb, _p(file_idx)), "set_reads")

    def load_fastx(self, paths):
        """C++ ingest (needletail record semantics): FASTA/FASTQ, gz or plain, one sample per file -> reads resident in HBM"""
        n = C.c_uint32()
        self._chk(self.L.svh_load_fastx(self.h, "\n".join(paths).encode(), C.byref(n)), "load_fastx")
        return n.value

    def write_outputs(self, out_dir, sample_names=("sample",), pooled=False):
        """final_asvs.fasta, feature-table.tsv, final_clusters.tsv (src/main.rs:153-199) after refine_asv_depths_with_em"""
        self._chk(self.L.svh_write_outputs(self.h, out_dir.encode(), "\n".join(sample_names).encode(), 1 if pooled else 0), "write_outputs")

    def run_asv(self):
        """the whole of `savont asv` on the resident reads (src/main.rs:49-152) in ONE library call (svh_run_asv); under a communicator
        (set_shard_comm) or an exchange hook with world > 1 the library deals the stages out over the ranks and issues the exchanges itself"""
        self._chk(self.L.svh_run_asv(self.h), "run_asv")
        self.n_asvs = self.L.svh_consensus_count(self.h, 0)        # the ASV set is the final consensus set of this run (svh_consensus_to_asvs inside the call)
        return self.em_result()

    def set_shard_comm(self, rank, world, comm_id):
        """join the RCCL communicator of the 128 id bytes `comm_id` (hip.shard_comm_id() on one rank, handed to every rank by the caller)"""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(comm_id))
        self._chk(self.L.svh_set_shard_comm(self.h, int(rank), int(world), buf), "set_shard_comm")

    def repack(self):
        """K0 (2-bit pack) again from the ASCII bases kept in HBM (set_option("keep_ascii", 1) before set_reads)"""
        self._chk(self.L.svh_repack(self.h), "repack")

    def read_to_split_kmers(self, fetch=True):
        """Stage 1a.  The sorted table stays in HBM (Stage 1b reads two short selections of it); fetch=True also copies it
        out -- the B1 return value -- fetch=False returns (n_distinct, n_kept) only."""
        self._chk(self.L.svh_read_to_split_kmers(self.h), "read_to_split_kmers")
        n = self.L.svh_count_size(self.h)
        if not fetch:
            return self.L.svh_count_distinct(self.h), n
        km = np.zeros(n, np.uint64); rev = np.zeros(n, np.uint32); fwd = np.zeros(n, np.uint32)
        self._chk(self.L.svh_count_fetch(self.h, _p(km), _p(rev), _p(fwd)), "count_fetch")
        return self.L.svh_count_distinct(self.h), km, rev, fwd

    def set_count_table(self, km, rev, fwd):
        self.L.svh_set_count_table(self.h, _p(km), _p(rev), _p(fwd), len(km))

    def get_snpmers_inplace_sort(self):
        self._chk(self.L.svh_get_snpmers(self.h), "get_snpmers_inplace_sort")
        return self.snpmers()

    def set_snpmers(self, split, mid0, mid1, high_freq):
        self.L.svh_set_snpmers(self.h, _p(split), _p(mid0), _p(mid1), len(split), _p(high_freq), len(high_freq))

    def twin_reads_from_snpmers(self, fetch=True):
        """Stage 1c.  The twin reads stay in the pipeline (host + HBM); fetch=True also copies their metadata into numpy arrays."""
        self._chk(self.L.svh_twin_reads(self.h), "twin_reads_from_snpmers")
        if not fetch:
            return dict(n=self.L.svh_twin_count(self.h), auto_low_poly=bool(self.L.svh_auto_low_polymorphism(self.h)))
        return self.twin_meta()

    def twin_meta(self):
        """metadata of the twin reads the pipeline holds (no stage is re-run)"""
        n = self.L.svh_twin_count(self.h)
        r = dict(n=n, orig=np.zeros(n, np.uint32), length=np.zeros(n, np.uint32), est_id=np.zeros(n, np.float64), est_valid=np.zeros(n, np.uint8),
                 n_mini=np.zeros(n, np.uint32), n_unique=np.zeros(n, np.uint32), n_snp_kept=np.zeros(n, np.uint32),
                 lsh=np.zeros((n, 20), np.uint64), lsh_valid=np.zeros(n, np.uint8))
        self.L.svh_twin_meta(self.h, _p(r["orig"]), _p(r["length"]), _p(r["est_id"]), _p(r["est_valid"]), _p(r["n_mini"]), _p(r["n_unique"]),
                             _p(r["n_snp_kept"]), _p(r["lsh"]), _p(r["lsh_valid"]))
        r["auto_low_poly"] = bool(self.L.svh_auto_low_polymorphism(self.h))
        return r

    def snpmers(self):
        """the SNPmers / high-frequency k-mers the pipeline holds (no stage is re-run)"""
        n = self.L.svh_snpmer_count(self.h)
        sp = np.zeros(n, np.uint64); m0 = np.zeros(n, np.uint8); m1 = np.zeros(n, np.uint8); c0 = np.zeros(n, np.uint32); c1 = np.zeros(n, np.uint32)
        self.L.svh_snpmer_fetch(self.h, _p(sp), _p(m0), _p(m1), _p(c0), _p(c1))
        hf = np.zeros(self.L.svh_high_freq_count(self.h), np.uint64)
        self.L.svh_high_freq_fetch(self.h, _p(hf))
        return dict(split=sp, mid0=m0, mid1=m1, cnt0=c0, cnt1=c1, high_freq=hf, thresh=self.L.svh_high_freq_thresh(self.h))

    def kmer_clusters(self):
        return self._clusters(0)[0]

    def snpmer_clusters(self):
        return self._clusters(1)[0]

    def _clusters(self, which):
        n = self.L.svh_cluster_count(self.h, which)
        off = np.zeros(n + 1, np.uint64); mem = np.zeros(self.L.svh_cluster_total(self.h, which), np.uint32); grp = np.zeros(n, np.uint32)
        self.L.svh_clusters_fetch(self.h, which, _p(off), _p(mem), _p(grp))
        return [mem[int(off[i]):int(off[i + 1])].copy() for i in range(n)], grp

    def cluster_reads_by_kmers(self, fetch=True):
        self._chk(self.L.svh_cluster_reads_by_kmers(self.h), "cluster_reads_by_kmers")
        return self._clusters(0)[0] if fetch else self.L.svh_cluster_count(self.h, 0)

    def cluster_reads_by_snpmers(self, fetch=True):
        """fetch=False: only the number of clusters (they stay in the pipeline for Stage 4)"""
        self._chk(self.L.svh_cluster_reads_by_snpmers(self.h), "cluster_reads_by_snpmers")
        return self._clusters(1)[0] if fetch else self.L.svh_cluster_count(self.h, 1)

    def snpmer_pre_clusters(self):
        return self._clusters(2)

    def _consensus_set(self, which):
        n = self.L.svh_consensus_count(self.h, which)
        seq = np.zeros(max(1, self.L.svh_consensus_bases(self.h, which)), np.uint8); off = np.zeros(n + 1, np.uint64)
        depth = np.zeros(n, np.uint64); cid = np.zeros(n, np.uint64); nl = np.zeros(n, np.uint32)
        self.L.svh_consensus_fetch(self.h, which, _p(seq), _p(off), _p(depth), _p(cid), _p(nl))
        seqs = [seq[int(off[i]):int(off[i + 1])].tobytes() for i in range(n)]
        return dict(seqs=seqs, depth=depth, id=cid, n_low_quality=nl)

    def consensus(self, which=1):
        """Stage 4 (src/main.rs:84-110): align_and_consensus + generate_consensus_pileups + estimate_quality_error_rates +
        analyze_pileup_consensuses + decompress.  Returns (kept, low_quality) consensus sets."""
        self._chk(self.L.svh_consensus(self.h, which), "consensus")
        return self._consensus_set(0), self._consensus_set(1)

    def keep_pileups(self, keep=True):
        """test hook: keep the Stage-4 pile-ups and pre-analysis consensuses for raw_consensuses()"""
        self.L.svh_keep_pileups(self.h, 1 if keep else 0)

    def raw_consensuses(self):
        """-> list of dict(seq, depth, id, col_off, kind, base, qual, hp): POA consensus + its pile-up (entries in push order; hp = run length of a Base entry under use_hpc)"""
        out = []
        for ci in range(self.L.svh_raw_consensus_count(self.h)):
            n = self.L.svh_raw_consensus_len(self.h, ci)
            seq = np.zeros(n, np.uint8); d = C.c_uint64(); i = C.c_uint64(); m = C.c_uint64()
            self.L.svh_raw_consensus_fetch(self.h, ci, _p(seq), C.byref(d), C.byref(i), C.byref(m))
            ne = self.L.svh_pileup_entries(self.h, ci)
            off = np.zeros(n + 1, np.uint64); kind = np.zeros(max(1, ne), np.uint8); base = np.zeros(max(1, ne), np.uint8); qual = np.zeros(max(1, ne), np.uint8)
            self.L.svh_pileup_fetch(self.h, ci, _p(off), _p(kind), _p(base), _p(qual))
            hp = np.zeros(max(1, ne), np.uint8)
            self.L.svh_pileup_fetch_hp(self.h, ci, _p(hp))
            out.append(dict(seq=seq.tobytes(), depth=d.value, id=i.value, col_off=off, kind=kind[:ne], base=base[:ne], qual=qual[:ne], hp=hp[:ne]))
        return out

    def merge_similar_consensuses(self):
        """Stage 5 (src/alignment.rs:1213-1517) on the Stage-4 result -> the merged consensus set"""
        self._chk(self.L.svh_merge_similar_consensuses(self.h), "merge_similar_consensuses")
        return self._consensus_set(0)

    def detect_chimeras(self):
        """Stage 6 (src/chimera.rs:37-269 + filter_chimeras) -> (consensus set without chimeras, debug ids of the removed ones)"""
        self._chk(self.L.svh_detect_chimeras(self.h), "detect_chimeras")
        ids = np.zeros(max(1, self.L.svh_chimera_count(self.h)), np.uint32)
        self.L.svh_chimera_fetch(self.h, _p(ids))
        return self._consensus_set(0), ids[:self.L.svh_chimera_count(self.h)]

    @staticmethod
    def _flat(seqs, quals):
        off = np.zeros(len(seqs) + 1, np.uint64); off[1:] = np.cumsum([len(x) for x in seqs])
        seq = np.frombuffer(b"".join(seqs), np.uint8).copy()
        w = np.frombuffer(b"".join(quals), np.uint8).copy() if quals is not None else None
        return seq, w, off

    def poa_consensus_batch(self, clusters, use_gpu=True, engine=None, with_graph_size=False):
        """clusters: list of (seqs, quals|None) -> consensus per cluster.  engine: 0 host DP, 2 K12 (graphs resident on the
        device), 3 K12 for poa_device_share percent of the clusters; use_gpu=True/False is the old spelling of engine 2 / 0.  with_graph_size: -> (consensuses, nodes of every final graph)"""
        if engine is None:
            engine = 2 if use_gpu else 0
        seqs = [s for c in clusters for s in c[0]]
        quals = None if clusters[0][1] is None else [q for c in clusters for q in c[1]]
        seq, w, off = self._flat(seqs, quals)
        cl_off = np.zeros(len(clusters) + 1, np.uint64); cl_off[1:] = np.cumsum([len(c[0]) for c in clusters])
        cap = int(off[-1]) + 64
        out = np.zeros(cap, np.uint8); out_off = np.zeros(len(clusters) + 1, np.uint64)
        gn = np.zeros(len(clusters), np.uint64)
        self._chk(self.L.svh_poa_consensus_batch(self.h, int(engine), _p(seq), _p(w), _p(off), _p(cl_off), len(clusters), _p(out), _p(out_off), cap, _p(gn)), "poa_consensus_batch")
        cons = [out[int(out_off[i]):int(out_off[i + 1])].tobytes() for i in range(len(clusters))]
        return (cons, gn.tolist()) if with_graph_size else cons

    def quality_error_map(self):
        n = self.L.svh_quality_map(self.h, None, None)
        q = np.zeros(n, np.uint8); r = np.zeros(n, np.float64)
        self.L.svh_quality_map(self.h, _p(q), _p(r))
        return dict(zip(q.tolist(), r.tolist()))

    def consensus_to_asvs(self):
        self._chk(self.L.svh_consensus_to_asvs(self.h), "consensus_to_asvs")
        self.n_asvs = self.L.svh_consensus_count(self.h, 0)

    def set_asvs(self, seq, offsets):
        self._asv_keep = (seq, offsets)
        self.n_asvs = len(offsets) - 1
        self._chk(self.L.svh_set_asvs(self.h, _p(seq), _p(offsets), self.n_asvs), "set_asvs")

    def refine_asv_depths_with_em(self):
        self._chk(self.L.svh_refine_asv_depths_with_em(self.h), "refine_asv_depths_with_em")
        return self.em_result()

    def em_result(self):
        """the Stage-7 result the pipeline holds (no stage is re-run)"""
        n = self.n_asvs
        d = np.zeros(n, np.uint64); u = np.zeros(n, np.uint64); a = np.zeros(n, np.uint64); l = np.zeros(n, np.uint64)
        tot = C.c_uint64(); fil = C.c_uint64(); ko = C.c_int()
        self.L.svh_em_fetch(self.h, _p(d), _p(u), _p(a), _p(l), C.byref(tot), C.byref(fil), C.byref(ko))
        nt = self.L.svh_twin_count(self.h)
        nb = np.zeros(nt, np.uint32); nm = np.zeros(nt, np.int32); fa = np.zeros(nt, np.uint32)
        self.L.svh_em_read_assignments(self.h, _p(nb), _p(nm), _p(fa))
        return dict(rc=1 if ko.value else 0, depth=d, unambig=u, ambig=a, leq10=l, total=tot.value, filtered=fil.value, n_best=nb, best_nm=nm, first_asv=fa)

    def compute_per_sample_depths(self, n_samples):
        out = np.zeros((self.n_asvs, n_samples), np.uint64)
        self._chk(self.L.svh_compute_per_sample_depths(self.h, n_samples, _p(out)), "compute_per_sample_depths")
        return out


def synth_reads(hap_seq, hap_off, weights, n_reads, seed):
    """Deterministic synthetic ONT-like amplicon reads.  Host tooling, no GPU needed:
    only the generator symbol of libsavont_asv.so is used."""
    L = load()
    hap_seq = np.ascontiguousarray(hap_seq, np.uint8); hap_off = np.ascontiguousarray(hap_off, np.uint64)
    weights = np.ascontiguousarray(weights, np.float64)
    nh = len(hap_off) - 1
    maxlen = int((hap_off[1:] - hap_off[:-1]).max())
    cap = int(n_reads * (maxlen * 1.25 + 32))
    seq = np.zeros(cap, np.uint8); qual = np.zeros(cap, np.uint8); off = np.zeros(n_reads + 1, np.uint64)
    hap = np.zeros(n_reads, np.uint32); strand = np.zeros(n_reads, np.uint8)
    tot = L.svh_synth_reads(_p(hap_seq), _p(hap_off), nh, _p(weights), n_reads, seed, _p(seq), _p(qual), _p(off), _p(hap), _p(strand))
    return seq[:tot].copy(), qual[:tot].copy(), off, hap, strand


def poa_consensus(seqs, quals=None, with_graph_size=False, wide_cells=False, no_band=False):
    """generate_consensus_poa (src/alignment.rs:193-231) on the host; needs no GPU.  with_graph_size: -> (consensus, #graph nodes);
    wide_cells: the plain int32 DP instead of the SIMD 16-bit rows (same result); no_band: the hidden --no-band flag (unbanded DP, :198,217)"""
    L = load()
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    seq = np.frombuffer(b"".join(seqs), np.uint8).copy() if seqs else np.zeros(1, np.uint8)
    w = np.frombuffer(b"".join(quals), np.uint8).copy() if quals is not None else None
    cap = int(off[-1]) + 16
    out = np.zeros(cap, np.uint8)
    nodes = C.c_uint64()
    n = L.svh_poa_consensus(_p(seq), _p(w) if w is not None else None, _p(off), len(seqs), _p(out), cap, C.byref(nodes), (1 if wide_cells else 0) | (2 if no_band else 0))
    if n < 0:
        raise RuntimeError("svh_poa_consensus failed")
    return (out[:n].tobytes(), nodes.value) if with_graph_size else out[:n].tobytes()


def gunzip_digest(path, decoder):
    """(bytes, FNV-1a digest, seconds) of a .gz file inflated by zlib (decoder 0) or by the library's own decoder (1); ValueError on a corrupt file"""
    L = load()
    L.svh_gunzip_digest.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_char_p, C.c_uint64]
    n = C.c_uint64(); d = C.c_uint64(); sec = C.c_double(); err = C.create_string_buffer(512)
    if L.svh_gunzip_digest(str(path).encode(), int(decoder), C.byref(n), C.byref(d), C.byref(sec), err, 512) != 0:
        raise ValueError(err.value.decode())
    return n.value, d.value, sec.value


def fastx_digest(path):
    """C++ ingest of one FASTA/FASTQ (gz or plain) file -> (records, bases, has_qual, FNV-1a digest of ids/sequences/qualities); no GPU"""
    L = load()
    n = C.c_uint64(); b = C.c_uint64(); q = C.c_int(); d = C.c_uint64(); err = C.create_string_buffer(512)
    if L.svh_fastx_digest(path.encode(), C.byref(n), C.byref(b), C.byref(q), C.byref(d), err, 512) != 0:
        raise ValueError(err.value.decode())
    return n.value, b.value, bool(q.value), d.value


def minimizer_seeds(seq, w=10, k=21):
    """seeding::minimizer_seeds_positions (k-mer values) on the host; needs no GPU."""
    L = load()
    seq = np.ascontiguousarray(np.frombuffer(seq, np.uint8) if isinstance(seq, (bytes, bytearray)) else seq, np.uint8)
    out = np.zeros(len(seq) + 1, np.uint64)
    n = L.svh_minimizer_seeds(_p(seq), len(seq), w, k, _p(out), len(out))
    return out[:n].copy()


def snpmers_from_table(km, rev, fwd, k=17, single_strand=False):
    """Host SNPmer calling (kmer_comp::get_snpmers_inplace_sort) on a count table; needs no GPU."""
    L = load()
    km = np.ascontiguousarray(km, np.uint64); rev = np.ascontiguousarray(rev, np.uint32); fwd = np.ascontiguousarray(fwd, np.uint32)
    n = len(km)
    sp = np.zeros(n, np.uint64); m0 = np.zeros(n, np.uint8); m1 = np.zeros(n, np.uint8); c0 = np.zeros(n, np.uint32); c1 = np.zeros(n, np.uint32)
    hf = np.zeros(n, np.uint64); nhf = C.c_uint32(); th = C.c_uint32()
    ns = L.svh_snpmers_from_table(_p(km), _p(rev), _p(fwd), n, k, int(single_strand), _p(sp), _p(m0), _p(m1), _p(c0), _p(c1), _p(hf), C.byref(nhf), C.byref(th))
    if ns < 0:
        raise hip.SavontHipError("svh_snpmers_from_table failed")
    return dict(split=sp[:ns], mid0=m0[:ns], mid1=m1[:ns], cnt0=c0[:ns], cnt1=c1[:ns], high_freq=hf[:nhf.value], thresh=th.value)
