"""ctypes binding of the CPU oracle (oracle/libsavont_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product package (savont_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libsavont_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [
        ("k", C.c_uint32), ("c", C.c_uint32), ("min_read_length", C.c_uint32), ("max_read_length", C.c_uint32),
        ("quality_value_cutoff", C.c_double), ("minimum_base_quality", C.c_uint32), ("single_strand", C.c_uint32),
        ("min_cluster_size", C.c_uint32), ("max_iterations_recluster", C.c_uint32),
        ("primary_clustering_threshold", C.c_double), ("align_band", C.c_uint32), ("threads", C.c_uint32), ("low_polymorphism", C.c_uint32), ("no_snpmers", C.c_uint32), ("no_band", C.c_uint32), ("nm_contract", C.c_uint32),
    ]


def build(force=False):
    src = os.path.join(ORACLE_DIR, "savont_oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, u8p, u32p, u64p, i32p, dp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.POINTER(OrcParams)]
        L.orc_destroy.argtypes = [vp]
        L.orc_last_error.restype = C.c_char_p
        L.orc_last_error.argtypes = [vp]
        L.orc_last_stage_seconds.restype = C.c_double
        L.orc_last_stage_seconds.argtypes = [vp]
        L.orc_default_params.argtypes = [C.POINTER(OrcParams)]
        L.orc_mm_hash64.restype = C.c_uint64
        L.orc_mm_hash64.argtypes = [C.c_uint64]
        L.orc_fx_hash_pair.restype = C.c_uint64
        L.orc_fx_hash_pair.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_byte_to_seq.restype = C.c_uint8
        L.orc_byte_to_seq.argtypes = [C.c_uint8]
        L.orc_qual_bin.restype = C.c_uint8
        L.orc_qual_bin.argtypes = [C.c_uint8]
        L.orc_pack_2bit.argtypes = [u8p, C.c_uint64, u32p]
        L.orc_kmer_from_ascii.restype = C.c_uint64
        L.orc_kmer_from_ascii.argtypes = [C.c_char_p, C.c_uint32]
        L.orc_revcomp_kmer.restype = C.c_uint64
        L.orc_revcomp_kmer.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_reverse_complement.argtypes = [u8p, C.c_uint64, u8p]
        L.orc_kmer_from_position.restype = C.c_uint64
        L.orc_kmer_from_position.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_split_kmer_mid.restype = C.c_uint64
        L.orc_split_kmer_mid.argtypes = [u8p, u8p, C.c_uint64, C.c_uint32, C.c_uint8, u64p]
        L.orc_estimate_identity.restype = C.c_double
        L.orc_estimate_identity.argtypes = [u8p, C.c_uint64, C.POINTER(C.c_int)]
        L.orc_binomial_test.restype = C.c_double
        L.orc_binomial_test.argtypes = [C.c_uint64, C.c_uint64, C.c_double]
        L.orc_fisher_two_tail.restype = C.c_double
        L.orc_fisher_two_tail.argtypes = [C.c_uint32] * 4
        L.orc_lsh_signatures.argtypes = [u64p, C.c_uint32, u64p, u8p]
        L.orc_band_for.restype = C.c_int32
        L.orc_band_for.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_align_nm.restype = C.c_int32
        L.orc_align_nm.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_int, C.c_uint32]
        L.orc_align_nm_affine.restype = C.c_int32
        L.orc_align_nm_affine.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_int, C.c_uint32, i32p]
        L.orc_align_nm_affine_near.restype = C.c_int32
        L.orc_align_nm_affine_near.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_int, C.c_uint32, i32p]
        L.orc_align_pileup_row.restype = C.c_int32
        L.orc_align_pileup_row.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, u8p, C.c_int, C.c_uint32, u64p, u32p]
        L.orc_align_pileup_row_tags.restype = C.c_int32
        L.orc_align_pileup_row_tags.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, u8p, u8p, C.c_int, C.c_uint32, u64p, u32p]
        L.orc_hpc_qual.restype = C.c_uint64
        L.orc_hpc_qual.argtypes = [u8p, u8p, C.c_uint64, u8p, u8p, u8p]
        L.orc_strand_vote.restype = None
        L.orc_strand_vote.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_hpc.restype = C.c_uint64
        L.orc_hpc.argtypes = [u8p, C.c_uint64, u8p, u8p]
        L.orc_set_reads.argtypes = [vp, u8p, u8p, u64p, C.c_uint32, C.c_char_p, u32p]
        for name in ("orc_count_split_kmers", "orc_get_snpmers", "orc_twin_reads", "orc_cluster_by_kmers",
                     "orc_cluster_by_snpmers", "orc_refine_depths_em", "orc_auto_low_polymorphism"):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = C.c_int
        for name in ("orc_count_raw_distinct", "orc_count_size", "orc_kmer_cluster_total", "orc_snpmer_cluster_total",
                     "orc_snpmer_pre_cluster_total", "orc_em_total_assigned", "orc_em_filtered"):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = C.c_uint64
        for name in ("orc_snpmer_count", "orc_high_freq_thresh", "orc_high_freq_count", "orc_twin_count",
                     "orc_kmer_cluster_count", "orc_snpmer_cluster_count", "orc_snpmer_pre_cluster_count"):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = C.c_uint32
        L.orc_count_fetch.argtypes = [vp, u64p, u32p, u32p]
        L.orc_snpmer_fetch.argtypes = [vp, u64p, u8p, u8p, u32p, u32p]
        L.orc_high_freq_fetch.argtypes = [vp, u64p]
        L.orc_set_snpmers.argtypes = [vp, u64p, u8p, u8p, C.c_uint32, u64p, C.c_uint32]
        L.orc_twin_meta.argtypes = [vp, u32p, u32p, dp, u8p, u32p, u32p, u32p, u32p]
        L.orc_twin_minimizers.argtypes = [vp, u32p, u64p, u8p]
        L.orc_twin_snpmers.argtypes = [vp, u32p, u64p, u8p]
        L.orc_twin_lsh.argtypes = [vp, u64p, u8p]
        L.orc_read_qual_bins.restype = C.c_uint64
        L.orc_read_qual_bins.argtypes = [vp, C.c_uint32, u8p]
        L.orc_read_seeds.argtypes = [vp, C.c_uint32, u32p, u32p, u64p, u32p, u32p, u64p]
        L.orc_kmer_clusters_fetch.argtypes = [vp, u64p, u32p]
        L.orc_snpmer_clusters_fetch.argtypes = [vp, u64p, u32p]
        L.orc_snpmer_pre_clusters_fetch.argtypes = [vp, u64p, u32p, u32p]
        L.orc_set_asvs.argtypes = [vp, u8p, u64p, C.c_uint32]
        L.orc_em_fetch.argtypes = [vp, u64p, u64p, u64p, u64p]
        L.orc_em_read_assignments.argtypes = [vp, u32p, i32p, u32p]
        L.orc_per_sample_depths.argtypes = [vp, C.c_uint32, u64p]
        L.orc_set_count_table.argtypes = [vp, u64p, u32p, u32p, C.c_uint64, C.c_uint64]
        L.orc_em_read_classes.argtypes = [vp, u64p, u32p]
        L.orc_em_read_classes.restype = C.c_uint64
        # stages 4-6 as one chain (oracle/stage456_oracle.inc)
        L.orc_stage456_run.restype = vp
        L.orc_stage456_run.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double]
        L.orc_stage456_free.argtypes = [vp]
        L.orc_stage456_count.restype = C.c_uint32
        L.orc_stage456_count.argtypes = [vp, C.c_int]
        L.orc_stage456_bases.restype = C.c_uint64
        L.orc_stage456_bases.argtypes = [vp, C.c_int, C.c_int]
        L.orc_stage456_fetch.argtypes = [vp, C.c_int, C.c_int, u8p, u64p, u64p, u32p]
        L.orc_stage456_quality_map.restype = C.c_uint32
        L.orc_stage456_quality_map.argtypes = [vp, u8p, dp]
        L.orc_stage456_chimeras.restype = C.c_uint32
        L.orc_stage456_chimeras.argtypes = [vp, u32p]
        L.orc_stage456_seconds.argtypes = [vp, dp]
        L.orc_poa_consensus.restype = C.c_uint64
        L.orc_poa_consensus.argtypes = [u8p, u8p, u64p, C.c_uint32, u8p, C.c_uint64, u64p]
        L.orc_poa_consensus2.restype = C.c_uint64
        L.orc_poa_consensus2.argtypes = [u8p, u8p, u64p, C.c_uint32, u8p, C.c_uint64, u64p, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_params(**kw):
    p = OrcParams()
    lib().orc_default_params(C.byref(p))
    known = {f[0] for f in OrcParams._fields_}
    for k, v in kw.items():
        if k not in known:
            raise TypeError("unknown oracle parameter %r" % k)
        setattr(p, k, v)
    return p


class Oracle:
    """Stateful pipeline oracle.  Reads are given as (seq u8[total], qual u8[total] | None, offsets u64[n+1])."""

    def __init__(self, **params):
        self.L = lib()
        self.params = default_params(**params)
        self.h = self.L.orc_create(C.byref(self.params))

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def seconds(self):
        return self.L.orc_last_stage_seconds(self.h)

    def set_reads(self, seq, qual, offsets, ids=None, file_idx=None):
        self._keep = (seq, qual, offsets, file_idx)
        idb = None if ids is None else ("\n".join(ids)).encode()
        self.n_reads = len(offsets) - 1
        self.L.orc_set_reads(self.h, _p(seq), _p(qual), _p(offsets), self.n_reads, idb, _p(file_idx))

    # ---- stage 1
    def count_split_kmers(self):
        rc = self.L.orc_count_split_kmers(self.h)
        n = self.L.orc_count_size(self.h)
        km = np.zeros(n, np.uint64); rev = np.zeros(n, np.uint32); fwd = np.zeros(n, np.uint32)
        self.L.orc_count_fetch(self.h, _p(km), _p(rev), _p(fwd))
        return rc, self.L.orc_count_raw_distinct(self.h), km, rev, fwd

    def set_count_table(self, km, rev, fwd, raw_distinct):
        km = np.ascontiguousarray(km, np.uint64); rev = np.ascontiguousarray(rev, np.uint32); fwd = np.ascontiguousarray(fwd, np.uint32)
        self.L.orc_set_count_table(self.h, _p(km), _p(rev), _p(fwd), len(km), int(raw_distinct))

    def em_read_classes(self):
        """-> (off u64[n_twin + 1], members u32[]) of the last refine_depths_em"""
        n = self.L.orc_twin_count(self.h)
        off = np.zeros(n + 1, np.uint64)
        m = self.L.orc_em_read_classes(self.h, _p(off), None)
        mem = np.zeros(max(1, m), np.uint32)
        self.L.orc_em_read_classes(self.h, _p(off), _p(mem))
        return off, mem[:m]

    def get_snpmers(self):
        rc = self.L.orc_get_snpmers(self.h)
        n = self.L.orc_snpmer_count(self.h)
        sp = np.zeros(n, np.uint64); m0 = np.zeros(n, np.uint8); m1 = np.zeros(n, np.uint8)
        c0 = np.zeros(n, np.uint32); c1 = np.zeros(n, np.uint32)
        self.L.orc_snpmer_fetch(self.h, _p(sp), _p(m0), _p(m1), _p(c0), _p(c1))
        nh = self.L.orc_high_freq_count(self.h)
        hf = np.zeros(nh, np.uint64)
        self.L.orc_high_freq_fetch(self.h, _p(hf))
        return dict(rc=rc, split=sp, mid0=m0, mid1=m1, cnt0=c0, cnt1=c1, high_freq=hf,
                    thresh=self.L.orc_high_freq_thresh(self.h))

    def set_snpmers(self, split, mid0, mid1, high_freq):
        self.L.orc_set_snpmers(self.h, _p(split), _p(mid0), _p(mid1), len(split), _p(high_freq), len(high_freq))

    def twin_reads(self):
        self.L.orc_twin_reads(self.h)
        n = self.L.orc_twin_count(self.h)
        orig = np.zeros(n, np.uint32); ln = np.zeros(n, np.uint32); est = np.zeros(n, np.float64)
        ev = np.zeros(n, np.uint8); nm = np.zeros(n, np.uint32); nmk = np.zeros(n, np.uint32)
        ns = np.zeros(n, np.uint32); nsk = np.zeros(n, np.uint32)
        self.L.orc_twin_meta(self.h, _p(orig), _p(ln), _p(est), _p(ev), _p(nm), _p(nmk), _p(ns), _p(nsk))
        tm = int(nm.sum()); ts = int(ns.sum())
        mp = np.zeros(tm, np.uint32); mk = np.zeros(tm, np.uint64); mkeep = np.zeros(tm, np.uint8)
        sp = np.zeros(ts, np.uint32); sk = np.zeros(ts, np.uint64); skeep = np.zeros(ts, np.uint8)
        self.L.orc_twin_minimizers(self.h, _p(mp), _p(mk), _p(mkeep))
        self.L.orc_twin_snpmers(self.h, _p(sp), _p(sk), _p(skeep))
        sig = np.zeros((n, 20), np.uint64); val = np.zeros((n, 20), np.uint8)
        self.L.orc_twin_lsh(self.h, _p(sig), _p(val))
        return dict(n=n, orig=orig, length=ln, est_id=est, est_valid=ev, n_mini=nm, n_mini_kept=nmk, n_snp=ns,
                    n_snp_kept=nsk, mini_pos=mp, mini_kmer=mk, mini_kept=mkeep, snp_pos=sp, snp_kmer=sk,
                    snp_kept=skeep, lsh=sig, lsh_valid=val, auto_low_poly=bool(self.L.orc_auto_low_polymorphism(self.h)))

    def read_seeds(self, orig):
        nm = C.c_uint32(); ns = C.c_uint32()
        self.L.orc_read_seeds(self.h, orig, C.byref(nm), None, None, C.byref(ns), None, None)
        mp = np.zeros(nm.value, np.uint32); mk = np.zeros(nm.value, np.uint64)
        sp = np.zeros(ns.value, np.uint32); sk = np.zeros(ns.value, np.uint64)
        self.L.orc_read_seeds(self.h, orig, C.byref(nm), _p(mp), _p(mk), C.byref(ns), _p(sp), _p(sk))
        return mp, mk, sp, sk

    def qual_bins(self, orig):
        n = self.L.orc_read_qual_bins(self.h, orig, None)
        b = np.zeros(n, np.uint8)
        self.L.orc_read_qual_bins(self.h, orig, _p(b))
        return b

    @staticmethod
    def _clusters(count, total, fetch):
        off = np.zeros(count + 1, np.uint64); mem = np.zeros(total, np.uint32)
        fetch(off, mem)
        return [mem[int(off[i]):int(off[i + 1])].copy() for i in range(count)]

    def cluster_by_kmers(self):
        self.L.orc_cluster_by_kmers(self.h)
        return self._clusters(self.L.orc_kmer_cluster_count(self.h), self.L.orc_kmer_cluster_total(self.h),
                              lambda o, m: self.L.orc_kmer_clusters_fetch(self.h, _p(o), _p(m)))

    def cluster_by_snpmers(self):
        self.L.orc_cluster_by_snpmers(self.h)
        return self._clusters(self.L.orc_snpmer_cluster_count(self.h), self.L.orc_snpmer_cluster_total(self.h),
                              lambda o, m: self.L.orc_snpmer_clusters_fetch(self.h, _p(o), _p(m)))

    def snpmer_pre_clusters(self):
        n = self.L.orc_snpmer_pre_cluster_count(self.h)
        grp = np.zeros(n, np.uint32)
        cl = self._clusters(n, self.L.orc_snpmer_pre_cluster_total(self.h),
                            lambda o, m: self.L.orc_snpmer_pre_clusters_fetch(self.h, _p(o), _p(m), _p(grp)))
        return cl, grp

    def set_asvs(self, seq, offsets):
        self._asv_keep = (seq, offsets)
        self.n_asvs = len(offsets) - 1
        self.L.orc_set_asvs(self.h, _p(seq), _p(offsets), self.n_asvs)

    def refine_depths_em(self):
        rc = self.L.orc_refine_depths_em(self.h)
        n = self.n_asvs
        d = np.zeros(n, np.uint64); u = np.zeros(n, np.uint64); a = np.zeros(n, np.uint64); l = np.zeros(n, np.uint64)
        self.L.orc_em_fetch(self.h, _p(d), _p(u), _p(a), _p(l))
        nt = self.L.orc_twin_count(self.h)
        nb = np.zeros(nt, np.uint32); nm = np.zeros(nt, np.int32); fa = np.zeros(nt, np.uint32)
        self.L.orc_em_read_assignments(self.h, _p(nb), _p(nm), _p(fa))
        return dict(rc=rc, depth=d, unambig=u, ambig=a, leq10=l, total=self.L.orc_em_total_assigned(self.h),
                    filtered=self.L.orc_em_filtered(self.h), n_best=nb, best_nm=nm, first_asv=fa)

    def stage456(self, chimera_allowable_errors=1, chimera_detect_length=0, mask_low_quality=False, n_depth_cutoff=250, posterior_threshold_ln=30.0):
        """stages 4-6 on the SNPmer clusters of this oracle (after cluster_by_snpmers): POA consensus, pile-ups, Bayesian polish, merge, chimera
        filter (oracle/stage456_oracle.inc) -> dict of sets raw / kept / low / merged / final, each dict(seqs, decompressed, depth, id), + qmap,
        chimera_ids, seconds (POA, pile-ups, statistics, merge, chimera)"""
        h = self.L.orc_stage456_run(self.h, chimera_allowable_errors, chimera_detect_length, 1 if mask_low_quality else 0, n_depth_cutoff, posterior_threshold_ln)
        if not h:
            raise RuntimeError("orc_stage456_run: " + self.L.orc_last_error(self.h).decode())
        out = {}
        try:
            for si, name in enumerate(("raw", "kept", "low", "merged", "final")):
                n = self.L.orc_stage456_count(h, si)
                d = {}
                for dec, key in ((0, "seqs"), (1, "decompressed")):
                    buf = np.zeros(max(1, self.L.orc_stage456_bases(h, si, dec)), np.uint8); off = np.zeros(n + 1, np.uint64); depth = np.zeros(n, np.uint64); idv = np.zeros(n, np.uint32)
                    self.L.orc_stage456_fetch(h, si, dec, _p(buf), _p(off), _p(depth), _p(idv))
                    d[key] = [buf[int(off[i]):int(off[i + 1])].tobytes() for i in range(n)]
                    d["depth"] = depth; d["id"] = idv
                out[name] = d
            nq = self.L.orc_stage456_quality_map(h, None, None)
            q = np.zeros(nq, np.uint8); r = np.zeros(nq, np.float64)
            self.L.orc_stage456_quality_map(h, _p(q), _p(r))
            out["qmap"] = dict(zip(q.tolist(), r.tolist()))
            ncx = self.L.orc_stage456_chimeras(h, None); ids = np.zeros(max(1, ncx), np.uint32); self.L.orc_stage456_chimeras(h, _p(ids)); out["chimera_ids"] = ids[:ncx]
            sec = np.zeros(5, np.float64); self.L.orc_stage456_seconds(h, _p(sec)); out["seconds"] = dict(zip(("poa", "pileups", "statistics", "merge", "chimera"), sec.tolist()))
        finally:
            self.L.orc_stage456_free(h)
        return out

    def final_asvs(self, s456=None, **kw):
        """stages 4-7 of the oracle -> the final list of src/main.rs:140-152: (sequence, depth) of the ASVs whose EM depth is not 0, stable-sorted
        by depth descending; also returns the EM result and the stage456 dict"""
        s456 = s456 or self.stage456(**kw)
        seqs = s456["final"]["decompressed"]
        if not seqs:
            return [], None, s456
        cat = np.frombuffer(b"".join(seqs), np.uint8); off = np.zeros(len(seqs) + 1, np.uint64); off[1:] = np.cumsum([len(x) for x in seqs])
        self.set_asvs(cat, off)
        em = self.refine_depths_em()
        lst = [(seqs[i], int(em["depth"][i])) for i in range(len(seqs)) if int(em["depth"][i]) > 0]
        lst.sort(key=lambda x: -x[1])
        return lst, em, s456

    def per_sample_depths(self, n_samples):
        out = np.zeros((self.n_asvs, n_samples), np.uint64)
        self.L.orc_per_sample_depths(self.h, n_samples, _p(out))
        return out


# ---- stateless leaf wrappers ---------------------------------------------------------------------
def poa_consensus(seqs, quals=None, no_band=False):
    """generate_consensus_poa, the C++ twin of oracle/poa_oracle.py -> (consensus bytes, graph nodes); no_band: the hidden --no-band flag (unbanded DP)"""
    L = lib()
    cat = np.frombuffer(b"".join(seqs), np.uint8) if seqs else np.zeros(0, np.uint8)
    w = None if quals is None else np.frombuffer(b"".join(quals), np.uint8)
    off = np.zeros(len(seqs) + 1, np.uint64); off[1:] = np.cumsum([len(x) for x in seqs])
    cap = int(off[-1]) + 64
    out = np.zeros(cap, np.uint8); nodes = C.c_uint64(0)
    n = L.orc_poa_consensus2(_p(cat if len(cat) else np.zeros(1, np.uint8)), _p(w), _p(off), len(seqs), _p(out), cap, C.byref(nodes), 1 if no_band else 0)
    return out[:n].tobytes(), nodes.value


def split_kmer_mid(seq, qual, k, min_bq):
    L = lib()
    out = np.zeros(max(len(seq), 1), np.uint64)
    n = L.orc_split_kmer_mid(_p(seq), _p(qual), len(seq), k, min_bq, _p(out))
    return out[:n].copy()


def kmer_from_position(seq, pos, k):
    seq = np.ascontiguousarray(seq, np.uint8)
    return lib().orc_kmer_from_position(_p(seq), len(seq), int(pos), int(k))


def pack_2bit(seq):
    w = np.zeros((len(seq) + 15) // 16, np.uint32)
    lib().orc_pack_2bit(_p(seq), len(seq), _p(w))
    return w


def reverse_complement(seq):
    out = np.zeros(len(seq), np.uint8)
    lib().orc_reverse_complement(_p(seq), len(seq), _p(out))
    return out


def estimate_identity(qual):
    v = C.c_int()
    e = lib().orc_estimate_identity(_p(qual), len(qual), C.byref(v))
    return e, bool(v.value)


def lsh_signatures(kmers):
    kmers = np.ascontiguousarray(kmers, np.uint64)
    sig = np.zeros(20, np.uint64); val = np.zeros(20, np.uint8)
    lib().orc_lsh_signatures(_p(kmers), len(kmers), _p(sig), _p(val))
    return sig, val


def align_nm(q, t, reverse, band):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    return lib().orc_align_nm(_p(q), len(q), _p(t), len(t), int(reverse), int(band))


def align_nm_affine(q, t, reverse, band):
    """minimap2-style nm (K8a): -> dict(nm, score, q_end, t_end, n_max) or None when nothing aligns"""
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    out = np.zeros(5, np.int32)
    nm = lib().orc_align_nm_affine(_p(q), len(q), _p(t), len(t), int(reverse), int(band), _p(out))
    return None if nm < 0 else dict(nm=int(out[0]), score=int(out[1]), q_end=int(out[2]), t_end=int(out[3]), n_max=int(out[4]))


def align_nm_affine_near(q, t, reverse, band):
    """K8a inside the band around the unit-cost optimum (nm_contract 1): -> dict(nm (None when nothing aligns), score, band, d, end_diag)"""
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    out = np.zeros(8, np.int32)
    nm = lib().orc_align_nm_affine_near(_p(q), len(q), _p(t), len(t), int(reverse), int(band), _p(out))
    return dict(nm=None if nm < 0 else int(out[0]), score=int(out[1]) if nm >= 0 else 0, band=int(out[5]), d=int(out[6]), end_diag=int(out[7]))


def primary_hit_nm(query, refs, slack=8):
    """the acceptance criterion of tests/integration_test.rs:116-158 restated: map `query` to the references on both strands
    with the minimap2-style local affine alignment (K8a); the primary hit is the best-scoring one (ties: fewest nm)
    -> (nm, score, ref).  The unit-cost overlap distance (cheap) preselects the references within `slack` of the closest."""
    query = np.ascontiguousarray(query, np.uint8)
    ed = []
    for ri, r in enumerate(refs):
        for rev in (0, 1):
            d = align_nm(r, query, rev, band_for(len(r), len(query)))
            if d >= 0:
                ed.append((d, ri, rev))
    if not ed:
        return None
    lo = min(ed)[0]
    best = None
    for d, ri, rev in ed:
        if d <= lo + slack:
            a = align_nm_affine(refs[ri], query, rev, band_for(len(refs[ri]), len(query)))
            if a is not None and (best is None or (a["score"], -a["nm"]) > (best[1], -best[0])):
                best = (a["nm"], a["score"], ri)
    return best


def band_for(n, m):
    return lib().orc_band_for(int(n), int(m))


def hpc(seq):
    seq = np.ascontiguousarray(seq, np.uint8)
    o = np.zeros(len(seq), np.uint8); l = np.zeros(len(seq), np.uint8)
    n = lib().orc_hpc(_p(seq), len(seq), _p(o), _p(l))
    return o[:n].copy(), l[:n].copy()


def hpc_qual(seq, qual):
    """src/utils.rs:136-190 -> (hpc_seq, min quality per run, run lengths)"""
    seq = np.ascontiguousarray(seq, np.uint8); qual = np.ascontiguousarray(qual, np.uint8)
    o = np.zeros(len(seq), np.uint8); q = np.zeros(len(seq), np.uint8); l = np.zeros(len(seq), np.uint8)
    n = lib().orc_hpc_qual(_p(seq), _p(qual), len(seq), _p(o), _p(q), _p(l))
    return o[:n].copy(), q[:n].copy(), l[:n].copy()


def align_pileup_row_tags(q, t, qual, hp, reverse, band):
    """K9 row of a homopolymer-compressed read t (per-base quality and run length) on consensus q -> (nm, cells, span); hp in bits 56-63"""
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    qual = np.ascontiguousarray(qual, np.uint8); hp = np.ascontiguousarray(hp, np.uint8)
    cells = np.zeros(len(q), np.uint64); span = np.zeros(4, np.uint32)
    nm = lib().orc_align_pileup_row_tags(_p(q), len(q), _p(t), len(t), _p(qual), _p(hp), int(reverse), int(band), _p(cells), _p(span))
    return nm, cells, span


def strand_vote(a, b, k=17, c=11):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    sh = C.c_uint32(); sm = C.c_uint32()
    lib().orc_strand_vote(_p(a), len(a), _p(b), len(b), k, c, C.byref(sh), C.byref(sm))
    return sh.value, sm.value


def align_pileup_row(q, t, bins, reverse, band):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    bins = None if bins is None else np.ascontiguousarray(bins, np.uint8)
    cells = np.zeros(len(q), np.uint64); span = np.zeros(4, np.uint32)
    nm = lib().orc_align_pileup_row(_p(q), len(q), _p(t), len(t), _p(bins), int(reverse), int(band), _p(cells), _p(span))
    return nm, cells, span
