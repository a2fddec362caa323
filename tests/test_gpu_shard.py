"""svt_set_shard on one GPU: `world` pipelines of this process, one thread each, stand in for the ranks of a multi-GPU job (savont_amd/shard.py
LocalExchange copies the slices device-to-device where RCCL would broadcast them).  Every rank runs only its slice of the K5 pairs, of the K6
row tiles and of the K3/K4 reads; the exchanges must leave every rank with exactly what the unsharded pipeline computes, stage by stage."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stages(p, out):
    nd, nk = p.read_to_split_kmers(fetch=False)
    p.get_snpmers_inplace_sort()
    tw = p.twin_reads_from_snpmers(fetch=False)
    p.cluster_reads_by_kmers(fetch=False); p.cluster_reads_by_snpmers(fetch=False)
    out.update(n_kept=nk, snp=p.snpmers()["split"], tw=p.twin_meta(), kc=p.kmer_clusters(), sc=p.snpmer_clusters())


def _same(a, b):
    assert a["n_kept"] == b["n_kept"] and np.array_equal(a["snp"], b["snp"])
    for key in ("orig", "est_id", "n_mini", "lsh", "lsh_valid", "n_snp_kept"):
        assert np.array_equal(a["tw"][key], b["tw"][key]), key
    assert len(a["kc"]) == len(b["kc"]) and all(np.array_equal(x, y) for x, y in zip(a["kc"], b["kc"]))
    assert len(a["sc"]) == len(b["sc"]) and all(np.array_equal(x, y) for x, y in zip(a["sc"], b["sc"]))


@pytest.mark.parametrize("world,stage2", [(2, {}), (3, {}), (2, dict(stage2_device=1)), (3, dict(stage2_device=0, stage2_first_block=64, stage2_max_block=512, stage2_pair_cap=300)),
                                          (2, dict(stage2_device=1, stage2_first_block=16, stage2_max_block=64, stage2_pair_cap=40))])
def test_sharded_tiles_equal_the_unsharded_pipeline(world, stage2):
    """... round 6: Stage 2 is dealt out by slice of every block (candidate lists, K5, per-read decisions on the read's owner; decision words and fix-up records exchanged):
    with the host's bucket walk and with the device's lists, at the default blocks and at small ones (cuts by the pair cap on ONE rank's slice, unforeseen representatives)"""
    from savont_amd.pipeline import AsvPipeline
    from savont_amd.shard import LocalExchange
    from savont_amd.synth import zymo_community
    reads = zymo_community(30000, 1011)
    ref = {}
    p0 = AsvPipeline(0)
    p0.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
    _stages(p0, ref)
    p0.close()
    assert len(ref["sc"]) >= 40
    ex = LocalExchange(world)
    outs = [dict() for _ in range(world)]; errs = []; stats = [None] * world

    def rank_main(r):
        try:
            p = AsvPipeline(0)
            p.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
            dv = p.device()
            dv.set_shard(r, world, ex.hooks[r])
            dv.set_option("shard_seeds", 1)            # off by default (it moves more bytes than it saves kernel time); tested all the same
            for k_, v_ in stage2.items():
                p.set_option(k_, v_)
            _stages(p, outs[r])
            stats[r] = (dv.get_option("shard_exchanges"), dv.get_option("shard_bytes"))
            dv.set_shard(0, 1, None)
            p.close()
        except Exception as e:                       # a rank that dies must not leave the others at the barrier
            errs.append((r, repr(e)))
            ex.barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(600)
    assert not errs and not ex.failed, errs
    print("shard exchanges / bytes per rank:", stats)
    for r in range(world):
        _same(outs[r], ref)
        assert stats[r] == stats[0] and stats[r][0] >= 40 and stats[r][1] > 20 << 20, stats     # the ranks met the same exchanges; K5, K6 (and seeds) went through them


def test_torch_exchange_wraps_device_memory_in_place():
    """the RCCL hook (savont_amd/shard.py TorchExchange) must see the library's device arrays without a copy: its view of a raw pointer aliases
    the memory, and a call through the ctypes signature the library uses runs a real (one-rank) RCCL broadcast on it"""
    import ctypes as C
    import os
    import torch
    import torch.distributed as dist
    from savont_amd.shard import TorchExchange
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(34100 + os.getpid() % 1000))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        x = torch.zeros(4096, dtype=torch.uint8, device=dev)
        ex = TorchExchange(dist, dev, 1, 0)
        v = ex._view(x.data_ptr() + 100, 50)
        v.fill_(9)
        torch.cuda.synchronize()
        assert int(x.sum().item()) == 9 * 50 and bool((x[100:150] == 9).all())
        rc = ex.hook(None, x.data_ptr(), 4, (C.c_uint64 * 2)(10, 600))
        assert rc == 0 and ex.calls == 1 and ex.bytes == 4 * 590 and int(x.sum().item()) == 9 * 50
    finally:
        if created:
            dist.destroy_process_group()


def test_sharded_seed_extraction_every_array(zymo_like=None):
    """svt_extract_seeds under svt_set_shard + shard_seeds: every array a caller can fetch -- per-read records, the sorted sets through K5, the
    SNPmer lists (allocated per rank, so compared per read), bitset rows, quality bins, and the raw minimizer lists that are gathered only on
    demand -- equals the unsharded extraction; 3 ranks, 9 000 reads"""
    from savont_amd import hip
    from savont_amd.pipeline import AsvPipeline
    from savont_amd.shard import LocalExchange
    from savont_amd.synth import zymo_community
    reads = zymo_community(9000, 1012)
    K, Cp, MINBQ = 17, 11, 10
    # SNPmers from a single-rank pipeline run
    p = AsvPipeline(0)
    p.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
    p.read_to_split_kmers(fetch=False); s = p.get_snpmers_inplace_sort(); p.close()
    assert len(s["split"]) > 50
    hf = np.sort(s["high_freq"].astype(np.uint64)) if "high_freq" in s and len(s["high_freq"]) else np.zeros(0, np.uint64)

    def extract(dev):
        dev.set_snpmers(K, s["split"], s["mid0"], s["mid1"], hf, s["cnt0"] + s["cnt1"])
        b = dev.upload(reads["seq"], reads["qual"], reads["off"])
        dev.extract_seeds(b, K, Cp, MINBQ, True)
        g = dev.fetch_seeds(b)
        pa, pf, al = dev.snpmer_bits(b)
        rng = np.random.default_rng(3)
        ai = rng.integers(0, b.n, 6000).astype(np.uint32); bi = rng.integers(0, b.n, 6000).astype(np.uint32)
        sh, sm = dev.minimizer_shared_counts(b, b, ai, bi)
        b.free()
        return dict(g=g, bits=(pa, pf, al), k5=(sh, sm))

    d0 = hip.Device(0); ref = extract(d0); d0.close()
    world = 3
    ex = LocalExchange(world); outs = [None] * world; errs = []

    def rank_main(r):
        try:
            d = hip.Device(0)
            d.set_shard(r, world, ex.hooks[r]); d.set_option("shard_seeds", 1)
            outs[r] = extract(d)
            outs[r]["n_ex"] = d.get_option("shard_exchanges")
            d.set_shard(0, 1, None); d.close()
        except Exception as e:
            errs.append((r, repr(e))); ex.barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(600)
    assert not errs and not ex.failed, errs
    for r in range(world):
        g, e = outs[r]["g"], ref["g"]
        for key in e:
            assert np.array_equal(g[key], e[key]), (r, key)          # fetch_seeds returns read-ordered compact arrays: the device layout does not show
        for x, y in zip(outs[r]["bits"], ref["bits"]): assert np.array_equal(x, y)
        for x, y in zip(outs[r]["k5"], ref["k5"]): assert np.array_equal(x, y)
        assert outs[r]["n_ex"] >= 30


def test_rccl_communicator_one_rank():
    """svt_set_shard_comm on the one GPU of the box: the library loads RCCL, creates a ONE-rank communicator and -- with the test option
    shard_world1 -- runs every sharded code path with it: the slices are the whole arrays, every exchange is a grouped ncclBroadcast from rank 0 to
    itself on the library's stream.  Exercises what a one-GPU box can of the RCCL path (symbols, communicator, group calls, stream order); the
    rank logic itself is covered by the thread-ranks above.  The results must be the unsharded pipeline's, and the whole-path call (svh_run_asv)
    with the communicator must give the same final ASVs."""
    import ctypes as C
    from savont_amd import hip
    from savont_amd.pipeline import AsvPipeline
    from savont_amd.synth import zymo_community
    reads = zymo_community(20000, 1012)
    ref = {}
    p0 = AsvPipeline(0)
    p0.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
    _stages(p0, ref)
    em0 = p0.run_asv(); fin0 = p0._consensus_set(0)
    p0.close()
    p = AsvPipeline(0)
    p.set_reads(reads["seq"], reads["qual"], reads["off"], reads["ids"])
    p.set_shard_comm(0, 1, hip.shard_comm_id())
    dv = p.device()
    dv.set_option("shard_world1", 1); dv.set_option("shard_seeds", 1)
    out = {}
    _stages(p, out)
    _same(ref, out)
    n_ex = dv.get_option("shard_exchanges")
    assert n_ex > 0 and dv.get_option("shard_bytes") > 0                 # the grouped collectives really ran
    # C1 in one call: the read block [0, n) counted, then gathered (from itself) and merged -> the same table sizes and SNPmers
    nd = C.c_uint64()
    p._chk(p.L.svh_count_partial_device(p.h, 0, len(reads["ids"]), C.byref(nd)), "count_partial_device")
    p._chk(p.L.svh_count_shard_merge(p.h), "count_shard_merge")
    assert p.L.svh_count_size(p.h) == ref["n_kept"]
    p.get_snpmers_inplace_sort()
    assert np.array_equal(p.snpmers()["split"], ref["snp"])
    p._chk(p.L.svh_snpmers_check_ranks(p.h), "snpmers_check_ranks")
    assert dv.get_option("shard_exchanges") > n_ex
    em = p.run_asv(); fin = p._consensus_set(0)
    assert fin["seqs"] == fin0["seqs"] and np.array_equal(em["depth"], em0["depth"]) and em["total"] == em0["total"]
    # an aborted communicator (what a timed-out or failed collective leaves behind; here on request): the step fails at its first exchange with SVT_ERR_EXCHANGE
    # instead of running unsharded or waiting, the process lives on, and a new communicator makes the context whole again
    from savont_amd.hip import SavontHipError
    dv.set_option("shard_timeout_s", 30)
    assert p.L.svt_shard_abort(dv.h, b"test") == 0
    with pytest.raises(Exception) as ei:
        p.run_asv()
    assert "-7" in str(ei.value) or "exchange" in str(ei.value).lower(), str(ei.value)
    p.set_shard_comm(0, 1, hip.shard_comm_id())
    em2 = p.run_asv(); fin2 = p._consensus_set(0)
    assert fin2["seqs"] == fin0["seqs"] and np.array_equal(em2["depth"], em0["depth"])
    p.close()
