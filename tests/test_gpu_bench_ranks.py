"""GPU test of bench.py's own rank launcher: `python bench.py --gpus 2` with WORLD_SIZE unset must start two rank processes itself (a child
torch.distributed.run, before anything touches the GPU) and print ONE JSON line with n_gpus == 2 that carries both ways the path shards -- the
sample-per-rank leg at the top level and the pooled leg (`pooled`) -- each with roofline, cpu_baseline and a parity object that includes
final_asvs.  The GPU box has one device: --oversubscribe puts both ranks on it with gloo in RCCL's place (a test mode, the numbers mean
nothing); the RCCL communicator of the library is exercised by tests/test_gpu_shard.py::test_rccl_communicator_one_rank.
Reference parallel structure the ranks stand for: src/alignment.rs:1786 (par_iter over all reads), src/seq_parse.rs:168,396 (k-mer shards)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(lines[0])


SMALL = ["--reads", "6000", "--cpu-sample", "6000", "--steps", "2", "--warmup", "1", "--no-cpu-t20", "--no-extra-legs", "--in-flight", "2"]


def test_gpus_2_launches_two_ranks_and_reports_both_legs():
    out = _run(["--gpus", "2", "--oversubscribe", "--pooled-reads", "12000", "--samples", "4", "--pooled-steps", "1", "--pooled-warmup", "1"] + SMALL)
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["scaling"] == "weak"
    assert "OVERSUBSCRIBED" in out["config"]["collective_backend"]
    assert len(out["config"]["asvs_per_rank"]) == 2 and all(x > 0 for x in out["config"]["assigned_per_rank"])
    assert out["roofline"]["kernel"] and out["cpu_baseline"]["kind"] == "port"
    par = out["parity_6k"]
    assert par["ok"] and par["final_asvs"] is True
    pl = out["pooled"]
    assert pl["n_gpus"] == 2 and pl["scaling"] == "strong" and pl["config"]["samples"] == 4 and pl["roofline"]["kernel"]
    pp = pl["parity_pooled"]
    assert pp["ok"] and pp["mode"] == "oracle" and pp["final_asvs"] is True and pp["per_sample"] is True
    assert pp["final_asvs_equal_one_rank_run"] is True and pp["per_sample_equal_one_rank_run"] is True
    assert pl["shard"]["exchanges_per_step"] > 0 and pl["cpu_baseline"]["kind"] == "port"


def test_failing_exchange_costs_the_pooled_object_not_the_line():
    """the first N > 1 run must not be able to cost the whole line: rank 1's exchange hook fails in the middle of the pooled leg (it leaves the collective the
    others are in); rank 0 still prints ONE line -- the complete sample-per-GPU leg with `pooled` = {"error": ...} -- and the job ends non-zero well inside the deadline"""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SAVONT_TEST_FAIL_EXCHANGE"] = "1:5"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--pooled-reads", "12000", "--samples", "4", "--pooled-steps", "1", "--pooled-warmup", "1",
                        "--pooled-timeout", "240", "--no-cpu-baseline"] + SMALL, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    took = time.time() - t0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["roofline"]["kernel"] and out["config"]["ranks"] == 2
    assert "error" in out["pooled"] and took < 600, (out["pooled"], took)
    assert "fails on rank 1 by request" in r.stderr


def test_one_rank_line_keeps_its_shape():
    out = _run(["--gpus", "1"] + SMALL)
    assert out["n_gpus"] == 1 and "pooled" not in out and out["parity_6k"]["ok"] and out["parity_6k"]["final_asvs"] is True
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key


def test_more_ranks_than_gpus_is_refused_without_the_test_flag():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    assert r.returncode != 0 and "--oversubscribe" in (r.stderr + r.stdout)
