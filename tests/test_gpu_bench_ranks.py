"""bench.py's N > 1 step on a one-GPU box: the control flow an 8-GPU run takes must need no debugging
when it first meets such a node.  (a) `--gpus 2 --backend host`: two torch-free ranks (sharing the GPU, tables through the host) started
by bench.py itself, and the same two ranks under the driver's launcher (`python -m torch.distributed.run`); (b) `--gpus 1 --force-gather`: the N > 1 pipeline with the product's C++ gather (libugs_rccl.so, RCCL
communicator of one rank) issued beside the next step's kernels; (c) the plain one-GPU step.  The three search the same global
query stream, so their hit counts must agree, and the JSON lines must carry the contract's fields."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, launcher=None):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable] + (launcher or []) + [os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--db", "20000",
                          "--cpu-baseline", "none"] + list(args), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_host_ranks_one_gathering_rank_and_one_plain_rank_agree():
    one = _bench("--gpus", "1", "--queries", "8000")
    gat = _bench("--gpus", "1", "--queries", "8000", "--force-gather")
    two = _bench("--gpus", "2", "--queries", "4000", "--backend", "host")
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    # the driver's own command line for N > 1 (the launcher is a process of its own; the ranks stay torch-free)
    drv = _bench("--gpus", "2", "--queries", "4000", "--backend", "host",
                 launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)])
    assert drv["detail"]["hits_per_step"] == two["detail"]["hits_per_step"] and drv["n_gpus"] == 2
    for j, n in ((one, 1), (gat, 1), (two, 2), (drv, 2)):
        # ONE HIP runtime, ONE HSA runtime, at most one RCCL, no torch in any rank: from every rank's /proc/self/maps
        libs = j["detail"]["runtime_libs"]
        assert len(libs) == n
        for l_ in libs:
            assert l_["torch_imported"] is False
            assert len(l_["libamdhip64"]) == 1 and len(l_["libhsa-runtime64"]) == 1 and len(l_.get("librccl", [])) <= 1, l_
            assert all("/torch/" not in x for k in ("libamdhip64", "libhsa-runtime64", "librccl") for x in l_.get(k, [])), l_
        assert j["n_gpus"] == n and j["steps"] == 2 and j["warmup"] == 1 and j["unit"] == "query-seqs/s"
        assert j["scaling"] == "weak" and j["config"]["queries_per_step"] == 8000 and j["config"]["queries_per_gpu"] == 8000 // n
        assert j["value"] > 0 and abs(j["value"] - 8000 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
        assert j["roofline"]["kernel"] in ("k_rank", "k_rank2", "k_align") and "traffic_source" in j["roofline"]
    assert one["detail"]["hits_per_step"] > 5000
    assert gat["detail"]["hits_per_step"] == one["detail"]["hits_per_step"]
    assert two["detail"]["hits_per_step"] == one["detail"]["hits_per_step"]
    assert "ugs_gather_results" in gat["config"]["gather"] and "host transport" in two["config"]["gather"]
    assert len(gat["detail"]["runtime_libs"][0]["librccl"]) == 1
    pr = two["detail"]["per_rank"]
    assert len(pr) == 2 and all(r["queries"] == 4000 and r["ms_rank"] > 0 for r in pr)
    assert gat["detail"]["host_ms_gather"] > 0


def test_strong_scaling_variant_splits_one_batch():
    """`--scaling strong`: the same C2-sized batch in N contiguous shards - same hits as one GPU, labelled strong"""
    one = _bench("--gpus", "1", "--queries", "8000")
    two = _bench("--gpus", "2", "--queries", "8000", "--backend", "host", "--scaling", "strong")
    assert two["scaling"] == "strong" and two["n_gpus"] == 2 and two["config"]["queries_per_step"] == 8000 and two["config"]["queries_per_gpu"] == 4000
    assert two["detail"]["hits_per_step"] == one["detail"]["hits_per_step"]
    assert "STRONG" in two["config"]["workload"]
    pr = two["detail"]["per_rank"]
    assert len(pr) == 2 and all(r["queries"] == 4000 for r in pr)
