"""The N > 1 bench path on the REAL model, on one GPU: two fresh child processes (torch.distributed.run, gloo, both on GPU 0)
run tests/_dist_worker.py -- broadcast_params, gradient sinks, deferred grouped weight gradients, a GraphedStep captured with
a process group alive, FlatGradAllReduce.reduce() -- and compare the reduced flat buffer with the mean of the two shards'
plain-autograd gradients; then the real training step (graph.GraphedHungarianStep: num_masks averaged over the ranks between its two
graphs) against the eager mode="loss" step of each rank.  (No scaling figure can come from one GPU; this is the correctness half of SURVEY section 8e.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_reduce_the_mean_of_their_shards():
    env = dict(os.environ, S2F_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=850)      # children only: nothing is re-exec'ed
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RANK 0 OK" in r.stdout and "RANK 1 OK" in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("hungarian_graph_step OK") == 2, r.stdout[-2000:]
    assert r.stdout.count("flat_adamw OK") == 2, r.stdout[-2000:]


@pytest.mark.timeout(900)
def test_rccl_communicator_and_hipgraph_capture_coexist(tmp_path):
    """RCCL readiness on the box the driver uses (one GPU): ONE fresh child with S2F_FORCE_DIST=1 (backend nccl = RCCL, world 1:
    communicator, watchdog thread) captures the step as a hipGraph, replays it twice and runs FlatGradAllReduce.reduce() -- the
    N > 1 bench path minus the second rank -- and a second child does the same without a process group.  The forward is
    deterministic: the losses must agree bitwise; the gradients contain split-K fp32 atomics: 1e-5 of their scale."""
    import torch
    outs = {}
    for tag, extra in (("rccl", {"S2F_FORCE_DIST": "1", "MASTER_PORT": str(_free_port())}), ("plain", {})):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        env.pop("S2F_DIST_BACKEND", None)
        if tag == "plain":
            env.pop("S2F_FORCE_DIST", None)
        out = str(tmp_path / f"{tag}.pt")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py"), out], env=env, cwd=ROOT,
                           capture_output=True, text=True, timeout=400)          # a child process: nothing is re-exec'ed
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert f"backend {'nccl' if tag == 'rccl' else 'none'}" in r.stdout, r.stdout[-1000:]
        outs[tag] = torch.load(out)
    assert torch.equal(outs["rccl"]["loss"], outs["plain"]["loss"])
    a, b = outs["rccl"]["flat"], outs["plain"]["flat"]
    assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()
