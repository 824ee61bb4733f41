"""The multi-GPU training path with the REAL engine, rehearsed on one GPU: two ranks share cuda:0 and average their
gradients over gloo (RN_REHEARSE_ONE_GPU, retinanet_mi355x/ddp.py).  What the driver runs on 8 GPUs over RCCL differs
only in the backend and the device index.  The workers are child processes (3 processes on the card in total)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_reduced_gradients_are_the_mean_of_the_ranks(dev):
    env = dict(os.environ, RN_REHEARSE_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(HERE, "ddp_rehearsal_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["ok"] and res["params"] > 60, res
