"""CPU, world_size 2 over gloo: the gradient reducer the engine's backward feeds (retinanet_mi355x.ddp) averages
gradients across ranks bucket by bucket and hands back correctly shaped views."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "3d-playground_amd")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from retinanet_mi355x import ddp
    r, local, w = ddp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    red = ddp.GradReducer(bucket_bytes=4096)          # small buckets: several flushes in flight
    shapes = {"a.weight": (7, 3, 3, 3), "a.bias": (7,), "b.weight": (300, 5), "c.weight": (2, 2), "d.bias": (1025,)}
    g = torch.Generator().manual_seed(100 + rank)
    grads = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
    mine = {k: v.clone() for k, v in grads.items()}
    # the engine reports finished layers one at a time, in reverse network order
    red.hook({"d.bias": grads["d.bias"]})
    red.hook({"c.weight": grads["c.weight"], "b.weight": grads["b.weight"]})
    red.hook({"a.weight": grads["a.weight"], "a.bias": grads["a.bias"]})
    avg = red.finalize(grads)
    # reference: all-gather every rank's originals
    ok = True
    for k, s in shapes.items():
        gathered = [torch.zeros(s) for _ in range(world)]
        dist.all_gather(gathered, mine[k])
        want = sum(gathered) / world
        ok &= avg[k].shape == torch.Size(s) and torch.allclose(avg[k], want, atol=1e-6)
    # a second step must start from a clean state
    red.hook({"a.bias": torch.full((7,), float(rank))})
    again = red.finalize({"a.bias": None})
    ok &= torch.allclose(again["a.bias"], torch.full((7,), (world - 1) / 2.0))
    # flat path: the engine's persistent buffer, buckets = slices reduced in place as they complete
    arena = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red.backward_begins()
    red.bucket_ready(0, arena[0:300])
    red.bucket_ready(1, arena[300:1000])
    red.finalize_flat(arena)
    ok &= torch.allclose(arena, torch.arange(1000, dtype=torch.float32) * (sum(range(1, world + 1)) / world))
    ok &= not red.flat_flights
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_grad_reducer_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_grad_reducer_single_process_is_identity():
    sys.path.insert(0, PKG)
    from retinanet_mi355x import ddp
    red = ddp.GradReducer()
    g = {"w": torch.ones(3)}
    red.hook(g)
    assert red.finalize(g)["w"] is g["w"]
