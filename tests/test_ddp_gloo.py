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
    # defer_scale: the SUM stays in the buffer, the optimizer's grad_scale is 1/world
    red2 = ddp.GradReducer(defer_scale=True)
    arena = torch.arange(100, dtype=torch.float32) * (rank + 1)
    red2.bucket_ready(0, arena[0:100])
    red2.finalize_flat(arena)
    ok &= torch.allclose(arena, torch.arange(100, dtype=torch.float32) * sum(range(1, world + 1))) and red2.grad_scale == 1.0 / world
    # ... and the generic path keeps the same contract: finalize() returns SUMS, grad_scale times them is the mean (not 1/world^2)
    gsum = {"w": torch.full((10,), float(rank + 1))}
    red2.hook(gsum)
    summed = red2.finalize(gsum)
    ok &= torch.allclose(summed["w"] * red2.grad_scale, torch.full((10,), sum(range(1, world + 1)) / world))
    # loss scalars for logging / ReduceLROnPlateau (train_detector_3D_angle.py:374-381, 338, 412): one 3-float all-reduce; both
    # ranks step their scheduler with the SAME number and end up with the same learning rate, which rank-local losses do not
    local = [torch.tensor([1.0 + rank]), torch.tensor([0.5 * (1 + rank)]), torch.tensor([3.0 - rank])]
    m = ddp.mean_losses(*local)
    ok &= m.shape == (3,) and torch.allclose(m, torch.tensor([1.5, 0.75, 2.5]))
    w = torch.nn.Parameter(torch.zeros(1))
    traj = [1.0, 1.0, 1.0 + 0.2 * rank, 1.0 - 0.3 * rank, 1.0 + 0.1 * rank]     # rank 1 alone sees an improvement at step 3
    lrs = {}
    for which in ("local", "mean"):
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(torch.optim.SGD([w], lr=1.0), mode="min", patience=1)
        for v in traj:
            x = torch.tensor([v])
            sched.step(float(x if which == "local" else ddp.mean_losses(x)[0]))
        lr = torch.tensor([sched.optimizer.param_groups[0]["lr"]])
        both = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(both, lr)
        lrs[which] = [float(b) for b in both]
    ok &= lrs["mean"][0] == lrs["mean"][1] and lrs["local"][0] != lrs["local"][1]
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


def test_flat_plan_cuts_the_exposed_tail_bucket():
    """Engine._flat_plan: the last bucket (final only when the backward ends) is cut down to the last whole layers that fit in
    tail_bytes; ResNet-50: layer2 + layer1 + stem (< 6 MB) instead of 28.6 MB."""
    sys.path.insert(0, PKG)
    from retinanet_mi355x import arch, engine

    class W:                                        # stands in for the parameter (only .shape is read by the plan)
        def __init__(self, shape):
            self.shape = shape
    eng = engine.Engine("resnet50", 8, 12)
    shapes = arch.state_dict_shapes("resnet50", 8, 12)
    for name, L in eng.layers.items():
        L.weight = W(tuple(shapes[name + ".weight"]))
    eng.set_flat_grads(32 << 20, tail_bytes=None)
    old = eng._flat_plan("cpu")["buckets"]
    eng.set_flat_grads(32 << 20)
    plan = eng._flat_plan("cpu")
    b = plan["buckets"]
    assert len(b) == len(old) + 1 and b[:-2] == old[:-1]
    assert b[0][0] == 0 and b[-1][1] == plan["total"] and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert 4 * (b[-1][1] - b[-1][0]) <= 6 << 20 and b[-1][2] == "conv1"
    order = eng.finish_order()
    assert order[order.index(b[-2][2]) + 1].startswith("layer2.")          # the tail starts with layer2's last block
    assert [x[2] for x in b] == sorted((x[2] for x in b), key=order.index)  # bucket ends follow the backward's order
