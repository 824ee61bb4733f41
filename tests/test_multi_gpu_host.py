"""CPU tests (gloo, world 2) of the two multi-GPU host paths that have no collective in their kernels:

* retinanet_mi355x/trainer.py -- the reference trainer's loop (train_detector_3D_angle.py:337-417) for one process per GPU:
  both ranks take the same skip decision when ONE rank's forward raises, step ReduceLROnPlateau with the same mean loss, end
  every epoch with identical weights and learning rate, and rank 0's per-epoch checkpoint is the state_dict both hold.
  (The HIP model has no CPU path, so a small torch module with the detector's call signature stands in; tools/train_ddp.py
  runs the real one and tests/test_gpu_train_ddp.py rehearses it on the GPU box.)
* retinanet_mi355x/multicam.py -- BASELINE configs[3] over N ranks: the camera -> rank map, the ragged gather of the ranks'
  survivors and a merge order that does not depend on N (MC3D_crop_tracker.py:1051-1088, 1489-1509).
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "3d-playground_amd")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Stand_in(torch.nn.Module):
    """Three losses from [im, label], like the detector's training forward; gradients averaged over the ranks as they arrive."""

    def __init__(self, world):
        super().__init__()
        torch.manual_seed(0)                                   # same weights on every rank
        self.a = torch.nn.Linear(6, 4)
        self.b = torch.nn.Linear(4, 3)
        self.fail_at = None
        self.calls = 0
        if world > 1:
            for p in self.parameters():
                p.register_hook(lambda g: _avg(g, world))

    def freeze_bn(self):
        pass

    def forward(self, inputs):
        im, label = inputs
        self.calls += 1
        if self.fail_at is not None and self.calls == self.fail_at:
            raise RuntimeError("stack expects a non-empty TensorList (no image in the batch has a label)")
        out = self.b(torch.tanh(self.a(im)))
        d = (out - label) ** 2
        return d[:, 0:1].mean(0), d[:, 1:2].mean(0), d[:, 2:3].mean(0)


def _avg(g, world):
    g = g.clone()
    dist.all_reduce(g)
    return g / world


def _train_worker(rank, world, port, tmp, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from retinanet_mi355x import ddp, trainer
    ddp.init_from_env(backend="gloo")
    net = _Stand_in(world)
    if rank == 1:
        net.fail_at = 2                                        # ONE rank's second forward raises: every rank must skip that iteration
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=0, mode="min", factor=0.5, threshold=10.0)   # "no improvement" every epoch

    def batches(epoch):
        g = torch.Generator().manual_seed(100 * epoch + rank)  # rank-local shards
        for _ in range(3):
            yield torch.randn(5, 6, generator=g), torch.randn(5, 3, generator=g)
    logs = []
    hist = trainer.train(net, opt, sched, batches, 3, clip_norm=0.1, checkpoint=os.path.join(tmp, "ck_e{}.pt"), rank=rank,
                         log=logs.append, log_every=1)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    lr = torch.tensor([opt.param_groups[0]["lr"]], dtype=torch.float64)
    lrs = [torch.zeros_like(lr) for _ in range(world)]
    dist.all_gather(lrs, lr)
    ok = torch.equal(both[0], both[1])                         # bit-identical replicas after 8 optimizer steps
    ok &= float(lrs[0]) == float(lrs[1]) and float(lr) < 1e-2 # the plateau scheduler moved, identically
    ok &= [h["skipped"] for h in hist] == [1, 0, 0] and [h["iterations"] for h in hist] == [2, 3, 3]
    means = torch.tensor([h["mean_loss"] for h in hist], dtype=torch.float64)
    ms = [torch.zeros_like(means) for _ in range(world)]
    dist.all_gather(ms, means)
    ok &= torch.equal(ms[0], ms[1])                            # the scheduler saw the same numbers
    ok &= any("non-empty TensorList" in m for m in logs) == (rank == 1)
    dist.barrier()
    if rank == 0:                                              # the last checkpoint is the state_dict both ranks hold
        sd = torch.load(os.path.join(tmp, "ck_e2.pt"), weights_only=True)
        ok &= sorted(os.listdir(tmp)) == ["ck_e0.pt", "ck_e1.pt", "ck_e2.pt"]
        ok &= all(torch.equal(sd[k], v) for k, v in net.state_dict().items())
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_trainer_loop_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_train_worker, args=(world, port, str(tmp_path), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _unequal_worker(rank, world, port, tmp, out):
    """Shards of unequal length (rank 0: three batches, rank 1: two): the exhausted rank keeps answering the per-iteration all-reduce,
    nobody hangs, both ranks count the same iterations and end with identical weights."""
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from retinanet_mi355x import ddp, trainer
    ddp.init_from_env(backend="gloo")
    net = _Stand_in(world)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)

    def batches(epoch):
        g = torch.Generator().manual_seed(7 * epoch + rank)
        for _ in range(3 - rank):
            yield torch.randn(5, 6, generator=g), torch.randn(5, 3, generator=g)
    hist = trainer.train(net, opt, None, batches, 2, rank=rank, log=lambda m: None)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    ok = torch.equal(both[0], both[1])
    ok &= [h["iterations"] for h in hist] == [2, 2] and [h["skipped"] for h in hist] == [1, 1]     # the third iteration: one rank short
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_trainer_unequal_shards_do_not_hang_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_unequal_worker, args=(world, port, str(tmp_path), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_checkpoint_with_dataparallel_keys_round_trip(tmp_path):
    """train(..., dataparallel_keys=True) writes the ``module.``-prefixed keys of the reference's multi-GPU trainer
    (train_detector_3D_angle.py:415-417); stripping the prefix (what its to_cpu helper does by hand, :39-59, and what this package's
    load_state_dict does itself) gives back the weights.  (No reference fixture covers the round trip: parity unpinned.)"""
    sys.path.insert(0, PKG)
    from retinanet_mi355x import trainer
    net = _Stand_in(1)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)

    def batches(epoch):
        g = torch.Generator().manual_seed(epoch)
        yield torch.randn(4, 6, generator=g), torch.randn(4, 3, generator=g)
    trainer.train(net, opt, None, batches, 1, checkpoint=str(tmp_path / "dp_e{}.pt"), dataparallel_keys=True, log=lambda m: None)
    sd = torch.load(tmp_path / "dp_e0.pt", weights_only=True)
    assert sd and all(k.startswith("module.") for k in sd)
    twin = _Stand_in(1)
    twin.load_state_dict({k[len("module."):]: v for k, v in sd.items()})
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), twin.state_dict().values()))
    trainer.train(net, opt, None, batches, 1, checkpoint=str(tmp_path / "bare_e{}.pt"), log=lambda m: None)
    assert not any(k.startswith("module.") for k in torch.load(tmp_path / "bare_e0.pt", weights_only=True))


def test_trainer_single_process_skips_like_the_reference():
    """World 1: an iteration whose forward raises is printed and skipped (train_detector_3D_angle.py:406-408), a zero loss is
    skipped before backward (:380-381); nothing else changes."""
    sys.path.insert(0, PKG)
    from retinanet_mi355x import trainer
    net = _Stand_in(1)
    net.fail_at = 1
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    before = [p.detach().clone() for p in net.parameters()]

    def batches(epoch):
        yield torch.ones(2, 6), torch.zeros(2, 3)              # raises
        x = torch.ones(2, 6)
        with torch.no_grad():
            y = net.b(torch.tanh(net.a(x)))
        yield x, y                                             # loss exactly 0: skipped
    logs = []
    hist = trainer.train(net, opt, None, batches, 1, log=logs.append)
    assert hist[0]["skipped"] == 2 and hist[0]["iterations"] == 0
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
    assert any("non-empty TensorList" in m for m in logs)


# ---------------------------------------------------------------------------------------------------- cameras over ranks
def test_camera_shards_cover_every_camera_once():
    sys.path.insert(0, PKG)
    from retinanet_mi355x import multicam
    assert len(multicam.CAMERAS) == 18 and multicam.CAMERAS[0] == "p1c1" and multicam.CAMERAS[-1] == "p3c6"
    for world in (1, 2, 3, 4, 8, 18, 20):
        sh = multicam.shards(18, world)
        assert sorted(c for s in sh for c in s) == list(range(18))
        sizes = [len(s) for s in sh]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert [len(s) for s in multicam.shards(18, 8)] == [3, 3, 2, 2, 2, 2, 2, 2]
    assert multicam.shard(18, 8, 1) == [1, 9, 17]
    with pytest.raises(ValueError):
        multicam.shard(18, 8, 8)
    local = torch.tensor([0, 2, 1, 1])
    assert multicam.to_global(local, [1, 9, 17]).tolist() == [1, 17, 9, 9]


def _fake_detections(cam, k):
    """What one camera's share of a MULTI_FRAME call looks like: scores descending, with values that collide across cameras."""
    g = torch.Generator().manual_seed(50 + cam)
    scores = torch.sort(torch.randint(1, 6, (k,), generator=g).float() / 8, descending=True)[0]
    labels = torch.randint(0, 8, (k,), generator=g)
    boxes = torch.rand(k, 20, generator=g) * 1000 + cam
    return scores, labels, boxes, torch.full((k,), cam, dtype=torch.int64)


def _rank_part(cams, counts):
    """A rank's detector output over its cameras: per-camera survivors interleaved by descending score, as batched_nms returns."""
    parts = [_fake_detections(c, counts[c]) for c in cams]
    s = torch.cat([p[0] for p in parts])
    order = torch.sort(s, descending=True, stable=True)[1]
    return tuple(torch.cat([p[i] for p in parts])[order] for i in range(4))


def test_merge_does_not_depend_on_the_number_of_ranks():
    sys.path.insert(0, PKG)
    from retinanet_mi355x import multicam
    counts = {c: (0 if c == 4 else 3 + (c * 7) % 5) for c in range(18)}                  # ragged, one camera with nothing
    ref = multicam.merge([_rank_part(list(range(18)), counts)])
    assert ref[0].numel() == sum(counts.values())
    assert bool((ref[0][:-1] >= ref[0][1:]).all())
    for world in (2, 3, 8):
        got = multicam.merge([_rank_part(s, counts) for s in multicam.shards(18, world)])
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), world


def _gather_worker(rank, world, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from retinanet_mi355x import ddp, multicam
    ddp.init_from_env(backend="gloo")
    counts = {c: (0 if c % 2 == 1 else 2 + c) for c in range(5)}                         # rank 1 (cameras 1, 3) has NO survivors
    mine = _rank_part(multicam.shard(5, world, rank), counts)
    parts = multicam.gather_detections(*mine)
    ok = len(parts) == world
    for r, p in enumerate(parts):                                                        # bit-exact transport, ragged sizes
        want = _rank_part(multicam.shard(5, world, r), counts)
        ok &= all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(p, want))
    merged = multicam.merge(parts)
    ref = multicam.merge([_rank_part(list(range(5)), counts)])
    ok &= all(torch.equal(a, b) for a, b in zip(merged, ref))
    empty = multicam.gather_detections(mine[0][:0], mine[1][:0], mine[2][:0], mine[3][:0])   # a time step with no detection anywhere
    ok &= all(p[0].numel() == 0 and p[2].shape == (0, 20) for p in empty)
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_detections_world2_gloo():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_gather_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
