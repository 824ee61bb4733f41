"""Image-sharded data parallelism: one process per GPU, gradients averaged with bucketed all-reduce over
RCCL / xGMI (``torch.distributed`` backend "nccl" is RCCL on ROCm), overlapped with the backward schedule.

The reference uses ``torch.nn.DataParallel`` (train_detector_3D_angle.py:317): one process scatters the batch,
replicates the weights every step and reduces gradients onto GPU 0.  Here every rank owns its images and its
replica; the only exchange is the gradient average (36.6 M parameters, 146.6 MB fp32 for ResNet-50).  The engine's
backward runs layers in a fixed reverse order (heads -> FPN -> layer4 .. stem) and reports each layer's finished
gradients through ``Engine.grad_hook``; ``GradReducer`` packs them into flat buckets and starts an asynchronous
all-reduce per bucket while the remaining dgrad / wgrad kernels keep the compute stream busy.  Buckets are large
(default 32 MB: xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so few large ring steps beat many small
ones) and the frozen batch-norm needs no statistics traffic.  Per-image losses are self-normalised, so the mean
of per-rank losses equals the global mean for equal shards.

Device agnostic: with the "gloo" backend the same class averages CPU tensors (used by the CPU tests).
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, bucket_bytes=32 << 20, group=None):
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._reset()

    def _reset(self):
        self.pending = []            # [(name, grad)] not yet flushed
        self.pending_bytes = 0
        self.flights = []            # [(work, flat, [(name, shape, offset, numel)])]

    def hook(self, grads):
        """Engine.grad_hook: called with {name: grad} as soon as a layer's gradients are final."""
        if self.world == 1:
            return
        for name, g in grads.items():
            self.pending.append((name, g))
            self.pending_bytes += g.numel() * g.element_size()
        if self.pending_bytes >= self.bucket_bytes:
            self._flush()

    def _flush(self):
        if not self.pending:
            return
        flat = torch.cat([g.reshape(-1) for _, g in self.pending])
        layout, off = [], 0
        for name, g in self.pending:
            layout.append((name, tuple(g.shape), off, g.numel()))
            off += g.numel()
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.flights.append((work, flat, layout))
        self.pending, self.pending_bytes = [], 0

    def finalize(self, grads):
        """Wait for every bucket and return {name: averaged grad} (views into the flat buckets)."""
        if self.world == 1:
            return grads
        self._flush()
        out = dict(grads)
        inv = 1.0 / self.world
        for work, flat, layout in self.flights:
            work.wait()
            flat.mul_(inv)
            for name, shape, off, n in layout:
                out[name] = flat[off:off + n].view(shape)
        self._reset()
        return out


def init_from_env(backend=None):
    """torch.distributed.run / torchrun rendezvous (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RN_REHEARSE_ONE_GPU"):
        # Development only: rehearse the N > 1 code path on a box with ONE GPU -- every rank uses cuda:0 and the
        # gradients travel over gloo (RCCL refuses two ranks on one device).  Never set by bench.py or the driver.
        backend, local = "gloo", 0
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
