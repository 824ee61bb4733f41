"""Image-sharded data parallelism: one process per GPU, gradients averaged with bucketed all-reduce over
RCCL / xGMI (``torch.distributed`` backend "nccl" is RCCL on ROCm), overlapped with the backward schedule.

The reference uses ``torch.nn.DataParallel`` (train_detector_3D_angle.py:317): one process scatters the batch,
replicates the weights every step and reduces gradients onto GPU 0.  Here every rank owns its images and its
replica; the only exchange is the gradient average (36.6 M parameters, 146.6 MB fp32 for ResNet-50).

The engine's backward runs layers in a fixed reverse order (heads -> FPN -> layer4 .. stem).  With a reducer attached
(``ResNet.set_gradient_reducer``) the engine writes every parameter gradient straight into its slot of ONE persistent
flat buffer laid out in that order (``Engine.set_flat_grads``); a bucket is a contiguous slice of the buffer, and the
moment the last layer of a slice has retired the engine hands the slice to ``GradReducer.bucket_ready``, which starts an
asynchronous all-reduce on it IN PLACE while the remaining dgrad / wgrad kernels keep the compute stream busy.  No
packing copy, no per-step allocation, and the gradient pointers the optimizer sees never change.  Buckets are large
(default 32 MB: xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so few large ring steps beat many small ones)
and the frozen batch-norm needs no statistics traffic.  Per-image losses are self-normalised, so the mean of per-rank
losses equals the global mean for equal shards.

``hook`` / ``finalize`` are the generic form of the same thing for gradients that arrive as separate tensors (used by
the CPU tests over "gloo"; it packs them with one copy per bucket).
"""
import time

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, bucket_bytes=32 << 20, group=None, timeline=False, tail_bytes=6 << 20, defer_scale=False):
        """tail_bytes: size the last (fully exposed) bucket is cut down to (Engine.set_flat_grads).  defer_scale: leave the
        gradient SUM in the flat buffer -- the caller's optimizer multiplies by 1/world while it reads the gradients anyway
        (optim.ClipAdam(grad_scale=reducer.grad_scale)): saves one pass over the 147 MB buffer after the last bucket.  Off by
        default because torch.optim.Adam / clip_grad_norm_ (the reference's loop) expect averaged p.grad."""
        self.bucket_bytes = bucket_bytes
        self.tail_bytes = tail_bytes
        self.defer_scale = defer_scale
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.timeline = [] if timeline else None      # per step: [{bucket, bytes, launch_ms, done_ms}], backward_ms
        self._reset()

    def _reset(self):
        self.pending = []            # [(name, grad)] not yet flushed
        self.pending_bytes = 0
        self.flights = []            # [(work, flat, [(name, shape, offset, numel)])]
        self.flat_flights = []       # [(bucket, work, slice, launch event or None)]
        self.t0 = None

    # ---- flat path: slices of the engine's persistent gradient buffer
    def attach(self, engine):
        engine.set_flat_grads(self.bucket_bytes, self.tail_bytes)
        engine.bucket_hook = self.bucket_ready

    @property
    def grad_scale(self):
        """What the optimizer must multiply the gradients by: 1/world with defer_scale, else 1."""
        return 1.0 / self.world if (self.defer_scale and self.world > 1) else 1.0

    def backward_begins(self):
        if self.timeline is not None and torch.cuda.is_available():
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t0.record()
            self.t0_host = time.perf_counter()

    def bucket_ready(self, index, flat_slice):
        """Engine.bucket_hook: every gradient inside flat_slice is final (kernels enqueued on the current stream)."""
        if self.world == 1:
            return
        ev = None
        if self.t0 is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                               # retires when the bucket's last unpack kernel has retired
        work = dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.flat_flights.append([index, work, flat_slice, ev, time.perf_counter(), None])
        self._poll()

    def _poll(self):
        if self.t0 is None:
            return
        for f in self.flat_flights:
            if f[5] is None and f[1].is_completed():
                f[5] = time.perf_counter()

    def finalize_flat(self, arena):
        """Wait for every bucket and scale the whole buffer by 1/world (one launch; with defer_scale the optimizer does it)."""
        if self.world == 1:
            return
        end = None
        if self.t0 is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()                              # the backward's last kernel
        for f in self.flat_flights:
            f[1].wait()
            if f[5] is None:
                f[5] = time.perf_counter()
        if not self.defer_scale:
            arena.mul_(1.0 / self.world)
        if self.t0 is not None:
            torch.cuda.synchronize()
            self.timeline.append({
                "backward_gpu_ms": self.t0.elapsed_time(end),
                "buckets": [{"bucket": f[0], "mbytes": round(f[2].numel() * 4 / 1e6, 1),
                             "ready_gpu_ms": round(self.t0.elapsed_time(f[3]), 2),      # when its gradients were final
                             "launch_host_ms": round(1e3 * (f[4] - self.t0_host), 2),   # when the host issued the all-reduce
                             "done_host_ms": round(1e3 * (f[5] - self.t0_host), 2)} for f in self.flat_flights]})
        self._reset()

    # ---- generic path: separate gradient tensors
    def hook(self, grads):
        """Called with {name: grad} as soon as a layer's gradients are final."""
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
        """Wait for every bucket and return {name: averaged grad} (views into the flat buckets); with defer_scale the gradient
        SUMS, to be multiplied by ``grad_scale`` by the optimizer -- the same contract as finalize_flat."""
        if self.world == 1:
            return grads
        self._flush()
        out = dict(grads)
        inv = 1.0 / self.world
        for work, flat, layout in self.flights:
            work.wait()
            if not self.defer_scale:                  # defer_scale: the optimizer applies grad_scale (as after finalize_flat)
                flat.mul_(inv)
            for name, shape, off, n in layout:
                out[name] = flat[off:off + n].view(shape)
        self._reset()
        return out


def mean_losses(*losses, group=None):
    """Mean over the ranks of the step's loss scalars, as ONE all-reduce of len(losses) floats issued after backward; returns a
    device tensor [len(losses)] (no host synchronisation here).

    The reference's DataParallel gathers the replicas' losses and takes their mean (train_detector_3D_angle.py:374-378); that
    mean is what is printed, kept in epoch_loss and handed to ReduceLROnPlateau (:338, 389-395, 412).  With one process per GPU
    a scheduler stepped on a rank-LOCAL loss would give the ranks different learning rates and then different weights: every
    rank must step its scheduler with THIS value (identical on all ranks: same bits from the same collective)."""
    dev = losses[0].device
    t = torch.stack([l.detach().reshape(-1)[0].float() for l in losses]).to(dev)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t /= dist.get_world_size(group)
    return t


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
        if backend == "nccl":
            torch.cuda.set_device(local)              # RCCL binds the communicator to the current device
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
