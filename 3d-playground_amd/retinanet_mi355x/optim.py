"""Fused global-norm clip + Adam over all parameters (three HIP launches, no host sync; step number AND hyper-parameters live
on the device, so the step can be captured into a hipGraph and a scheduler's lr still reaches the replays).

Same arithmetic as the reference trainer's ``torch.nn.utils.clip_grad_norm_(params, 0.1)`` followed by
``torch.optim.Adam(params, lr=1e-4).step()`` (train_detector_3D_angle.py:337, 385-387); the reference's torch
optimizer keeps working with the drop-in model too -- this is the native alternative ``bench.py`` uses.
"""
import struct

import numpy as np
import torch

from . import _hip

CHUNK = 4096


class ClipAdam(torch.optim.Optimizer):
    """Not built on Optimizer.__init__ (no per-parameter Python state); subclassing only satisfies the isinstance
    check of torch's lr schedulers."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=0.1, grad_scale=1.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no parameters")
        _hip.need_gpu(*self.params)
        self.betas, self.eps, self.max_norm = betas, eps, max_norm
        # every gradient is multiplied by this before the norm and the update: 1/world for a data-parallel gradient SUM
        # (ddp.GradReducer(defer_scale=True) leaves the sum in the buffer and saves the pass that would scale it)
        self.grad_scale = float(grad_scale)
        # torch.optim-style group so lr schedulers (ReduceLROnPlateau, train_detector_3D_angle.py:338, 412) can drive it
        self.param_groups = [{"params": self.params, "lr": lr, "betas": betas, "eps": eps}]
        self.defaults = {"lr": lr}
        self.step_count = 0
        dev = self.params[0].device
        self.m = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.v = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        chunks = []
        for ti, p in enumerate(self.params):
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise RuntimeError("ClipAdam takes contiguous fp32 parameters")
            chunks += [(ti, c) for c in range((p.numel() + CHUNK - 1) // CHUNK)]
        self.n_chunks = len(chunks)
        self.chunk_table = torch.tensor(chunks, dtype=torch.int32, device=dev)
        # Pointer table (param, grad, m, v, numel per tensor).  The upload is asynchronous, so a pinned host table may be
        # rewritten only after the copy that read it has completed, and a device table only after the kernels that read
        # it have run: two (host, device) pairs used in turn, each guarded by the event recorded after its last use; the
        # upload is skipped altogether while the pointers stay what the device table already holds (persistent
        # gradients: Engine.flat_grads).
        n5 = len(self.params) * 5
        self.tables = [{"host": torch.empty(n5, dtype=torch.int64).pin_memory(),
                        "dev": torch.empty(n5, dtype=torch.int64, device=dev), "event": None, "ptrs": None}
                       for _ in range(2)]
        self.turn = 0
        lib = _hip.load()
        self.ws = torch.empty(lib.rn_opt_workspace_bytes(self.n_chunks), dtype=torch.uint8, device=dev)
        self.total_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)    # the step number lives on the device (hipGraph replay)
        # ... and so do the hyper-parameters [lr, max_norm, beta1, beta2, eps, grad_scale] (rn_opt_clip_adam_hp): a
        # scheduler's new lr reaches a replayed graph through sync_hyperparameters(), nothing is frozen into the capture
        self.hp_dev = torch.zeros(6, dtype=torch.float32, device=dev)
        self.hp_host = [torch.zeros(6, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.hp_turn = 0
        self.hp_now = None
        self.sync_hyperparameters()

    def _hyper(self):
        g = self.param_groups[0]
        b = g.get("betas", self.betas)
        return (float(g["lr"]), float(self.max_norm if self.max_norm else 0.0), float(b[0]), float(b[1]),
                float(g.get("eps", self.eps)), float(self.grad_scale))

    def sync_hyperparameters(self):
        """Upload lr / max_norm / betas / eps / grad_scale if they changed since the last upload (asynchronous, on the current
        stream).  step() calls it; a loop that REPLAYS a captured graph calls it itself after scheduler.step()."""
        hp = self._hyper()
        if hp == self.hp_now:
            return False
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ClipAdam: a hyper-parameter changed inside a graph capture (lr %s -> %s); change it outside "
                               "the graph and call sync_hyperparameters() before the replay" % (self.hp_now, hp))
        self.hp_turn ^= 1
        host = self.hp_host[self.hp_turn]              # two pinned blocks in turn: the previous upload may still be reading the other
        host.copy_(torch.tensor(hp, dtype=torch.float32))
        self.hp_dev.copy_(host, non_blocking=True)
        self.hp_now = hp
        return True

    # ---- checkpointing: the reference saves no optimizer state (train_detector_3D_angle.py:416-417); a resumed run here keeps
    # Adam's moments and bias correction
    def state_dict(self):
        return {"step": int(self.step_dev.item()), "m": [t.detach().clone() for t in self.m], "v": [t.detach().clone() for t in self.v],
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}],
                "max_norm": self.max_norm, "grad_scale": self.grad_scale}

    def load_state_dict(self, sd):
        if len(sd["m"]) != len(self.m) or any(a.shape != b.shape for a, b in zip(sd["m"], self.m)):
            raise ValueError("ClipAdam.load_state_dict: parameter list does not match")
        with torch.no_grad():
            for dst, src in zip(self.m, sd["m"]):
                dst.copy_(src)
            for dst, src in zip(self.v, sd["v"]):
                dst.copy_(src)
            self.step_dev.fill_(int(sd["step"]))
        self.step_count = int(sd["step"])
        self.param_groups[0].update(sd["param_groups"][0])
        self.max_norm = sd.get("max_norm", self.max_norm)
        self.grad_scale = sd.get("grad_scale", self.grad_scale)
        self.sync_hyperparameters()

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        """-> device tensor [1] with the pre-clip global gradient norm (clip_grad_norm_'s return value)."""
        lib = _hip.load()
        ptrs = []
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                raise RuntimeError("parameter without gradient")
            if not g.is_contiguous() or g.dtype != torch.float32:
                g = p.grad = g.float().contiguous()
            ptrs += (p.data_ptr(), g.data_ptr(), self.m[i].data_ptr(), self.v[i].data_ptr(), p.numel())
        capturing = torch.cuda.is_current_stream_capturing()
        tab = self.tables[self.turn]
        if tab["ptrs"] != ptrs:
            if capturing:
                raise RuntimeError("ClipAdam.step inside a graph capture needs the pointer table of the warm-up steps: run "
                                   "a few eager steps with persistent gradients (net.use_flat_gradients()) first")
            tab = self.tables[self.turn ^ 1]
            self.turn ^= 1
            if tab["event"] is not None:
                tab["event"].synchronize()            # the step that last used this pair has left the device
            tab["host"].copy_(torch.tensor(ptrs, dtype=torch.int64))
            tab["dev"].copy_(tab["host"], non_blocking=True)
            tab["ptrs"] = ptrs
        self.step_count += 1                          # host mirror (information only; the kernels read step_dev)
        self.sync_hyperparameters()                   # (raises inside a capture if a scheduler changed something since the warm-up)
        _hip.check(lib.rn_opt_clip_adam_hp(tab["dev"].data_ptr(), self.chunk_table.data_ptr(), self.n_chunks, self.hp_dev.data_ptr(),
                                           self.step_dev.data_ptr(), 1, self.ws.data_ptr(), self.total_norm.data_ptr(),
                                           _hip.stream()), "rn_opt_clip_adam_hp")
        if not capturing:
            if tab["event"] is None:
                tab["event"] = torch.cuda.Event()
            tab["event"].record()
        # the kernel updated the parameters through raw pointers: bump their version counters so that caches
        # keyed on (data_ptr, _version) -- the engine's packed weights -- see the change
        setter = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
        if setter is not None:
            setter(self.params, [p._version + 1 for p in self.params])
        else:
            for p in self.params:
                p.add_(0)
        return self.total_norm
