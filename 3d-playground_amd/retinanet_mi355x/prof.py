"""HIP-event timing of individual kernel launches on the current stream (used by bench.py for the roofline
object: achieved = algorithmic work / measured launch duration)."""
import collections

import torch

ACTIVE = None          # a KernelTimer while bench.py is measuring, else None
BY_SHAPE = False       # tools/profile_layers.py: key the bf16 conv launches by layer shape instead of by kernel family


class KernelTimer:
    def __init__(self):
        self.rows = []                 # (kind, work, start event, end event)

    def launch(self, kind, work, fn, nbytes=0.0):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.rows.append((kind, work, e0, e1, nbytes))
        return out

    def summary(self):
        """{kind: dict(launches, ms_total, ms_avg, work_total, bytes_total)} after a device sync.  bytes_total: ALGORITHMIC bytes
        of the launches (every operand once: inputs, weights, addend / mask, result), what roofline.traffic is compared with."""
        torch.cuda.synchronize()
        agg = collections.OrderedDict()
        for kind, work, e0, e1, nbytes in self.rows:
            a = agg.setdefault(kind, {"launches": 0, "ms_total": 0.0, "work_total": 0.0, "bytes_total": 0.0})
            a["launches"] += 1
            a["ms_total"] += e0.elapsed_time(e1)
            a["work_total"] += work
            a["bytes_total"] += nbytes
        for a in agg.values():
            a["ms_avg"] = a["ms_total"] / max(a["launches"], 1)
        return agg


def timed(kind, work, fn, nbytes=0.0):
    if ACTIVE is None:
        return fn()
    return ACTIVE.launch(kind, work, fn, nbytes)


def sources_digest():
    """sha256 over the kernel sources and the launch schedule (csrc/*.hip, csrc/*.h, include/*.h, engine.py, conv.py): what a
    committed PMC collection (profiles/pmc_traffic.json) is stamped with, and what bench.py compares it against before it quotes
    that collection's bytes beside THIS run's times (there is no git on the GPU box to ask for a commit)."""
    import glob
    import hashlib
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(here))
    files = sorted(glob.glob(os.path.join(here, "..", "csrc", "*.hip")) + glob.glob(os.path.join(here, "..", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "include", "*.h")) + [os.path.join(here, "engine.py"), os.path.join(here, "conv.py")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()
