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
