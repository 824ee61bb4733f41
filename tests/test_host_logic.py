"""CPU tests of the host-side logic around the HIP path: the engine's flat gradient layout, bench.py's rank launcher,
the util_track namespace merge with a reference checkout, and that the committed golden recipe still regenerates the
committed fixtures (the last two only where /root/reference exists: never on the GPU box)."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "3d-playground_amd")
REF = "/root/reference"


@pytest.mark.parametrize("arch", ["resnet18", "resnet50", "resnet101"])
def test_flat_gradient_plan_covers_every_parameter_once(arch):
    from retinanet_mi355x import arch as A, engine
    eng = engine.Engine(arch, 8, 12)
    shapes = A.state_dict_shapes(arch, 8, 12)
    for L in eng.layers.values():
        L.weight = torch.empty(shapes[L.spec.name + ".weight"])
    order = eng.finish_order()
    assert order[0] == "regressionModel.output" and order[-1] == "conv1" and len(set(order)) == len(order)
    eng.set_flat_grads(8 << 20)
    plan = eng._flat_plan(torch.device("cpu"))
    assert sorted(plan["slots"]) == sorted(eng.param_names)
    spans = sorted((o, o + n) for o, n, _ in plan["slots"].values())
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "slots overlap"
    assert all(o % 64 == 0 for o, _ in spans)
    for name, (o, n, shp) in plan["slots"].items():
        assert tuple(shp) == tuple(shapes[name]) and n == int(np.prod(shp))
    b = plan["buckets"]
    assert b[0][0] == 0 and b[-1][1] == plan["total"] and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    # every bucket is full except the last one -- and, when the exposed tail was cut down (set_flat_grads: tail_bytes), the
    # piece that was in front of it
    assert all(4 * (e - s) >= (8 << 20) for s, e, _ in b[:-2])
    assert 4 * (b[-1][1] - b[-1][0]) <= (8 << 20) + (6 << 20)
    # a bucket ends with the layer that completes it, in finish order
    pos = {n: i for i, n in enumerate(order)}
    assert [pos[x[2]] for x in b] == sorted(pos[x[2]] for x in b)


def test_bench_refuses_to_report_fewer_gpus_than_asked():
    """`python bench.py --gpus 2` without a torchrun environment starts its own ranks -- and on a node with fewer GPUs
    exits non-zero instead of printing a single-GPU number as n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RN_REHEARSE_ONE_GPU")}
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the GPUs")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "n_gpus" in r.stderr
    assert '"metric"' not in r.stdout


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_util_track_merges_with_the_reference_namespace():
    """with 3d-playground_amd first on sys.path, util_track.kf is the drop-in while the tracker's
    other imports (util_track.mp_loader / mp_writer, MC3D_crop_tracker.py:22-24) still resolve inside the reference."""
    code = ("import sys, importlib.util as u\n"
            "sys.path[:0] = [%r, %r]\n"
            "for m in ('util_track.kf', 'util_track.mp_loader', 'util_track.mp_writer'):\n"
            "    print(m, u.find_spec(m).origin)\n" % (PKG, REF))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines())
    assert got["util_track.kf"].startswith(PKG + os.sep)
    assert got["util_track.mp_loader"].startswith(REF + os.sep)
    assert got["util_track.mp_writer"].startswith(REF + os.sep)
    assert not os.path.exists(os.path.join(PKG, "util_track", "__init__.py"))


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
@pytest.mark.timeout(600)
def test_golden_recipe_regenerates_the_committed_fixtures(tmp_path):
    """tools/make_golden.py, run as committed, reproduces every tests/golden/*.npz array: integers, strings and hashes bit
    for bit; floating-point arrays bit for bit too unless torch's multi-threaded CPU reductions took another order in
    this run (seen on gradient sums, 1 ulp of a float32) -- those must agree to 1e-6 of the array's max magnitude."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "make_golden.py"), "--out", str(tmp_path)],
                       capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-3000:]
    committed = os.path.join(REPO, "tests", "golden")
    names = sorted(f for f in os.listdir(committed) if f.endswith(".npz"))
    assert names == sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    texts = sorted(f for f in os.listdir(committed) if f.endswith(".csv"))
    assert texts == sorted(f for f in os.listdir(tmp_path) if f.endswith(".csv")) and texts
    for fn in texts:
        assert open(os.path.join(tmp_path, fn), "rb").read() == open(os.path.join(committed, fn), "rb").read(), fn
    for fn in names:
        a, b = np.load(os.path.join(tmp_path, fn)), np.load(os.path.join(committed, fn))
        assert sorted(a.files) == sorted(b.files), fn
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (fn, k)
            if a[k].tobytes() == b[k].tobytes():
                continue
            assert a[k].dtype.kind == "f", (fn, k)
            x, y = a[k].astype(np.float64), b[k].astype(np.float64)
            assert np.array_equal(np.isnan(x), np.isnan(y)), (fn, k)
            assert np.nanmax(np.abs(x - y)) <= 1e-6 * np.nanmax(np.abs(y)), (fn, k, np.nanmax(np.abs(x - y)))


def test_custom_operators_are_registered_and_have_no_cpu_kernel():
    """north_star: the HIP entry points are exposed as PyTorch custom ops (torch.ops.retinanet_mi355x.*) with schemas and
    shape-only (fake) implementations; there is no CPU kernel behind them."""
    from retinanet_mi355x import torch_ops
    for name in torch_ops.OPERATORS:
        op = getattr(torch.ops.retinanet_mi355x, name)
        assert "retinanet_mi355x::" + name in str(op.default._schema)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.retinanet_mi355x.decode_dir(torch.zeros(1, 10, 4), torch.zeros(2, 10, 12))
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():                              # shape inference without a device
        r = torch.ops.retinanet_mi355x.decode_dir(torch.empty(1, 10, 4, device="cuda"), torch.empty(2, 10, 12, device="cuda"))
        assert tuple(r.shape) == (2, 10, 20)
        l, ws = torch.ops.retinanet_mi355x.focal_loss_fwd(torch.empty(2, 100, 8, device="cuda"), torch.empty(2, 100, 12, device="cuda"),
                                                          torch.empty(1, 100, 4, device="cuda"), torch.empty(2, 5, 27, device="cuda"), True)
        assert tuple(l.shape) == (3,) and ws.dtype == torch.uint8


def test_data_parallel_checkpoint_keys_load_without_their_prefix():
    """train_detector_3D_angle.py:416-417 saves nn.DataParallel(model).state_dict(): every key is 'module.<name>', and the
    reference strips the prefix before loading (to_cpu, :39-59).  The drop-in's load_state_dict takes such a dict as it is;
    a dict in which only SOME keys carry the prefix is not one and keeps strict's errors."""
    import collections
    from retinanet_mi355x import modules
    net = modules.resnet18(num_classes=3)
    sd = net.state_dict()
    dp = collections.OrderedDict(("module." + k, (v.float() + 1).to(v.dtype)) for k, v in sd.items())
    dp._metadata = collections.OrderedDict((("module." + k) if k else "module", v) for k, v in sd._metadata.items())
    res = net.load_state_dict(dp)
    assert not res.missing_keys and not res.unexpected_keys
    now = net.state_dict()
    assert list(now) == list(sd) and all(torch.equal(now[k], dp["module." + k]) for k in sd)
    # round trip: what a DataParallel wrapper of THIS model would save loads back into a fresh model
    wrapped = collections.OrderedDict(("module." + k, v) for k, v in now.items())
    other = modules.resnet18(num_classes=3)
    other.load_state_dict(wrapped)
    assert all(torch.equal(other.state_dict()[k], now[k]) for k in now)
    half = collections.OrderedDict(dp)
    half["conv1.weight"] = sd["conv1.weight"]
    with pytest.raises(RuntimeError, match="Missing key|Unexpected key"):
        net.load_state_dict(half)
    # strict=False still filters by name only after the prefix is gone (ImageNet backbone into a detector)
    part = collections.OrderedDict((k, v) for k, v in dp.items() if k.startswith("module.layer1."))
    res = net.load_state_dict(part, strict=False)
    assert not res.unexpected_keys and res.missing_keys


def test_run_time_options_are_read_once_and_range_checked():
    """rn_get_option / rn_set_option (include/retinanet_mi355x.h: RN_OPT_*): defaults, the environment read at first use only (the launch
    paths never call getenv), values outside an option's range refused.  No GPU needed: host-side state of the library."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
os.environ["RN_MF16_MIN"] = "7"
from retinanet_mi355x import _hip, conv
lib = _hip.load()
assert conv.get_option(conv.OPT_MF16) == 1 and conv.get_option(conv.OPT_BF16_P8) == 1 and conv.get_option(conv.OPT_FP8_P8) == 1
assert conv.get_option(conv.OPT_WGRAD_ONCE) == 1 and conv.get_option(conv.OPT_SPLITK) == 1 and conv.get_option(conv.OPT_DETERMINISTIC) == 0
assert conv.get_option(conv.OPT_MF16_MIN) == 7                      # from the environment, at first use
os.environ["RN_MF16_MIN"] = "9"
assert conv.get_option(conv.OPT_MF16_MIN) == 7                      # ... and never again
conv.set_option(conv.OPT_MF16_MIN, 3)
assert conv.get_option(conv.OPT_MF16_MIN) == 3
conv.set_option(conv.OPT_BF16_P8, 2)
assert lib.rn_set_option(conv.OPT_BF16_P8, 3) != 0 and conv.get_option(conv.OPT_BF16_P8) == 2      # out of range: refused, unchanged
assert lib.rn_set_option(conv.OPT_MF16, 2) != 0 and lib.rn_set_option(99, 1) != 0 and lib.rn_get_option(99) == -1
assert lib.rn_set_option(conv.OPT_MF16_MIN, -1) != 0
print("ok")
''' % PKG
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout, r.stderr[-2000:])


def test_winograd_planes_are_an_odd_number_of_blocks_apart():
    """conv.wino_tpad (round 5): rows of one of the 36 planes of V / M / Z -- a multiple of 256 (no GEMM tile straddles two planes), at least
    the tile count, and an ODD multiple (an even one puts the planes a multiple of 512 KiB apart for 256 channels: the output transform,
    which combines one value of each plane, loses 8 % of its bandwidth -- profiles/r05_wgrad_stream.txt's neighbour r05_mf16_bounds.txt)."""
    from retinanet_mi355x import conv
    for T in (1, 255, 256, 257, 511, 512, 513, 4080, 16320, 21896, 22016, 22017, 100000):
        tp = conv.wino_tpad(T)
        assert tp >= T and tp % 256 == 0 and (tp // 256) % 2 == 1 and tp - T < 512, (T, tp)
    class _S:                                                   # (a stream's identity is its handle)
        def __init__(self, h):
            self.cuda_stream = h
    conv.SIDE_HELD.setdefault(1, []).append((None,))
    conv.SIDE_HELD.setdefault(2, []).append((None,))
    conv.side_release(_S(1))
    assert list(conv.SIDE_HELD) == [2]                          # another engine's operands stay held
    conv.side_release()
    assert conv.SIDE_HELD == {}
