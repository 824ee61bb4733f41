"""The N-GPU tools with the REAL engine, rehearsed on one GPU: two ranks share cuda:0 and talk over gloo
(RN_REHEARSE_ONE_GPU, retinanet_mi355x/ddp.py); on the driver's 8-GPU node the same commands run over RCCL.

* tools/train_ddp.py -- the reference trainer's loop (train_detector_3D_angle.py:337-417): epochs of forward + losses + backward
  with the gradient all-reduce inside + clip + Adam, ReduceLROnPlateau on the mean loss, per-epoch checkpoints; both ranks end
  with the same weights, and a checkpoint re-saved with DataParallel's ``module.`` prefix resumes.
* tools/bench_infer.py --gpus 2 -- BASELINE configs[3]'s camera sharding (retinanet_mi355x/multicam.py): the merged, parsed
  result of two ranks equals the one-rank run of the same protocol on the same frames.
Child processes: 3 on the card at a time."""
import collections
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=600, **env):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    r = subprocess.run([sys.executable] + cmd, env=e, capture_output=True, text=True, timeout=timeout, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_train_ddp_two_ranks_on_one_gpu(dev, tmp_path):
    out = str(tmp_path / "ck")
    common = ["tools/train_ddp.py", "--arch", "resnet18", "--height", "72", "--width", "104", "--batch", "2", "--iters", "3", "--out", out]
    res = _run(common + ["--gpus", "2", "--epochs", "2"], RN_REHEARSE_ONE_GPU="1")
    assert res["ranks"] == 2 and res["backend"] == "gloo" and res["weights_identical_across_ranks"] is True
    assert [e["iterations"] for e in res["epochs"]] == [3, 3] and all(e["skipped"] == 0 for e in res["epochs"])
    assert all(e["mean_loss"] == e["mean_loss"] and e["mean_loss"] > 0 for e in res["epochs"])        # finite
    assert res["checkpoints"] == ["corrected_data_e0.pt", "corrected_data_e1.pt"]
    # the reference's multi-GPU run saves DataParallel's keys (train_detector_3D_angle.py:416-417): such a file resumes
    sd = torch.load(os.path.join(out, "corrected_data_e1.pt"), weights_only=True)
    assert len(sd) == 156                                                                              # ResNet-18's state_dict
    dp = collections.OrderedDict(("module." + k, v) for k, v in sd.items())
    torch.save(dp, os.path.join(out, "dataparallel_style.pt"))
    res1 = _run(common + ["--gpus", "1", "--epochs", "1", "--resume", os.path.join(out, "dataparallel_style.pt")])
    assert res1["ranks"] == 1 and res1["epochs"][0]["iterations"] == 3
    assert res1["epochs"][0]["mean_loss"] < res["epochs"][0]["mean_loss"] * 1.5                        # a trained start, not a crash course


def test_camera_sharding_two_ranks_equal_one(dev):
    """Same four 1080p frames, same detector: one rank running all four cameras through the sharded protocol against two ranks
    with two cameras each (calls of one camera, so that the per-call 10 000-candidate cap sees the same frames either way)."""
    common = ["tools/bench_infer.py", "--cams", "4", "--batch", "1", "--iters", "1", "--checksum"]
    one = _run(common + ["--sharded"])
    two = _run(common + ["--gpus", "2"], RN_REHEARSE_ONE_GPU="1")
    assert one["ranks"] == 1 and two["ranks"] == 2 and two["cameras_per_rank"] == [2, 2]
    assert one["detections_kept"] > 100
    assert two["detections_kept"] == one["detections_kept"] and two["objects_parsed"] == one["objects_parsed"]
    assert one["objects_parsed"] > 0 and two["parsed_states_checksum"] == one["parsed_states_checksum"]   # the same states, bit for bit
