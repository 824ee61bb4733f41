"""GPU parity of the frame ingest: rn_frame_ingest against oracle/ingest.py (the restated
F.to_tensor + F.normalize of the reference's loaders), bit for bit, in both output layouts, with and without the
BGR->RGB swap, on sizes that take the 4-pixel path and the byte path; and the model fed raw uint8 frames against the
model fed the oracle's float tensor."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import ingest as oing
from retinanet_mi355x import modules, ops, synth

pytestmark = pytest.mark.gpu


def _frames(B, H, W, seed):
    u = synth.uniform((B, H, W, 3), seed)
    return torch.from_numpy((u * 256).astype(np.uint8))


@pytest.mark.parametrize("shape", [(2, 36, 52), (1, 17, 23), (3, 9, 7), (1, 270, 480)])
@pytest.mark.parametrize("swap", [False, True])
def test_ingest_matches_oracle_bitwise(dev, shape, swap):
    B, H, W = shape
    f = _frames(B, H, W, seed=31 + H)
    want = oing.to_tensor_normalize(f, swap_rb=swap)
    got = ops.frame_ingest(f.to(dev), swap_rb=swap)
    assert got.shape == (B, 3, H, W) and got.dtype == torch.float32
    assert torch.equal(got.cpu(), want)
    got4 = ops.frame_ingest(f.to(dev), swap_rb=swap, nhwc4=True)
    assert got4.shape == (B, H, W, 4)
    assert torch.equal(got4[..., :3].cpu(), want.permute(0, 2, 3, 1))
    assert float(got4[..., 3].abs().max()) == 0.0


def test_ingest_custom_statistics_and_single_frame(dev):
    f = _frames(1, 12, 20, seed=5)[0]
    mean, std = (0.1, 0.2, 0.3), (0.5, 0.25, 2.0)
    want = oing.to_tensor_normalize(f[None], mean=mean, std=std)
    assert torch.equal(ops.frame_ingest(f.to(dev), mean=mean, std=std).cpu(), want)
    with pytest.raises(RuntimeError, match="uint8"):
        ops.frame_ingest(f.float().to(dev))


def test_model_takes_uint8_frames(dev):
    """model(uint8 frames) == model(to_tensor + normalize of the same frames), LOCALIZE outputs bit for bit."""
    sd, _, _ = gc.model_inputs("resnet18", True)
    net = modules.resnet18(num_classes=4)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    H, W = gc.MODEL_HW
    f = _frames(2, H, W, seed=77)
    for swap in (False, True):
        net.ingest_swap_rb = swap
        b_u8, c_u8 = net(f.to(dev), LOCALIZE=True)
        b_f, c_f = net(oing.to_tensor_normalize(f, swap_rb=swap).to(dev), LOCALIZE=True)
        assert torch.equal(b_u8, b_f) and torch.equal(c_u8, c_f)
