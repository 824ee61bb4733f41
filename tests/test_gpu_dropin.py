"""GPU: the drop-in import paths (same module / class names as the reference) and the Homography classes."""
import pickle
import sys

import numpy as np
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu


def test_directional_package_names(dev):
    import retinanet.model as m
    from retinanet import losses, utils, anchors
    from retinanet.model import resnet18, resnet50, batched_nms, nms          # noqa: F401
    net = resnet18(num_classes=8).to(dev)
    assert net.regressionModel.output.weight.shape == (9 * 12, 256, 3, 3)
    assert isinstance(net.focalLoss, losses.FocalLoss) and isinstance(net.regressBoxes, utils.BBoxTransform)
    a = anchors.Anchors()(torch.zeros(1, 3, 64, 96, device=dev))
    assert a.shape == (1, 1161, 4)
    with pytest.raises(RuntimeError):
        resnet50(num_classes=8, pretrained=True)
    # nms / batched_nms standalone
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60.]], device=dev)
    scores = torch.tensor([0.9, 0.8, 0.7], device=dev)
    assert m.nms(boxes, scores, 0.5).tolist() == [0, 2]
    assert m.batched_nms(boxes, scores, torch.tensor([0, 1, 0], device=dev), 0.5).tolist() == [0, 1, 2]


def test_homography_dropin(dev, golden):
    import homography as hgmod
    z = golden("homography")
    names, state, cam, _, _ = gc.homography_inputs()

    def make(P, H):
        hg = hgmod.Homography()
        hg.correspondence = {n: {"P": P[i], "H": H[i], "H_inv": np.linalg.inv(H[i])} for i, n in enumerate(names)}
        hg.default_correspondence = names[0]
        return hg
    hg = make(z["P"], z["H"])
    cams = [names[i] for i in cam]
    assert np.array_equal(hg.state_to_space(state).numpy(), z["space"])
    im = hg.state_to_im(state, name=cams)
    assert im.dtype == torch.float64 and not im.is_cuda                       # CPU in -> CPU out, like the reference
    assert np.allclose(im.numpy(), z["im_list"], rtol=1e-9, atol=1e-9)
    assert np.allclose(hg.state_to_im(state, name="p1c3").numpy(), z["im_one"], rtol=1e-9, atol=1e-9)
    assert np.allclose(hg.state_to_im(state).numpy(), z["im_default"], rtol=1e-9, atol=1e-9)
    h = state[:, 4]
    back = hg.im_to_state(im, name=cams, heights=h)
    assert back.dtype == torch.float32
    assert np.allclose(back.numpy(), z["back_state_list"], rtol=1e-6, atol=1e-5)
    assert np.allclose(hg.im_to_space(im, name=cams, heights=h).numpy(), z["back_space_list"], rtol=1e-9, atol=1e-7)
    assert np.array_equal(hg.guess_heights(["sedan", "semi", 3, "nonsense", "trailer", "truck (other)"]).numpy(),
                          z["guess_heights"])
    assert np.allclose(hg.height_from_template(im, h, im * 1.07 + 3.0).numpy(), z["height_from_template"], rtol=1e-6)
    wr = hgmod.Homography_Wrapper(hg1=hg, hg2=make(z["P2"], z["H2"]))
    assert np.allclose(wr.state_to_im(state, name=cams).numpy(), z["wr_im_list"], rtol=1e-9, atol=1e-9)
    assert np.allclose(wr.state_to_im(state, name="p2c4").numpy(), z["wr_im_one"], rtol=1e-9, atol=1e-9)
    wb = wr.im_to_state(torch.from_numpy(z["wr_im_list"]), name=cams, heights=h)
    assert np.allclose(wb.numpy(), z["wr_back_state_list"], rtol=1e-6, atol=1e-5)
    # the reference pickles these objects
    hg2 = pickle.loads(pickle.dumps(hg))
    assert np.allclose(hg2.state_to_im(state, name=cams).numpy(), z["im_list"], rtol=1e-9, atol=1e-9)


def test_flat2d_package_names(dev):
    import importlib
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-playground_amd", "flat2d")
    saved = {k: v for k, v in sys.modules.items() if k == "retinanet" or k.startswith("retinanet.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, root)
    try:
        m = importlib.import_module("retinanet.model")
        net = m.resnet18(num_classes=8).to(dev)
        assert net.regressionModel.output.weight.shape == (9 * 4, 256, 3, 3)
        assert not net.directional
        fl = importlib.import_module("retinanet.losses").FocalLoss()
        assert not fl.directional
    finally:
        sys.path.remove(root)
        for k in [k for k in sys.modules if k == "retinanet" or k.startswith("retinanet.")]:
            del sys.modules[k]
        sys.modules.update(saved)
