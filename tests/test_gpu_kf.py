"""GPU parity of the tracker's Kalman filter: the drop-in Torch_KF (rn_kf_view / rn_kf_predict /
rn_kf_update) against the goldens produced by running the reference's own class (tests/golden/kf.npz) and the oracle.
fp32 algebra: 1e-5 relative for predict / view (same operations, possibly fused differently), 1e-4 for update (a 5x5
inverse by a different algorithm)."""
import numpy as np
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu


def _filter(dev, default=False):
    from util_track.kf import Torch_KF
    INIT, det, directions, times, speed, upd_ids, z, dts = gc.kf_inputs()
    kf = Torch_KF(dev) if default else Torch_KF(dev, INIT={k: v.clone() for k, v in INIT.items()}, ADD_MEAN_R=True)
    ids = list(range(100, 100 + len(det)))
    kf.add(det.clone(), ids, directions.clone(), times.clone())
    return kf, ids, (speed, upd_ids, z, dts)


def close(a, b, rtol, atol=1e-5):
    assert np.allclose(a.detach().cpu().numpy(), b, rtol=rtol, atol=atol), float(np.abs(a.detach().cpu().numpy() - b).max())


def test_kf_sequence_golden(dev, golden):
    z = golden("kf")
    kf, ids, (speed, upd_ids, meas, dts) = _filter(dev)
    kf.X[:, 5] = speed.to(dev)
    close(kf.X, z["X0"], 0), close(kf.P, z["P0"], 0)
    assert kf.X.is_cuda and kf.T.dtype == torch.float64
    for tag, dt in (("1", None), ("2", 0.05), ("3", dts)):
        kf.predict() if dt is None else kf.predict(dt=dt.clone() if isinstance(dt, torch.Tensor) else dt)
        close(kf.X, z["X" + tag], 1e-6), close(kf.P, z["P" + tag], 1e-5), close(kf.T, z["T" + tag], 1e-15, 0)
    idl, v = kf.view(dt=dts.clone(), with_direction=True)
    assert idl == ids
    close(v, z["view_dir"], 1e-6)
    close(kf.view(dt=1 / 30.0)[1], z["view_plain"], 1e-6)
    close(kf.objs()[1], z["X3"], 1e-6)
    kf.update(meas.clone(), [ids[i] for i in upd_ids])
    close(kf.X, z["X4"], 1e-4, 1e-4), close(kf.P, z["P4"], 1e-4, 1e-4)
    kf.remove([ids[0], ids[5]])
    close(kf.X, z["X5"], 1e-4, 1e-4), close(kf.T, z["T5"], 1e-15, 0)
    assert np.array_equal(np.array(kf.view()[0]), z["ids5"])


def test_kf_default_constructor_golden(dev, golden):
    """The diagonal default filter (huge P0 = 10000 I: the first update is an ill-conditioned inverse)."""
    z = golden("kf")
    kf, ids, (speed, upd_ids, meas, dts) = _filter(dev, default=True)
    kf.predict()
    kf.update(meas.numpy(), [ids[i] for i in upd_ids])                       # numpy measurements, as the tracker passes
    close(kf.X, z["Xd"], 1e-4, 1e-3), close(kf.P, z["Pd"], 1e-3, 1e-3)


def test_kf_get_dt_and_errors(dev):
    kf, ids, (speed, upd_ids, meas, dts) = _filter(dev)
    t0 = kf.T.clone()
    assert torch.allclose(kf.get_dt(12.5), 12.5 - t0)
    d = kf.get_dt([11.0, 11.5], idxs=[3, 7])
    assert abs(float(d[3]) - (11.0 - float(t0[3]))) < 1e-6 and abs(float(d[0]) - 1 / 30.0) < 1e-7
    with pytest.raises(RuntimeError, match="twice"):
        kf.update(meas[:2], [ids[1], ids[1]])
    from util_track.kf import Torch_KF
    with pytest.raises(RuntimeError, match="GPU"):
        Torch_KF(torch.device("cpu"))
