"""bcnn_hip_cost_metric: the cost node's scalar on the device against the reference's host loops
(bcnn_compute_error, bcnn_cost_layer.c:142-244) restated in numpy -- every metric, ties and the FLT_MIN start of
the arg-max included."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FLT_MIN = np.float32(1.17549435e-38)


def _host(metric, pred, label, grad):
    B, per = pred.shape
    acc = 0.0
    if metric == 0:
        for i in range(B):
            pm, best = FLT_MIN, 0
            for j in range(per):
                if pred[i, j] > pm:
                    pm, best = pred[i, j], j
            if label[i, best] == 0:
                acc += 1.0
        return np.float32(acc)
    if metric == 1:
        q = np.clip(pred, np.float32(1e-8), np.float32(1.0) - np.float32(1e-8))
        return np.float32((-np.log(q.astype(np.float64)))[label > 0].sum())
    if metric in (2, 3, 4):
        acc = (grad.astype(np.float64) ** 2).sum()
        return np.float32(acc / per if metric == 3 else acc)
    for i in range(B):
        t = (pred[i] > 0.5).astype(np.float32)
        n = int((label[i] * t).astype(np.int32).sum())
        d = int((label[i] + t).astype(np.int32).sum())
        acc += float(np.float32(2.0 * n + 1.0) / np.float32(d + 1.0))
    return np.float32(acc)


@pytest.mark.parametrize("metric", range(6))
@pytest.mark.parametrize("B,per", [(128, 1000), (3, 7), (16, 64)])
def test_device_cost_metric_matches_host_loop(metric, B, per):
    from bcnn_amd import _lib
    L = _lib.load()
    rs = np.random.RandomState(metric * 100 + B)
    pred = rs.uniform(0, 1, (B, per)).astype(np.float32)
    pred[0, :] = -1.0                      # nothing above FLT_MIN: the arg-max stays at index 0
    if per > 3:
        pred[1, 2] = pred[1, 3] = 2.0      # tie: the first maximum wins
    label = np.zeros((B, per), np.float32)
    label[np.arange(B), rs.randint(0, per, B)] = 1.0
    label[1, 2 if per > 3 else 0] = 1.0
    grad = pred - label
    d = lambda a: torch.from_numpy(a).cuda()
    p_d, l_d, g_d = d(pred), d(label), d(grad)
    out = torch.full((1,), -7.0, device="cuda")
    L.bcnn_hip_cost_metric(metric, p_d.data_ptr(), l_d.data_ptr(), g_d.data_ptr(), B, per, out.data_ptr())
    torch.cuda.synchronize()
    want = float(_host(metric, pred, label, grad))
    got = float(out.item())
    if metric in (0, 5):
        assert got == pytest.approx(want, rel=1e-6)
    else:
        assert got == pytest.approx(want, rel=1e-6)
