"""CPU checks of the numpy restatement of the TF-side local aggregation operators (oracle/local_aggregation_oracle.py).
TensorFlow cannot run here (parity unpinned by execution), so what CAN be checked on CPU is internal consistency: the analytic
gradients the GPU tests compare against agree with finite differences of the forward restatement, the shadow-row and
padding-count conventions hold, and PosPool's channel sharing matches the reference's reshape
(/root/reference/tensorflow/models/local_aggregation_operators.py:227-231)."""
import numpy as np
import pytest

from oracle import local_aggregation_oracle as LA


def _case(C, seed=0, n0=50, n=20, K=7):
    rng = np.random.default_rng(seed)
    s = rng.uniform(0, 1, (n0, 3)).astype(np.float32)
    q = (s[:n] + 0.01).astype(np.float32)
    idx = rng.integers(0, n0 + 1, (n, K)).astype(np.int32)           # n0 = shadow neighbour
    idx[0, :] = n0                                                    # a point without any neighbour
    f = rng.normal(size=(n0, C)).astype(np.float32)
    return q, s, idx, f, rng


@pytest.mark.parametrize("pe,C", [("sin_cos", 12), ("sin_cos", 9), ("xyz", 6), ("direction_exp_-d", 9), ("direction_d", 20), ("two_order", 18),
                                  ("three_order", 36), ("three_order", 9), ("one", 5), ("exp_-d", 4), ("distance", 3)])
@pytest.mark.parametrize("red", ["sum", "mean", "max"])
def test_pospool_gradient_matches_finite_differences(pe, C, red):
    q, s, idx, f, rng = _case(C, seed=C)
    out, geo, agg = LA.pospool(q, s, idx, f, 0.1, pe, red)
    assert out.shape == (len(q), C) and np.isfinite(out).all()
    go = rng.normal(size=out.shape).astype(np.float32)
    g = LA.pospool_grad_features(q, s, idx, f, 0.1, go, pe, red)
    for (i, c) in [(3, C - 1), (11, 0), (49, C // 2)]:
        f2 = f.copy(); f2[i, c] += 1e-2
        fd = ((LA.pospool(q, s, idx, f2, 0.1, pe, red)[0].astype(np.float64) - out) * go).sum() / 1e-2
        assert abs(fd - g[i, c]) < 2e-2 * max(1.0, abs(g[i, c])), (pe, red, i, c, fd, g[i, c])


def test_pospool_conventions():
    q, s, idx, f, _ = _case(12)
    out, geo, _ = LA.pospool(q, s, idx, f, 0.1, "sin_cos", "mean")
    assert np.all(out[0] == 0)                                        # only shadow neighbours: zero features / (0 + 1e-5)
    assert geo.shape[-1] == 12                                        # sin_cos: one embedding value per channel
    assert np.all(LA.pospool(q, s, idx, f, 0.1, "sin_cos", "max")[0][0] == -65535.0)      # :243-249
    _, geo, _ = LA.pospool(q, s, idx, f, 0.1, "xyz", "sum")
    assert geo.shape[-1] == 3                                         # 'xyz': channels [0,4) x, [4,8) y, [8,12) z
    with pytest.raises(ValueError):
        LA.pospool(q, s, idx, f[:, :10], 0.1, "sin_cos", "mean")


def test_adaptive_weight_and_kpconv_gradients_match_finite_differences():
    q, s, idx, f, rng = _case(6, seed=3)
    W = rng.normal(size=(3, 6)).astype(np.float32); b = rng.normal(size=6).astype(np.float32)
    out = LA.adaptive_weight(q, s, idx, f, 0.1, W, b, "mean")
    go = rng.normal(size=out.shape).astype(np.float32)
    gf, gW, gb = LA.adaptive_weight_grads(q, s, idx, f, 0.1, W, b, go, "mean")
    W2 = W.copy(); W2[1, 2] += 1e-2
    fd = ((LA.adaptive_weight(q, s, idx, f, 0.1, W2, b, "mean").astype(np.float64) - out) * go).sum() / 1e-2
    assert abs(fd - gW[1, 2]) < 2e-2 * max(1.0, abs(gW[1, 2]))
    kp = (rng.normal(size=(5, 3)) * 0.05).astype(np.float32); kw = rng.normal(size=(5, 6)).astype(np.float32)
    out = LA.kpconv(q, s, idx, f, kp, kw, 0.08, "linear", "sum")
    go = rng.normal(size=out.shape).astype(np.float32)
    gf, gkw = LA.kpconv_grads(q, s, idx, f, kp, kw, 0.08, go, "linear", "sum")
    kw2 = kw.copy(); kw2[2, 4] += 1e-2
    fd = ((LA.kpconv(q, s, idx, f, kp, kw2, 0.08, "linear", "sum").astype(np.float64) - out) * go).sum() / 1e-2
    assert abs(fd - gkw[2, 4]) < 2e-2 * max(1.0, abs(gkw[2, 4]))
