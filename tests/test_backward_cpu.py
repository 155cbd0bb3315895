"""CPU suite, next-row N2: the oracle's gradients of the filter stack against the finite-difference Jacobian of its own
forward -- the method of the reference's only gradient test (nn.Jacobian.testJacobian, tests/test_cascad.lua:21-25):
every output unit's gradient w.r.t. every input / parameter, central differences."""
import numpy as np
import pytest

from tests import oracle as orc


def _jacobian_fd(f, x, eps):
    """d f(x) / d x by central differences: [x.size, f(x).size]"""
    x = x.astype(np.float32).copy()
    n_out = f(x).size
    J = np.zeros((x.size, n_out), np.float64)
    flat = x.reshape(-1)
    for k in range(flat.size):
        o = flat[k]
        flat[k] = o + eps
        hi = f(x).astype(np.float64).reshape(-1)
        flat[k] = o - eps
        lo = f(x).astype(np.float64).reshape(-1)
        flat[k] = o
        J[k] = (hi - lo) / (2 * eps)
    return J


@pytest.mark.parametrize("kH,kW,use_map", [(3, 3, False), (1, 5, False), (4, 1, False), (3, 2, True)])
def test_convolution_gradients_match_finite_differences(kH, kW, use_map):
    rng = np.random.default_rng(kH * 10 + kW)
    nIn, nOut, H, W = 3, 4, 6, 7
    x = rng.standard_normal((nIn, H, W)).astype(np.float32)
    b = rng.standard_normal(nOut).astype(np.float32)
    if use_map:
        conn = np.array([(1, 1), (3, 1), (2, 2), (1, 3), (2, 3), (3, 4), (1, 4)], np.int32)
        w = rng.standard_normal((len(conn), kH, kW)).astype(np.float32)
        fwd = lambda xx, ww=w, bb=b: orc.spatial_convolution_map(xx, ww, bb, conn, nOut)
        fwd_w = lambda ww: orc.spatial_convolution_map(x, ww, b, conn, nOut)
        fwd_b = lambda bb: orc.spatial_convolution_map(x, w, bb, conn, nOut)
    else:
        conn = None
        w = rng.standard_normal((nOut, nIn, kH, kW)).astype(np.float32)
        fwd = lambda xx: orc.spatial_convolution(xx, w, b)
        fwd_w = lambda ww: orc.spatial_convolution(x, ww, b)
        fwd_b = lambda bb: orc.spatial_convolution(x, w, bb)
    out = fwd(x)
    Jx, Jw, Jb = _jacobian_fd(fwd, x, 0.25), _jacobian_fd(fwd_w, w, 0.25), _jacobian_fd(fwd_b, b, 0.25)
    Bx, Bw, Bb = np.zeros_like(Jx), np.zeros_like(Jw), np.zeros_like(Jb)
    for j in range(out.size):
        go = np.zeros(out.size, np.float32)
        go[j] = 1
        gi, gw, gb = orc.spatial_convolution_backward(x, w, go.reshape(out.shape), conn=conn, nOut=nOut)
        Bx[:, j], Bw[:, j], Bb[:, j] = gi.reshape(-1), gw.reshape(-1), gb.reshape(-1)
    assert np.abs(Jx - Bx).max() < 1e-5 and np.abs(Jw - Bw).max() < 1e-5 and np.abs(Jb - Bb).max() < 1e-5   # test_cascad.lua:23 precision
    # accumulation and scale: accGradParameters adds scale * grad to what is there
    go = rng.standard_normal(out.shape).astype(np.float32)
    _, gw1, gb1 = orc.spatial_convolution_backward(x, w, go, conn=conn, nOut=nOut, scale=0.5)
    _, gw2, gb2 = orc.spatial_convolution_backward(x, w, go, conn=conn, nOut=nOut, scale=1.0)
    assert np.allclose(2 * gw1, gw2, rtol=1e-6, atol=1e-6) and np.allclose(2 * gb1, gb2, rtol=1e-6, atol=1e-6)


def test_tanh_softmax_logsoftmax_gradients_match_finite_differences():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 7)).astype(np.float32)
    # tanh
    J = _jacobian_fd(orc.tanh, x, 1e-2)
    out = orc.tanh(x)
    B = np.zeros_like(J)
    for j in range(x.size):
        go = np.zeros(x.size, np.float32); go[j] = 1
        B[:, j] = orc.tanh_backward(out, go.reshape(x.shape)).reshape(-1)
    assert np.abs(J - B).max() < 2e-4
    # log-softmax over rows
    J = _jacobian_fd(orc.log_softmax, x, 1e-2)
    out = orc.log_softmax(x)
    assert np.allclose(np.exp(out).sum(1), 1, atol=1e-6)
    B = np.zeros_like(J)
    for j in range(x.size):
        go = np.zeros(x.size, np.float32); go[j] = 1
        B[:, j] = orc.log_softmax_backward(out, go.reshape(x.shape)).reshape(-1)
    assert np.abs(J - B).max() < 2e-4
    # soft-max (the oracle's softmin of the negated input IS nn.SoftMax of the input)
    sm = lambda xx: orc.softmin(-xx)
    J = _jacobian_fd(sm, x, 1e-2)
    out = sm(x)
    B = np.zeros_like(J)
    for j in range(x.size):
        go = np.zeros(x.size, np.float32); go[j] = 1
        B[:, j] = orc.softmax_backward(out, go.reshape(x.shape)).reshape(-1)
    assert np.abs(J - B).max() < 2e-4
