"""csrc/proj.hip (the tall-skinny products of the TGCN step: Q = X W2, P = X[self] W1[:D] + b, their data / weight gradients,
tgcn.py:20-37) against fp64 torch.  Tolerance: exact-fp32 MFMA = an fp32 fma chain over k (<= 128 terms forward, up to
3e5 rows in the weight gradient, folded wave by wave), so 1e-5 of the result's scale forward and 2e-5 for the reductions."""
import numpy as np
import pytest
import torch

import tagrec_amd as T
from tagrec_amd import proj as PJ

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


def _rnd(gen, *shape):
    return (torch.randn(*shape, generator=gen) * 0.5).to(DEV)


def _close(got, want, tol):
    want = want.double()
    scale = float(want.abs().max()) + 1e-30
    err = float((got.double() - want).abs().max())
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize("K,NO", [(128, 32), (128, 64), (64, 32), (16, 16), (32, 128), (64, 128), (32, 16), (128, 128)])
@pytest.mark.parametrize("n", [1, 37, 4099])
def test_tall_mm_plain_and_bias(K, NO, n):
    gen = torch.Generator().manual_seed(K * 7 + NO + n)
    X, W, b = _rnd(gen, n, K), _rnd(gen, K, NO), _rnd(gen, 1, NO)
    out = torch.full((n, NO), float("nan"), device=DEV)
    PJ.tall_mm(X, W, out, b1=b)
    _close(out, X.double() @ W.double() + b.double(), 1e-5)
    # accumulate, transposed small matrix (stored [NO, K])
    Wt = W.t().contiguous()
    base = _rnd(gen, n, NO)
    out2 = base.clone()
    PJ.tall_mm(X, Wt, out2, transposed=True, accumulate=True)
    _close(out2, base.double() + X.double() @ W.double(), 1e-5)


def test_tall_mm_splits_gather_and_row_sliced_weights():
    """The forms the TGCN step uses: P for both neighbour types in one pass (weights = row slices [:D] of two [D + dw, A]
    matrices, two biases, two outputs, gathered rows) and dXs += [dP1 | dP2] [W1a[:D]^T ; W1b[:D]^T]."""
    gen = torch.Generator().manual_seed(5)
    n_tab, m, D, A, dw = 5000, 1777, 128, 32, 10
    X = _rnd(gen, n_tab, D)
    sel = torch.randperm(n_tab, generator=gen)[:m].to(DEV)
    Wa, Wb = _rnd(gen, D + dw, A), _rnd(gen, D + dw, A)
    ba, bb = _rnd(gen, 1, A), _rnd(gen, 1, A)
    P1 = torch.full((m, A), float("nan"), device=DEV)
    P2 = torch.full((m, A), float("nan"), device=DEV)
    PJ.tall_mm(X, Wa[:D], P1, sel=sel, W2=Wb[:D], w_split=2, b1=ba, b2=bb, out2=P2)
    Xs = X[sel].double()
    _close(P1, Xs @ Wa[:D].double() + ba.double(), 1e-5)
    _close(P2, Xs @ Wb[:D].double() + bb.double(), 1e-5)
    dP1, dP2 = _rnd(gen, m, A), _rnd(gen, m, A)
    base = _rnd(gen, m, D)
    dXs = base.clone()
    PJ.tall_mm(dP1, Wa[:D], dXs, X2=dP2, W2=Wb[:D], w_split=1, transposed=True, accumulate=True)
    _close(dXs, base.double() + dP1.double() @ Wa[:D].double().t() + dP2.double() @ Wb[:D].double().t(), 1e-5)


@pytest.mark.parametrize("KI,NO,n", [(128, 32, 300_001), (128, 64, 70_000), (64, 32, 5), (16, 16, 0), (32, 128, 1000), (128, 128, 2049)])
def test_tall_wgrad(KI, NO, n):
    gen = torch.Generator().manual_seed(KI + NO + n)
    X = _rnd(gen, n, KI)
    if NO >= 32:
        d1, d2 = _rnd(gen, n, NO // 2), _rnd(gen, n, NO // 2)
        dY = torch.cat([d1, d2], dim=1)
        db1, db2 = torch.ones(1, NO // 2, device=DEV), torch.ones(1, NO // 2, device=DEV)
        dW = PJ.tall_wgrad(X, d1, d2, db1=db1, db2=db2, acc_b=True)
        _close(db1, 1 + d1.double().sum(0, keepdim=True), 2e-5)
        _close(db2, 1 + d2.double().sum(0, keepdim=True), 2e-5)
    else:
        dY = _rnd(gen, n, NO)
        dW = PJ.tall_wgrad(X, dY)
    want = X.double().t() @ dY.double()
    if n == 0:
        assert float(dW.abs().max()) == 0.0
    else:
        _close(dW, want, 2e-5)
    # accumulate into an existing gradient, single dY with its bias gradient
    dW2 = torch.ones(KI, NO, device=DEV)
    db = torch.zeros(NO, device=DEV)
    PJ.tall_wgrad(X, dY.contiguous(), dW=dW2, db1=db, acc_w=True)
    _close(dW2, 1 + want, 2e-5)
    _close(db, dY.double().sum(0), 2e-5)


def test_small_mm_and_row_add_at():
    gen = torch.Generator().manual_seed(2)
    ewp, W, dWT = _rnd(gen, 18, 10), _rnd(gen, 138, 32), _rnd(gen, 18, 32)
    _close(PJ.small_mm(ewp, W[128:]), ewp.double() @ W[128:].double(), 1e-6)
    acc = torch.ones(10, 32, device=DEV)
    PJ.small_mm(ewp.t(), dWT, out=acc, accumulate=True)
    _close(acc, 1 + ewp.double().t() @ dWT.double(), 1e-6)
    _close(PJ.small_mm(dWT, W[128:].t()), dWT.double() @ W[128:].double().t(), 1e-6)
    dst = _rnd(gen, 1000, 64)
    pos = torch.randperm(1000, generator=gen)[:300].to(DEV)
    src = _rnd(gen, 300, 64)
    want = dst.clone()
    want[pos] += src
    PJ.row_add_at(dst, pos, src)
    assert torch.equal(dst, want)


def test_masked_colsum():
    gen = torch.Generator().manual_seed(4)
    for n, D in ((100_003, 128), (7, 64), (0, 32), (5000, 16)):
        d, o = _rnd(gen, n, D), _rnd(gen, n, D)
        want = (d.double() * (o > 0)).sum(0)
        got = PJ.masked_colsum(d, o)
        if n == 0:
            assert float(got.abs().max()) == 0.0
        else:
            _close(got, want, 2e-5)


def test_shape_checks_fail_loudly():
    X, W = torch.zeros(8, 24, device=DEV), torch.zeros(24, 32, device=DEV)
    with pytest.raises(T.TagrecError, match="16, 32, 64 or 128"):
        PJ.tall_mm(X, W, torch.zeros(8, 32, device=DEV))
    assert not PJ.supported(24, 32) and PJ.supported(128, 32, 64)
