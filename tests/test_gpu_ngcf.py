"""GPU parity of the NGCF path (SpMM on the non-symmetric D^-1 A + I, MFMA dense layer, hand-written
backward) against the reference's golden vectors and the CPU oracle.

Tolerances: activations rtol 1e-5 / atol 1e-6; losses rtol 1e-5; gradients rtol 2e-3 with an absolute floor
of 1e-6 x max|grad| (fp32 sums over N rows in a different order); parameters after Adam atol 2e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tagrec_amd as T
from tagrec_amd import ngcf as NG
from oracle import models as om
from test_gpu_lightgcn import DEV, _ds_from_fixture


def _model(fx, **kw):
    cfg = T.get_config("ngcf", use_tag=bool(int(fx["use_tag"])), dim_layer_list=[int(x) for x in fx["layers"]],
                       dim_latent=int(fx["D"]), reg=float(fx["reg"]), device=DEV, **kw)
    m = T.NGCF(_ds_from_fixture(fx), config=cfg)
    m.load_state_dict({k[5:]: torch.from_numpy(fx[k]) for k in fx if k.startswith("init.")})
    return m


def _grad_close(got, want, name, scale=None):
    """atol from `scale` = the largest gradient entry of the whole model when given: entries that are sums
    cancelling to ~1e-7 carry absolute errors of the size of their terms."""
    scale = float(np.abs(want).max()) if scale is None else scale
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=1e-6 * max(1e-3, scale), err_msg=name)


def test_state_dict_layout(golden):
    fx = golden("ngcf_toy")
    m = _model(fx)
    assert list(m.state_dict().keys()) == [k[5:] for k in fx if k.startswith("init.")]


@pytest.mark.parametrize("name", ["ngcf_toy", "ngcf_med"])
def test_ngcf_forward_loss_grads_golden(golden, name):
    fx = golden(name)
    m = _model(fx)
    m.eval()
    with torch.no_grad():
        outs = m.forward()
    for t, o in enumerate(outs):
        np.testing.assert_allclose(o.cpu().numpy(), fx[f"out.{t}"], rtol=1e-5, atol=1e-6)
    m.train()
    lossx = m.loss(torch.from_numpy(fx["batches"][0]).to(DEV))
    np.testing.assert_allclose([float(v) for v in lossx], fx["loss_parts"], rtol=1e-5, atol=1e-8)
    sum(lossx).backward()
    want = np.concatenate([fx[f"grad.embed.{t}"] for t in range(len(m.num_list))])
    scale = max([float(np.abs(want).max())] + [float(np.abs(fx[f"grad.mat.{k}"]).max()) for k in m.mat])
    _grad_close(m.table.grad.cpu().numpy(), want, "table", scale)
    for k, p in m.mat.items():
        _grad_close(p.grad.cpu().numpy(), fx[f"grad.mat.{k}"], k, scale)


@pytest.mark.parametrize("name", ["ngcf_toy", "ngcf_med"])
def test_ngcf_adam_steps_golden(golden, name):
    fx = golden(name)
    for n_steps in (1, 3):
        m = _model(fx)
        m.train()
        opt = T.Adam(m.parameters(), lr=float(fx["lr"]))
        prod = T.Fixed_training_data([fx["batches"][0]], fx["batches"].shape[1], DEV)
        prod.mini_batch = lambda: iter([torch.from_numpy(b).to(DEV) for b in fx["batches"][:n_steps]])
        losses = T.epoch_training(prod, m.loss, opt, verbose=False)
        np.testing.assert_allclose(losses, fx[f"step{n_steps}.losses"], rtol=5e-5)
        sd = m.state_dict()
        for k in sd:
            got, want = sd[k].cpu().numpy(), fx[f"step{n_steps}.{k}"]
            assert np.abs(got - want).max() <= 2e-4, k
            assert np.mean(np.abs(got - want) <= 2e-5) >= 0.99, k


def test_ngcf_unfused_path_matches_fused(golden):
    fx = golden("ngcf_toy")
    m = _model(fx)
    m2 = _model(fx, split_adj_k=2)
    assert isinstance(m2.norm_adj, list)
    b = torch.from_numpy(fx["batches"][0]).to(DEV)
    l1, l2 = m.loss(b), m2.loss(b)
    np.testing.assert_allclose([float(v) for v in l2], [float(v) for v in l1], rtol=1e-5)
    sum(l1).backward(); sum(l2).backward()
    scale = max([float(m.table.grad.abs().max())] + [float(m.mat[k].grad.abs().max()) for k in m.mat])
    _grad_close(m2.table.grad.cpu().numpy(), m.table.grad.cpu().numpy(), "table", scale)
    for k in m.mat:
        _grad_close(m2.mat[k].grad.cpu().numpy(), m.mat[k].grad.cpu().numpy(), k, scale)


@pytest.mark.parametrize("din,dout", [(64, 64), (64, 32), (32, 16), (16, 64), (128, 64), (16, 16), (64, 128), (128, 128)])
def test_dense_layer_kernels_vs_torch(din, dout):
    """The three MFMA kernels in isolation against torch autograd (fp64 reference), ragged row count."""
    torch.manual_seed(din * 7 + dout)
    n = 1000 + 37
    nei, x = torch.randn(n, din), torch.randn(n, din)
    w1p, w2p = torch.randn(din, dout) * 0.3, torch.randn(din, dout) * 0.3
    up = torch.randn(n, dout)
    ref = [t.double().requires_grad_() for t in (nei, x, w1p, w2p)]
    xp_ref = torch.nn.functional.leaky_relu((ref[0] + ref[1]) @ ref[2], 0.2) + \
        torch.nn.functional.leaky_relu((ref[0] * ref[1]) @ ref[3], 0.2)
    (xp_ref * up.double()).sum().backward()
    g = [t.to(DEV) for t in (nei, x, w1p, w2p)]
    xp, inv = torch.empty(n, dout, device=DEV), torch.empty(n, device=DEV)
    zbuf = torch.zeros(n, dout + 8, device=DEV)
    NG.dense_forward(g[0], g[1], g[2], g[3], xp, inv, zbuf[:, 4:], dout + 8)
    np.testing.assert_allclose(xp.cpu().numpy(), xp_ref.detach().float().numpy(), rtol=2e-5, atol=2e-5)
    zr = torch.nn.functional.normalize(xp_ref.detach().float(), p=2, dim=1)
    np.testing.assert_allclose(zbuf[:, 4:4 + dout].cpu().numpy(), zr.numpy(), rtol=2e-5, atol=2e-6)
    assert float(zbuf[:, :4].abs().max()) == 0 and float(zbuf[:, 4 + dout:].abs().max()) == 0
    d_nei, d_xd, dw1, dw2 = NG.dense_backward(up.to(DEV), *g)
    # d_nei / d_xd are the two halves of dX: dN = dA1 + dA2*X ; dXd = dA1 + dA2*N
    np.testing.assert_allclose(d_nei.cpu().numpy(), ref[0].grad.float().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(d_xd.cpu().numpy(), ref[1].grad.float().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dw1.cpu().numpy(), ref[2].grad.float().numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(dw2.cpu().numpy(), ref[3].grad.float().numpy(), rtol=1e-4, atol=2e-3)


def test_unsupported_width_is_reported_not_miscomputed():
    x = torch.randn(64, 48, device=DEV)
    w = torch.randn(48, 48, device=DEV)
    with pytest.raises(T.TagrecError, match="48 -> 48"):
        NG.dense_forward(x, x, w, w, torch.empty(64, 48, device=DEV), torch.empty(64, device=DEV),
                         torch.empty(64, 48, device=DEV), 48)


def test_ngcf_128_wide_layers_run_fused(golden):
    """128 -> 128 layers (one matrix per launch in the backward kernels): the model stays on the fused path and a training
    step gives the losses and gradients of the operator-by-operator path (SpMM kernel + torch matmul / autograd)."""
    ds = T.synth.make_cf_dataset(300, 250, 6000, seed=4)
    cfg = T.get_config("ngcf", use_tag=False, dim_layer_list=[128, 128], dim_latent=128, device=DEV, reg=1e-3)
    torch.manual_seed(0)
    m = T.NGCF(ds, config=cfg)
    m.train()
    assert m._fused_ok()
    b = torch.from_numpy(T.synth.sample_bpr_epoch(ds, 0)[:64]).to(DEV)
    l1 = m.loss(b)
    sum(l1).backward()
    got = {k: v.grad.clone() for k, v in m.named_parameters()}
    m.zero_grad()
    m._fused_ok = lambda: False                      # operator path
    l2 = m.loss(b)
    sum(l2).backward()
    np.testing.assert_allclose([float(v) for v in l1], [float(v) for v in l2], rtol=1e-5)
    scale = max(float(v.grad.abs().max()) for v in m.parameters())
    for k, v in m.named_parameters():
        _grad_close(got[k].cpu().numpy(), v.grad.cpu().numpy(), k, scale)


@pytest.mark.parametrize("layers", [[64, 64, 32], [64, 32], [32]])
def test_restricted_forward_equals_full_forward_step(layers):
    """NGCF.loss with the top two layers' neighbour sums restricted to the rows the loss depends on vs all rows: same
    loss parts, same gradients (table and W / b), on a graph with long rows."""
    from tagrec_amd import ngcf as NG
    ds = T.synth.make_bipartite_device(30_000, 20_000, 1_500_000, seed=5, device=DEV)
    cfg = T.get_config("ngcf", use_tag=False, dim_latent=64, dim_layer_list=layers, device=DEV, train_batch=128, reg=1e-3)
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], 30_000, 20_000, "ngcf")
    g = T.Graph(rp, col, val, (n, n))
    torch.manual_seed(3)
    m = T.NGCF(ds, config=cfg, graph=g)
    m.train()
    batch = T.BPR_training_data(ds, config=cfg, seed=2).all_train_data[:128]
    res = []
    try:
        for restrict in (False, True):
            NG.RESTRICT_FORWARD = restrict
            m.zero_grad()
            lossx = m.loss(batch)
            sum(lossx).backward()
            res.append(([float(v) for v in lossx], {k: p.grad.clone() for k, p in m.named_parameters()}))
    finally:
        NG.RESTRICT_FORWARD = True
    (l0, g0), (l1, g1) = res
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    top = max(float(v.double().norm()) for v in g0.values())
    for k in g0:
        a, b = g0[k].double(), g1[k].double()
        assert float((a - b).norm()) <= 1e-3 * float(a.norm()) + 1e-6 * top, k


@pytest.mark.parametrize("din,dout", [(64, 64), (32, 64), (128, 128)])
def test_dense_layer_kernels_row_mask_and_dz_flags(din, dout):
    """The *_rows_* forms (restricted training step): rows outside row_mask are neither read nor written (their output
    slots keep a sentinel, their inputs may hold NaN), the weight gradient counts them as zero, and rows with a zero
    dz_flags byte skip the normalize-backward term; the rows that are computed equal the unmasked kernels' rows."""
    torch.manual_seed(din + dout)
    n = 777
    nei, x = torch.randn(n, din, device=DEV), torch.randn(n, din, device=DEV)
    w1p, w2p = torch.randn(din, dout, device=DEV) * 0.3, torch.randn(din, dout, device=DEV) * 0.3
    mask = (torch.rand(n, device=DEV) < 0.3).to(torch.uint8)
    mask[16:48] = 0                                            # whole 16-row groups without a wanted row
    mask[n - 5:] = 1
    keep = mask.bool()
    dzf = ((torch.rand(n, device=DEV) < 0.2) & keep).to(torch.uint8)
    # full run
    xp, inv = torch.empty(n, dout, device=DEV), torch.empty(n, device=DEV)
    z = torch.empty(n, dout, device=DEV)
    NG.dense_forward(nei, x, w1p, w2p, xp, inv, z, dout)
    up, dz = torch.randn(n, dout, device=DEV), torch.randn(n, dout, device=DEV) * dzf[:, None]
    up_m = up * keep[:, None]
    ref = NG.dense_backward(up_m, nei, x, w1p, w2p, norm=(xp, inv, dz, dout))
    # masked run on poisoned inputs
    nan = float("nan")
    nei_p, x_p = nei.clone(), x.clone()
    nei_p[~keep] = nan
    x_p[~keep] = nan
    xp2, inv2 = torch.full((n, dout), 7.0, device=DEV), torch.full((n,), 7.0, device=DEV)
    NG.dense_forward(nei_p, x_p, w1p, w2p, xp2, inv2, None, 0, mask)
    assert torch.equal(xp2[keep], xp[keep]) and torch.equal(inv2[keep], inv[keep])
    assert bool((xp2[~keep] == 7.0).all()) and bool((inv2[~keep] == 7.0).all())
    up_p, dz_p, xp_p = up.clone(), dz.clone(), xp.clone()
    up_p[~keep] = nan
    dz_p[~dzf.bool()] = nan                                   # must not be read where dz_flags is 0
    xp_p[~dzf.bool()] = nan
    got = NG.dense_backward(up_p, nei_p, x_p, w1p, w2p, norm=(xp_p, inv, dz_p, dout), row_mask=mask, dz_flags=dzf)
    for a, b in zip(got[:2], ref[:2]):
        assert torch.equal(a[keep], b[keep])
    for a, b in zip(got[2:], ref[2:]):                        # weight gradients: same rows contribute, same fold order
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-4)
        assert bool(torch.isfinite(a).all())
