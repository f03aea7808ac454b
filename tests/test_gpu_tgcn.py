"""GPU parity of the TGCN path: HIP neighbour-attention kernels (forward + backward) and the whole model
against the reference's golden vectors (tests/golden/tgcn_toy.npz) and the CPU oracle.

Tolerances: activations rtol 2e-5 / atol 2e-6; losses rtol 1e-5; gradients rtol 5e-3 with an absolute floor of
2e-6 x max|grad| per tensor (float-atomic scatter order + different association of the W1 split)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tagrec_amd as T
from tagrec_amd import tgcn as TG
from oracle import models as om
from test_gpu_lightgcn import DEV, _ds_from_fixture


def _nbr(fx):
    return [(fx[f"nbr{r}.ids"], fx[f"nbr{r}.wts"]) for r in range(6)]


def _model(fx, seed_init=False, **kw):
    cfg = T.get_config("tgcn", use_tag=True, dim_layer_list=[int(x) for x in fx["layers"]], dim_latent=int(fx["D"]),
                       reg=float(fx["reg"]), neighbor_k=int(fx["neighbor_k"]), device=DEV, **kw)
    ds = _ds_from_fixture(fx)
    ds.num["weight"] = int(fx["n_weight"])
    if seed_init:
        torch.manual_seed(2020)
    m = T.TGCN(ds, config=cfg, neighbors=_nbr(fx))
    if not seed_init:
        m.load_state_dict({k[5:]: torch.from_numpy(fx[k]) for k in fx if k.startswith("init.")})
    return m


def _close(got, want, name, rtol=5e-3, scale=None):
    """atol is set from `scale` (the largest gradient entry of the whole model when given): tensors whose
    gradient is a sum that cancels to ~1e-7 (e.g. the last layer's bias, whose gradient the L2-normalise
    backward projects out) carry absolute errors of the size of the terms, not of the sum."""
    scale = float(np.abs(want).max()) if scale is None else scale
    np.testing.assert_allclose(got, want, rtol=rtol, atol=2e-6 * max(1e-3, scale), err_msg=name)


def test_seeded_init_reproduces_reference(golden):
    """torch.manual_seed(2020) + construction draws the reference's initial values (incl. the draws
    nn.Conv2d consumes at construction, and the ParameterDict key order)."""
    fx = golden("tgcn_toy")
    m = _model(fx, seed_init=True)
    sd = m.state_dict()
    assert list(sd.keys()) == [k[5:] for k in fx if k.startswith("init.")]
    for k, v in sd.items():
        np.testing.assert_array_equal(v.cpu().numpy(), fx["init." + k], err_msg=k)


def test_tgcn_forward_loss_grads_golden(golden):
    fx = golden("tgcn_toy")
    m = _model(fx)
    m.eval()
    with torch.no_grad():
        outs = m.forward()
    for t, o in enumerate(outs):
        np.testing.assert_allclose(o.cpu().numpy(), fx[f"out.{t}"], rtol=2e-5, atol=2e-6)
    m.train()
    lossx = m.loss(torch.from_numpy(fx["batches"][0]).to(DEV))
    np.testing.assert_allclose([float(v) for v in lossx], fx["loss_parts"], rtol=1e-5)
    sum(lossx).backward()
    scale = max(float(np.abs(fx["grad." + k]).max()) for k, _ in m.named_parameters())
    for k, p in m.named_parameters():
        g = p.grad.cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        _close(g, fx["grad." + k], k, scale=scale)


def test_tgcn_chunked_checkpointed_dense_equals_unchunked(golden):
    fx = golden("tgcn_toy")
    b = torch.from_numpy(fx["batches"][0]).to(DEV)
    grads = []
    for kw in ({"tgcn_chunk_rows": 7, "tgcn_checkpoint": True}, {"tgcn_chunk_rows": 10 ** 6, "tgcn_checkpoint": False}):
        m = _model(fx, **kw)
        m.train()
        sum(m.loss(b)).backward()
        grads.append({k: p.grad.cpu().numpy() for k, p in m.named_parameters() if p.grad is not None})
    scale = max(float(np.abs(v).max()) for v in grads[1].values())
    for k in grads[0]:
        _close(grads[0][k], grads[1][k], k, rtol=1e-4, scale=scale)


def test_transtag_phase_golden(golden):
    fx = golden("tgcn_toy")
    m = _model(fx)
    lossx = m.transtag_loss(torch.from_numpy(fx["tt_batch"]).to(DEV))
    np.testing.assert_allclose([float(v) for v in lossx], fx["tt_loss_parts"], rtol=1e-5)
    sum(lossx).backward()
    for k in ("user", "item", "tag"):
        _close(m.embed[k].grad.cpu().numpy(), fx[f"tt_grad.embed.{k}"], k, rtol=1e-4)


@pytest.mark.parametrize("pull", [False, True])
@pytest.mark.parametrize("D,k,A", [(128, 25, 32), (64, 25, 32), (16, 5, 32), (32, 64, 16), (256, 3, 64)])
def test_attention_kernels_vs_oracle(D, k, A, pull):
    """`Attention1` through the HIP kernels vs the oracle's direct restatement, C4-shaped rows (D=128, k=25)."""
    torch.manual_seed(D + k)
    n, m, nw, dw = 300, 200, 7, 10
    ev, ej, ew = torch.randn(n, D) * 0.3, torch.randn(m, D) * 0.3, torch.randn(nw, dw) * 0.3
    prm = {"W_1": torch.randn(D + dw, A) * 0.2, "W_2": torch.randn(D, A) * 0.2, "b": torch.randn(1, A) * 0.1,
           "v": torch.randn(1, A)}
    idx = torch.randint(0, m + 1, (n, k))
    idx[3] = 0                                                  # a node with no neighbours: all pads
    widx = torch.where(idx > 0, torch.randint(1, nw + 1, (n, k)), torch.zeros(n, k, dtype=torch.long))
    up = torch.randn(n, D)
    ref_in = {kk: v.clone().requires_grad_() for kk, v in prm.items()}
    rv, rj, rw = (t.clone().requires_grad_() for t in (ev, ej, ew))
    want = om.tgcn_attention1(ref_in, "", rv, rj, rw, idx, widx)
    (want * up).sum().backward()
    g = {kk: v.clone().to(DEV).requires_grad_() for kk, v in prm.items()}
    gv, gj, gw = (t.clone().to(DEV).requires_grad_() for t in (ev, ej, ew))
    ewp = torch.cat([gw.new_zeros(1, dw), gw])
    P = gv @ g["W_1"][:D] + g["b"]
    idx_d = idx.to(DEV, torch.int32).contiguous()
    inv = TG.InverseTable(idx_d, m) if pull else None        # pull form: dQ / dEj through the inverted table + SpMM
    got = TG.neighbour_attention(P, gj @ g["W_2"], ewp @ g["W_1"][D:], g["v"].reshape(-1), gj,
                                 idx_d, widx.to(DEV, torch.int32).contiguous(), inv)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-5, atol=2e-6)
    (got * up.to(DEV)).sum().backward()
    for name, a, b in (("ev", gv, rv), ("ej", gj, rj), ("ew", gw, rw)):
        _close(a.grad.cpu().numpy(), b.grad.numpy(), name, rtol=2e-3)
    for kk in prm:
        _close(g[kk].grad.cpu().numpy(), ref_in[kk].grad.numpy(), kk, rtol=2e-3)


def test_neighbor_tables_semantics():
    ds = T.synth.make_cf_dataset(50, 40, 400, seed=4, n_tag=15, n_assign=300)
    tabs = TG.neighbor_tables(ds, 6, seed=1)
    assert len(tabs) == 6
    ui = {}
    for u, i in zip(ds.ui_adj.row, ds.ui_adj.col):
        ui.setdefault(int(u), set()).add(int(i))
    ids, wts = tabs[0]
    assert ids.shape == (50, 6) and ids.dtype == np.int32
    for u in range(50):
        got = set(ids[u].tolist())
        if u in ui:
            assert 0 not in got and {g - 1 for g in got} <= ui[u] and (wts[u] == 1).all()
        else:
            assert got == {0}
    ids_ut, wts_ut = tabs[1]                       # user-tag: integer co-occurrence weights
    assert wts_ut.max() >= 1 and (wts_ut[ids_ut == 0] == 0).all()


@pytest.mark.parametrize("D,Dout,n", [(16, 16, 100), (64, 32, 333), (128, 128, 257), (32, 64, 64), (64, 64, 31), (128, 64, 700),
                                      (64, 128, 40000)])
def test_fused_dense_block_vs_operator_form(D, Dout, n):
    """csrc/tgcn_fuse.hip (type attention + convolutions + fusion in one kernel) against the same block
    written operator by operator (fp64 on the host); ragged node counts; gradients through autograd."""
    torch.manual_seed(D * 3 + Dout)
    A, C, V = 32, 32, 8
    ts = [torch.randn(n, D) * 0.5 for _ in range(3)]
    prm = [torch.randn(D, A) * 0.2, torch.randn(1, A) * 0.1, torch.randn(1, A), torch.randn(C, 1, 3, 1) * 0.5,
           torch.randn(V, 1, 1, D) * 0.2, torch.randn(V, 1, 2, D) * 0.2, torch.randn(V, 1, 3, D) * 0.2,
           torch.randn(C * D + 6 * V, Dout) * 0.05, torch.randn(1, Dout) * 0.1]
    up = torch.randn(n, Dout)
    # ReLU is not differentiable at 0: a pre-activation within rounding distance of 0 (|x| ~ 1e-8 happens among
    # the ~1e6 bit-level pre-activations) gets gradient 0 or 1 depending on the last bit, a finite difference no
    # tolerance covers.  Such nodes get a zero upstream gradient so they do not take part in the comparison.
    with torch.no_grad():
        st64 = torch.stack([t.double() for t in ts], 1)
        P64 = [x.double() for x in prm]
        S64 = st64 @ P64[0] + P64[1]
        e64 = torch.softmax(torch.relu(S64) @ P64[2].t(), dim=1) * st64
        bit64 = torch.einsum("cj,njd->ncd", P64[3][:, 0, :, 0], e64)
        amb = (bit64.abs().amin(dim=(1, 2)) < 2e-7) | (S64.abs().amin(dim=(1, 2)) < 2e-7)
        up[amb] = 0.0
    rt = [t.double().requires_grad_() for t in ts]
    rp = [x.double().requires_grad_() for x in prm]
    want = TG._dense_block(torch.stack(rt, dim=1), *rp)
    (want * up.double()).sum().backward()
    gt = [t.to(DEV).requires_grad_() for t in ts]
    gp = [x.to(DEV).requires_grad_() for x in prm]
    U, q, p, wb, w1, w2, w3, Wf, bf = gp
    got = TG._FusedDense.apply(gt[0], gt[1], gt[2], U, q.reshape(-1), p.reshape(-1), wb.reshape(C, 3), w1.reshape(V, -1),
                               w2.reshape(V, -1), w3.reshape(V, -1), Wf, bf.reshape(-1), 50)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().float().numpy(), rtol=1e-4, atol=2e-5)
    assert int(amb.sum()) <= n // 10
    (got * up.to(DEV)).sum().backward()
    scale = max(float(x.grad.abs().max()) for x in rt + rp)
    for name, a, b in [(f"t{k}", gt[k], rt[k]) for k in range(3)] + [(f"p{k}", gp[k], rp[k]) for k in range(len(prm))]:
        _close(a.grad.cpu().numpy(), b.grad.float().numpy(), name, rtol=2e-3, scale=scale)


def test_tgcn_fused_equals_operator_path(golden):
    fx = golden("tgcn_toy")
    b = torch.from_numpy(fx["batches"][0]).to(DEV)
    res = []
    for fused in (True, False):
        m = _model(fx, tgcn_fused_dense=fused)
        m.train()
        lossx = m.loss(b)
        sum(lossx).backward()
        res.append(([float(v) for v in lossx], {k: p.grad.cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-5)
    scale = max(float(np.abs(v).max()) for v in res[1][1].values())
    for k in res[1][1]:
        _close(res[0][1][k], res[1][1][k], k, rtol=2e-3, scale=scale)


def test_c4_size_kernels_on_sampled_nodes():
    """At the C4 row count (1M nodes of one type, D=128, k=25) the oracle is too slow and the reference cannot
    materialise the block at all; check the kernels on sampled nodes against the operator form evaluated on
    just those nodes, plus run-to-run determinism of forward and of the pull-form backward."""
    gen = torch.Generator(device=DEV).manual_seed(4)
    n, m, D, A, k, nw, C, V = 1_000_000, 1_000_000, 128, 32, 25, 16, 32, 8
    r = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=gen) * sc
    # --- neighbour attention
    P, Q, WT, v, Ej = r(n, A, sc=0.3), r(m, A, sc=0.3), r(nw + 1, A, sc=0.3), r(A), r(m, D, sc=0.5)
    WT[0] = 0
    idx = torch.randint(0, m + 1, (n, k), device=DEV, generator=gen, dtype=torch.int32)
    widx = torch.where(idx > 0, torch.randint(1, nw + 1, (n, k), device=DEV, generator=gen, dtype=torch.int32),
                       torch.zeros_like(idx))
    out = TG.neighbour_attention(P, Q, WT, v, Ej, idx, widx)
    pick = torch.randint(0, n, (500,), device=DEV, generator=gen)
    ji = idx[pick].long()
    Qp = torch.cat([Q.new_zeros(1, A), Q])[ji]                       # pad row 0 = zeros
    Ep = torch.cat([Ej.new_zeros(1, D), Ej])[ji]
    s = (torch.relu(P[pick, None, :] + WT[widx[pick].long()] + Qp) * v).sum(-1)
    want = (torch.softmax(s, dim=1)[..., None] * Ep).sum(1)
    np.testing.assert_allclose(out[pick].cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-6)
    assert torch.equal(out, TG.neighbour_attention(P, Q, WT, v, Ej, idx, widx))
    # pull-form backward: deterministic, and dEj equals the scatter form up to atomic ordering
    inv = TG.InverseTable(idx, m)
    up = r(n, D)
    grads = []
    for tab in (inv, inv, None):
        leaves = [t.clone().requires_grad_() for t in (P, Q, Ej)]
        o = TG.neighbour_attention(leaves[0], leaves[1], WT, v, leaves[2], idx, widx, tab)
        (o * up).sum().backward()
        grads.append([t.grad for t in leaves])
    assert all(torch.equal(a, b) for a, b in zip(grads[0], grads[1]))
    for a, b in zip(grads[0], grads[2]):
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
    del grads, inv, out, Qp, Ep
    # --- fused dense block, forward, on sampled nodes
    ts = [r(n, D, sc=0.5) for _ in range(3)]
    prm = [r(D, A, sc=0.2), r(1, A, sc=0.1), r(1, A), r(C, 1, 3, 1, sc=0.5), r(V, 1, 1, D, sc=0.2), r(V, 1, 2, D, sc=0.2),
           r(V, 1, 3, D, sc=0.2), r(C * D + 6 * V, D, sc=0.05), r(1, D, sc=0.1)]
    U, q, p, wb, w1, w2, w3, Wf, bf = prm
    with torch.no_grad():
        got = TG._FusedDense.apply(ts[0], ts[1], ts[2], U, q.reshape(-1), p.reshape(-1), wb.reshape(C, 3), w1.reshape(V, -1),
                                   w2.reshape(V, -1), w3.reshape(V, -1), Wf, bf.reshape(-1), 65536)
        pick = torch.randint(0, n, (2000,), device=DEV, generator=gen)
        want = TG._dense_block(torch.stack([t[pick].double() for t in ts], 1), *[x.double() for x in prm])
    np.testing.assert_allclose(got[pick].cpu().numpy(), want.float().cpu().numpy(), rtol=1e-4, atol=3e-5)


def test_c4_size_fused_backward_kernels_vs_oracle():
    """`tgcn_fuse_bwd` / `tgcn_fuse_wf` at the C4 launch size (1 M nodes of one type, D = Dout = 128) against the ORACLE's
    `tgcn_atten2` + `tgcn_conv` + fusion layer (oracle/models.py, fp64) on sampled nodes.  The block is row-local, so
    (a) the input gradients of a sampled node depend on that node alone, and (b) a weight gradient is a sum of per-node
    terms, linear in the upstream gradient: W(up) - W(up with the sampled rows zeroed) is the sampled nodes' share, which
    the oracle can evaluate.  Both launches are full size (the upstream gradient is dense, so no rows are dropped)."""
    gen = torch.Generator(device=DEV).manual_seed(9)
    n, D, A, C, V = 1_000_000, 128, 32, 32, 8
    r = lambda *s, sc=1.0: torch.randn(*s, device=DEV, generator=gen) * sc
    ts = [r(n, D, sc=0.5) for _ in range(3)]
    prm = [r(D, A, sc=0.2), r(1, A, sc=0.1), r(1, A), r(C, 1, 3, 1, sc=0.5), r(V, 1, 1, D, sc=0.2), r(V, 1, 2, D, sc=0.2),
           r(V, 1, 3, D, sc=0.2), r(C * D + 6 * V, D, sc=0.05), r(1, D, sc=0.1)]
    names = ["U", "q", "p", "conv.bit_level.weight", "conv.vec_level.conv_1.weight", "conv.vec_level.conv_2.weight",
             "conv.vec_level.conv_3.weight", "Wf", "bf"]
    up = r(n, D)
    pick = torch.unique(torch.randint(0, n, (1500,), device=DEV, generator=gen))

    def oracle(idx):
        P = {"l." + k: v.detach().double().cpu().requires_grad_() for k, v in zip(names, prm)}
        t = [x[idx].double().cpu().requires_grad_() for x in ts]
        e3 = om.tgcn_atten2(P, "l.", *t)
        pre_bit = torch.einsum("cj,njd->ncd", P["l.conv.bit_level.weight"][:, 0, :, 0], e3)
        S = torch.stack(t, 1) @ P["l.U"] + P["l.q"]
        # ReLU kinks (see test_fused_dense_block_vs_operator_form): the fp32 kernel's e3 differs from the fp64 one by ~1e-6
        # relative, so a bit-level pre-activation of 2.5e-7 (met in this sample) lands on the other side of 0
        amb = (pre_bit.abs().amin(dim=(1, 2)) < 2e-6) | (S.abs().amin(dim=(1, 2)) < 1e-5)
        for j in (1, 2, 3):                                              # vector-level pre-activations: D- to 3D-term sums
            w = P[f"l.conv.vec_level.conv_{j}.weight"][:, 0]
            win = torch.stack([e3[:, h:h + j, :] for h in range(4 - j)], dim=1)
            amb |= torch.einsum("cad,nhad->nch", w, win).abs().amin(dim=(1, 2)) < 1e-5
        pre_out = om.tgcn_conv(P, "l.", e3) @ P["l.Wf"] + P["l.bf"]       # 4144-term fp32 sums: rounding ~1e-5
        amb |= pre_out.abs().amin(dim=1) < 5e-5
        return P, t, torch.relu(pre_out), amb

    with torch.no_grad():
        _, _, _, amb = oracle(pick)
    pick = pick[~amb.to(DEV)]
    assert pick.numel() >= 700

    def run(up_):
        leaves = [x.clone().requires_grad_() for x in ts]
        ps = [x.clone().requires_grad_() for x in prm]
        U, q, p, wb, w1, w2, w3, Wf, bf = ps
        out = TG._FusedDense.apply(leaves[0], leaves[1], leaves[2], U, q.reshape(-1), p.reshape(-1), wb.reshape(C, 3),
                                   w1.reshape(V, -1), w2.reshape(V, -1), w3.reshape(V, -1), Wf, bf.reshape(-1), 65536)
        TG.timing = {}
        (out * up_).sum().backward()
        launched = {k: len(v) for k, v in TG.timing.items()}
        TG.timing = None
        assert launched.get("fuse_bwd") == 1 and launched.get("fuse_wf") == 1        # one full-size launch each
        return out.detach(), [x.grad for x in leaves], [x.grad for x in ps]

    out, dts, dps = run(up)
    up0 = up.clone()
    up0[pick] = 0.0
    _, _, dps0 = run(up0)
    P, t, want_out, _ = oracle(pick)
    (want_out * up[pick].double().cpu()).sum().backward()
    np.testing.assert_allclose(out[pick].cpu().numpy(), want_out.detach().float().numpy(), rtol=1e-4, atol=3e-5)
    scale = max(float(x.grad.abs().max()) for x in t)
    for k in range(3):                                               # (a) input gradients, row-local
        _close(dts[k][pick].cpu().numpy(), t[k].grad.float().numpy(), f"d t{k}", rtol=2e-3, scale=scale)
    for name, a, b in zip(names, dps, dps0):                         # (b) weight gradients: the sampled nodes' share
        want = P["l." + name].grad.float().numpy().reshape(a.shape)
        got = (a.double() - b.double()).float().cpu().numpy()
        # the two full sums run over 1e6 nodes in fp32: their difference carries ~1e-4 of the FULL sum's magnitude
        floor = 2e-4 * float(a.abs().max()) + 1e-6
        assert np.all(np.abs(got - want) <= 2e-2 * np.abs(want) + floor + 2e-3 * np.abs(want).max()), name


def test_fused_dense_backward_drops_zero_gradient_rows():
    """`_FusedDense.backward` runs its kernels on the rows with a non-zero upstream gradient only (the batch rows / their
    sampled neighbours) and scatters the result back: same gradients as the all-rows pass."""
    from tagrec_amd import tgcn as TG
    n, D = 3000, 64
    gen = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: (torch.randn(*s, generator=gen) * 0.2).to(DEV)
    A, C, V = 32, 32, 8
    base = [rnd(n, D) for _ in range(3)] + [rnd(D, A), rnd(A), rnd(A), rnd(C, 3), rnd(V, D), rnd(V, 2 * D), rnd(V, 3 * D),
                                           rnd(C * D + 6 * V, D), rnd(D)]
    d_out = torch.zeros(n, D, device=DEV)
    rows = torch.randperm(n, generator=gen)[:97].to(DEV)
    d_out[rows] = rnd(97, D)
    grads = []
    old = TG._SPARSE_MIN_ROWS
    try:
        for thr in (10 ** 9, 0):                       # all rows, then compacted
            TG._SPARSE_MIN_ROWS = thr
            xs = [b.clone().requires_grad_() for b in base]
            out = TG._FusedDense.apply(*xs, 0)
            out.backward(d_out)
            grads.append([x.grad.clone() for x in xs])
    finally:
        TG._SPARSE_MIN_ROWS = old
    for a, b in zip(*grads):
        scale = float(a.abs().max()) + 1e-30
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol=1e-6 * scale)
    inactive = torch.ones(n, dtype=torch.bool, device=DEV)
    inactive[rows] = False
    assert float(grads[1][0][inactive].abs().max()) == 0.0


def test_attention_backward_drops_zero_gradient_rows():
    """`_NbrAttention.backward` on a gradient that is non-zero for a few nodes only: the compacted passes (scatter
    form for a handful of rows, pull form over a table inverted on the spot for more) give the gradients of the
    all-rows pass (pull form through the prebuilt inverted table)."""
    from tagrec_amd import tgcn as TG
    n, n_nbr, k, D, A, n_wt = 4000, 3000, 7, 64, 32, 9
    gen = torch.Generator(device="cpu").manual_seed(11)
    rnd = lambda *s: (torch.randn(*s, generator=gen) * 0.3).to(DEV)
    idx = torch.randint(0, n_nbr + 1, (n, k), generator=gen).to(DEV)          # 0 = pad row
    widx = torch.randint(0, n_wt, (n, k), generator=gen).to(DEV)
    idx32, widx32 = idx.to(torch.int32), widx.to(torch.int32)
    base = [rnd(n, A), rnd(n_nbr, A), rnd(n_wt, A), rnd(A), rnd(n_nbr, D)]              # ids are stored + 1; 0 = pad
    d_out = torch.zeros(n, D, device=DEV)
    rows = torch.randperm(n, generator=gen)[:61].to(DEV)
    d_out[rows] = rnd(61, D)
    inv = TG.InverseTable(idx32, n_nbr)
    grads = []
    old, old_pull = TG._SPARSE_MIN_ROWS, TG._PULL_MIN_ROWS
    try:
        # all rows (pull form, prebuilt table); compact scatter form; compact pull form (table inverted on the spot)
        for thr, pull, use_inv in ((10 ** 9, 10 ** 9, inv), (0, 10 ** 9, inv), (0, 10 ** 9, None), (0, 0, None),
                                   (10 ** 9, 0, None)):             # last: all rows of a subset call, no prebuilt table
            TG._SPARSE_MIN_ROWS, TG._PULL_MIN_ROWS = thr, pull
            xs = [b.clone().requires_grad_() for b in base]
            out = TG.neighbour_attention(xs[0], xs[1], xs[2], xs[3], xs[4], idx32, widx32, use_inv)
            out.backward(d_out)
            grads.append([x.grad.clone() for x in xs])
    finally:
        TG._SPARSE_MIN_ROWS, TG._PULL_MIN_ROWS = old, old_pull
    for other in grads[1:]:
        for a, b in zip(grads[0], other):
            scale = float(a.abs().max()) + 1e-30
            np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol=2e-6 * scale)


def test_pruned_forward_gives_the_full_forward_loss_and_gradients():
    """`TGCN.loss` with the layers restricted to the rows the batch's loss depends on (`_forward_rows`) vs the full
    forward: same loss parts, same gradient of every parameter."""
    ds = T.synth.make_cf_dataset(900, 700, 9000, seed=21, n_tag=300, n_assign=7000)
    cfg = T.get_config("tgcn", dim_layer_list=[32, 32, 32], dim_latent=32, device=DEV, neighbor_k=4, train_batch=24, reg=1e-3)
    torch.manual_seed(4)
    m = T.TGCN(ds, config=cfg)
    m.train()
    batch = torch.from_numpy(T.synth.sample_bpr_epoch(ds, 3)[:24]).to(DEV)
    need = m._needed_rows(batch)
    assert need[3]["tag"].numel() == 0 and need[3]["user"].numel() <= 24
    assert all(need[2][t] is None or need[2][t].numel() < need[1][t].numel() if need[1][t] is not None else True
               for t in ("user", "item", "tag"))
    out = []
    for prune in (False, True):
        m.prune_forward = prune
        m.zero_grad()
        lossx = m.loss(batch)
        sum(lossx).backward()
        out.append(([float(v) for v in lossx], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = out
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    assert set(g0) == set(g1)
    # sums over nodes run over different row sets / orders in the two passes (and the compact attention backward uses
    # float atomics): compare tensors as a whole
    worst, norms = {}, {}
    for k in g0:
        a, b = g0[k].double(), g1[k].double()
        norms[k] = float(a.norm())
        worst[k] = float((a - b).norm())
    top = max(norms.values())
    # per tensor: relative to its own norm, plus fp32 accumulation noise at the scale of the largest gradient (the top
    # layer's parameter gradients are ~1e-8 here, sums of +-1e-3 terms that cancel: both passes carry ~1e-8 of noise)
    bad = {k: (worst[k], norms[k]) for k in g0 if worst[k] > 2e-3 * norms[k] + 2e-6 * top}
    assert not bad, (top, sorted(bad.items(), key=lambda kv: -kv[1][0])[:6])


def test_pruned_and_row_sparse_step_equals_full_step_at_mid_scale():
    """A 200 k-node tripartite graph (D = 64, k = 25, L = 3), where both shortcuts switch on by themselves: the loss and
    the embedding gradients of the restricted step (needed rows only in the forward, non-zero-gradient rows only in the
    backward) against the all-rows step."""
    from tagrec_amd import tgcn as TG
    ds = T.synth.make_tripartite_device(50_000, 50_000, 100_000, 3_000_000, seed=6, device=DEV)
    cfg = T.get_config("tgcn", dim_latent=64, dim_layer_list=[64, 64, 64], device=DEV, train_batch=256, neighbor_k=25, reg=1e-4)
    torch.manual_seed(1)
    m = T.TGCN(ds, config=cfg)
    m.train()
    assert m.prune_forward
    prod = T.BPR_training_data(ds, config=cfg, seed=5)
    batch = prod.all_train_data[:256]
    need = m._needed_rows(batch)
    assert need[3]["user"].numel() <= 256 and need[2]["item"] is not None and need[2]["item"].numel() < 25_000
    res = []
    old = TG._SPARSE_MIN_ROWS
    try:
        for restricted in (False, True):
            m.prune_forward = restricted
            TG._SPARSE_MIN_ROWS = old if restricted else 10 ** 12
            m.zero_grad()
            lossx = m.loss(batch)
            sum(lossx).backward()
            res.append(([float(v) for v in lossx], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    finally:
        TG._SPARSE_MIN_ROWS = old
        m.prune_forward = True
    (l0, g0), (l1, g1) = res
    np.testing.assert_allclose(l1, l0, rtol=2e-6)
    top = max(float(g.double().norm()) for g in g0.values())
    for k in g0:
        a, b = g0[k].double(), g1[k].double()
        assert float((a - b).norm()) <= 2e-3 * float(a.norm()) + 2e-6 * top, k


def test_step_node_with_message_dropout_equals_all_rows_pass():
    """Message dropout (tgcn.py:217-219) in the hand-derived step node: the mask of a layer output is the library's
    counter-based one keyed by (seed, layer, node type, NODE id, column), so the restricted step on compact tables and the
    all-rows pass (`forward()` + autograd) handed the same seed drop the same elements: same loss, same gradients."""
    ds = T.synth.make_tripartite_device(50_000, 50_000, 100_000, 3_000_000, seed=6, device=DEV)
    cfg = T.get_config("tgcn", dim_latent=64, dim_layer_list=[64, 64, 64], device=DEV, train_batch=256, neighbor_k=25, reg=1e-4,
                       message_drop_list=[0.2, 0.0, 0.3])
    torch.manual_seed(1)
    m = T.TGCN(ds, config=cfg)
    m.train()
    batch = T.BPR_training_data(ds, config=cfg, seed=5).all_train_data[:256]
    res = []
    for restricted in (False, True):
        m.prune_forward = restricted
        m._drop_calls = 11
        m.zero_grad()
        lossx = m.loss(batch)
        sum(lossx).backward()
        res.append(([float(v) for v in lossx], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    m._drop_calls = 12
    other = [float(v) for v in m.loss(batch)]
    (l0, g0), (l1, g1) = res
    assert other != l1                                     # another seed, another mask
    np.testing.assert_allclose(l1, l0, rtol=2e-6)
    assert set(g0) == set(g1)
    top = max(float(g.double().norm()) for g in g0.values())
    for k in g0:
        a, b = g0[k].double(), g1[k].double()
        assert float((a - b).norm()) <= 2e-3 * float(a.norm()) + 2e-6 * top, k


def test_compact_pull_backward_with_a_popular_neighbour():
    """The on-the-spot inverted table of the compact attention backward when one destination row collects more than
    1024 (node, slot) pairs (a long row of the SpMM kernel) and pads are present: dQ / dEj equal the scatter form's."""
    from tagrec_amd import tgcn as TG
    n, n_nbr, k, D, A, n_wt = 6000, 500, 5, 32, 32, 4
    gen = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: (torch.randn(*s, generator=gen) * 0.3).to(DEV)
    idx = torch.randint(0, n_nbr + 1, (n, k), generator=gen)
    idx[:, 0] = 7                                             # every node lists destination 6 -> a long row
    idx[::3, 1] = 0                                           # pads
    idx32 = idx.to(DEV).to(torch.int32)
    widx32 = torch.randint(0, n_wt, (n, k), generator=gen).to(DEV).to(torch.int32)
    base = [rnd(n, A), rnd(n_nbr, A), rnd(n_wt, A), rnd(A), rnd(n_nbr, D)]
    d_out = torch.zeros(n, D, device=DEV)
    rows = torch.randperm(n, generator=gen)[:2500].to(DEV)
    d_out[rows] = rnd(2500, D)
    grads = []
    old, old_pull = TG._SPARSE_MIN_ROWS, TG._PULL_MIN_ROWS
    try:
        for pull in (10 ** 9, 0):
            TG._SPARSE_MIN_ROWS, TG._PULL_MIN_ROWS = 0, pull
            xs = [b.clone().requires_grad_() for b in base]
            TG.neighbour_attention(xs[0], xs[1], xs[2], xs[3], xs[4], idx32, widx32, None).backward(d_out)
            grads.append([x.grad.clone() for x in xs])
    finally:
        TG._SPARSE_MIN_ROWS, TG._PULL_MIN_ROWS = old, old_pull
    for a, b in zip(*grads):
        scale = float(a.abs().max()) + 1e-30
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol=3e-6 * scale)


@pytest.mark.parametrize("n,n_nbr,k,D,A", [(6000, 500, 5, 32, 32), (3000, 4000, 25, 128, 32), (2000, 100, 64, 64, 16), (70, 9, 3, 256, 32)])
def test_attention_backward_pulls_without_dh_equal_the_scatter_form(n, n_nbr, k, D, A):
    """The three-launch pull form of the attention backward (tagrec_attn_pull_da_f32 -> tagrec_tgcn_attn_bwd_ds_f32 ->
    tagrec_attn_pull_dq_f32: da formed by the destination-centric pull, 8 bytes per pair between the source- and the
    destination-centric halves) against the float-atomic scatter form of tagrec_tgcn_attn_bwd_f32: dP, dWT, dv, dQ, dEj,
    with pads, a destination that collects more than 1024 pairs (long row), destinations nobody points at, and the
    add-what-was-collected-so-far epilogues."""
    from tagrec_amd import tgcn_step as TS
    n_wt = 4
    gen = torch.Generator(device="cpu").manual_seed(n + k)
    rnd = lambda *s_: (torch.randn(*s_, generator=gen) * 0.3).to(DEV)
    idx = torch.randint(0, n_nbr + 1, (n, k), generator=gen)
    idx[:, 0] = 7                                             # every node lists destination 6 -> a long row when n > 1024
    idx[::3, 1 % k] = 0                                       # pads
    idx[idx == n_nbr] = 1                                     # destination n_nbr - 1 is never pointed at
    idx32 = idx.to(DEV).to(torch.int32).contiguous()
    widx32 = torch.randint(0, n_wt, (n, k), generator=gen).to(DEV).to(torch.int32)
    P, Q, WT, v, Ej = rnd(n, A), rnd(n_nbr, A), rnd(n_wt, A), rnd(A), rnd(n_nbr, D)
    d_out = rnd(n, D)
    _, attn = TS.attn_fwd(P, Q, WT, v, Ej, idx32, widx32)
    dQ0, dEj0 = torch.zeros(n_nbr, A, device=DEV), torch.zeros(n_nbr, D, device=DEV)
    dP0, dWT0, dv0 = TS.attn_bwd(P, Q, WT, v, Ej, idx32, widx32, attn, d_out, dQ0, dEj0, None)
    addQ, addX = rnd(n_nbr, A), rnd(n_nbr, D)
    old = TS.SEGMENTED_DQ
    try:
        for seg in (True, False):              # dQ by the segmented sum over the sorted pairs / by the row-per-wave pull
            TS.SEGMENTED_DQ = seg
            for aq, ax in ((None, None), (addQ, addX)):
                dP1, dWT1, dv1, dQ1, dEj1 = TS.attn_bwd_pulls(P, Q, WT, v, Ej, idx32, widx32, attn, d_out,
                                                              None if aq is None else aq.clone(), ax, w_major=1 if seg else -1)
                want = [dP0, dWT0, dv0, dQ0 if aq is None else dQ0 + aq, dEj0 if ax is None else dEj0 + ax]
                for name, a, b in zip(("dP", "dWT", "dv", "dQ", "dEj"), want, (dP1, dWT1, dv1, dQ1, dEj1)):
                    scale = float(a.abs().max()) + 1e-30
                    np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol=3e-6 * scale, err_msg=f"{name} seg={seg}")
    finally:
        TS.SEGMENTED_DQ = old


@pytest.mark.parametrize("renumber_dst", [False, True])
def test_sort_free_inversion_of_a_row_subset_equals_the_sorted_one(renumber_dst):
    """tagrec_inv_filter_i32: a step's inverted table as an order-preserving compaction of the relation's static pair list
    (sorted by destination once, at model start-up) instead of a radix sort per step -- same destination order, so the pulls
    sum in the same order: skey / pair / src / val identical to the sorted path's, and the attention backward built on it
    gives the same dP, dWT, dv, dQ, dEj.  With pads, a popular destination, rows outside the subset, and (renumber_dst) the
    destination table compact as well (layer >= 2 of the restricted step)."""
    from tagrec_amd import tgcn as TG, tgcn_step as TS
    n_all, n_nbr_all, k, D, A, n_wt = 5000, 3000, 7, 64, 32, 5
    gen = torch.Generator(device="cpu").manual_seed(11)
    rnd = lambda *s_: (torch.randn(*s_, generator=gen) * 0.3).to(DEV)
    idx_full = torch.randint(0, n_nbr_all + 1, (n_all, k), generator=gen)
    idx_full[:, 0] = 5
    idx_full[::4, 2] = 0
    if renumber_dst:                       # destinations restricted to a subset that holds every listed neighbour
        keep = torch.unique(torch.cat([torch.randperm(n_nbr_all, generator=gen)[:1200], torch.tensor([4])]))
        idx_full = torch.where(idx_full > 0, keep[torch.randint(0, keep.numel(), idx_full.shape, generator=gen)] + 1, idx_full)
        idx_full[:, 0] = 5
        pos_dst = torch.zeros(n_nbr_all + 1, dtype=torch.int32)
        pos_dst[keep + 1] = torch.arange(1, keep.numel() + 1, dtype=torch.int32)
        n_dst = keep.numel()
    else:
        pos_dst, n_dst = None, n_nbr_all
    idx_full = idx_full.to(DEV).to(torch.int32).contiguous()
    widx_full = torch.randint(0, n_wt, (n_all, k), generator=gen).to(DEV).to(torch.int32)
    inv = TG.InverseTable(idx_full, n_nbr_all)
    rows = torch.sort(torch.randperm(n_all, generator=gen)[:2100])[0].to(DEV)
    n = rows.numel()
    pos_src = torch.zeros(n_all + 1, dtype=torch.int32, device=DEV)
    pos_src[rows + 1] = torch.arange(1, n + 1, dtype=torch.int32, device=DEV)
    idx_c, widx_c = idx_full[rows], widx_full[rows].contiguous()
    if pos_dst is not None:
        pos_dst = pos_dst.to(DEV)
        idx_c = pos_dst[idx_c.long()]
    idx_c = idx_c.contiguous()
    P, Q, WT, v, Ej, d_out = rnd(n, A), rnd(n_dst, A), rnd(n_wt, A), rnd(A), rnd(n_dst, D), rnd(n, D)
    _, attn = TS.attn_fwd(P, Q, WT, v, Ej, idx_c, widx_c)
    want = TS.attn_bwd_pulls(P, Q, WT, v, Ej, idx_c, widx_c, attn, d_out, None, None)
    got = TS.attn_bwd_pulls(P, Q, WT, v, Ej, idx_c, widx_c, attn, d_out, None, None, static=(inv.perm32, inv.dest32, pos_src, pos_dst))
    for name, a, b in zip(("dP", "dWT", "dv", "dQ", "dEj"), want, got):
        scale = float(a.abs().max()) + 1e-30
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale, err_msg=name)
    # dEj sums in the same order on both paths: bit-identical
    assert torch.equal(want[4], got[4])


def test_tall_projection_weight_gradient_by_slabs():
    """`_TallMM`: X @ W whose weight gradient is summed slab by slab (n not a multiple of the slab count)."""
    from tagrec_amd import tgcn as TG
    gen = torch.Generator(device="cpu").manual_seed(3)
    X = torch.randn(70_001, 64, generator=gen).to(DEV).requires_grad_()
    W = (torch.randn(64, 32, generator=gen) * 0.1).to(DEV).requires_grad_()
    dY = torch.randn(70_001, 32, generator=gen).to(DEV)
    TG._tall_mm(X, W).backward(dY)
    gx, gw = X.grad.clone(), W.grad.clone()
    ref_w = (X.detach().double().t() @ dY.double())
    ref_x = dY.double() @ W.detach().double().t()
    assert float((gw.double() - ref_w).abs().max()) <= 1e-5 * float(ref_w.abs().max())
    assert float((gx.double() - ref_x).abs().max()) <= 1e-5 * float(ref_x.abs().max())


def test_sum_n_and_table_fan_out():
    """`sum_n` (one-pass sum of up to 8 tensors per launch, left to right) and `fan` (a table read by several consumers:
    aliases + row subsets whose gradients are folded by one n-way sum + index_add) against plain autograd."""
    torch.manual_seed(1)
    for shape in ((1003, 7), (4096, 128), (5,)):
        ts = [torch.randn(*shape, device=DEV) for _ in range(11)]
        ref = ts[0].clone()
        for t in ts[1:]:
            ref = ref + t
        assert torch.equal(TG.sum_n(ts[:3]), ts[0] + ts[1] + ts[2])          # same order of additions: bit-identical
        np.testing.assert_allclose(TG.sum_n(ts).cpu().numpy(), ref.cpu().numpy(), rtol=1e-6, atol=1e-6)
    x = torch.randn(300, 16, device=DEV, requires_grad=True)
    r1 = torch.randint(0, 300, (500,), device=DEV)
    r2 = torch.tensor([299, 0, 0], device=DEV)
    w = torch.randn(16, 4, device=DEV)

    def loss_of(a, b, c, s1, s2):
        return (a * a).sum() + (b @ w).sum() + (s1 ** 3).sum() + 2 * s2.sum()   # c: a reader that is never used

    (a, b, c), (s1, s2) = TG.fan(x, 3, r1, r2)
    loss_of(a, b, c, s1, s2).backward()
    got, x.grad = x.grad.clone(), None
    loss_of(x, x, x, x[r1], x[r2]).backward()
    np.testing.assert_allclose(got.cpu().numpy(), x.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        (a, b), (s1,) = TG.fan(x, 2, r1)
        assert a.data_ptr() == x.data_ptr() and torch.equal(s1, x[r1])


@pytest.mark.parametrize("sizes,k,B", [((1000, 800, 1500), 5, 64), ((5000, 4096, 8191), 25, 512), ((4095, 1, 3), 3, 7),
                                       ((70_000, 50_000, 90_000), 25, 512)])
def test_row_plan_kernels_equal_unique_and_nonzero(sizes, k, B):
    """csrc/plan.hip (mark + compact + lookup) against the torch formulation it replaces: torch.unique of the batch columns,
    boolean masks filled through the neighbour tables, torch.nonzero, arange position maps, searchsorted -- including a type
    whose rows are all needed (`all_types`), segment sizes on and next to the 4096-flag chunk boundary, and pad ids."""
    from tagrec_amd import plan as PL
    g = torch.Generator().manual_seed(sum(sizes) + k)
    names = ("user", "item", "tag")
    n = dict(zip(names, sizes))
    rel = (("user", "item"), ("user", "tag"), ("item", "user"), ("item", "tag"), ("tag", "user"), ("tag", "item"))
    nbr = []
    for src, nb in rel:                                         # ids 1-based, 0 = pad (a third of the slots)
        t = torch.randint(0, n[nb] + 1, (n[src], k), generator=g, dtype=torch.int32)
        t[torch.rand(n[src], k, generator=g) < 0.33] = 0
        nbr.append(t.to(DEV))
    batch = torch.stack([torch.randint(0, n["user"], (B,), generator=g), torch.randint(0, n["item"], (B,), generator=g),
                         torch.randint(0, n["item"], (B,), generator=g)], dim=1).to(DEV)
    plan = PL.RowPlan(n, torch.device(DEV))
    p0 = batch.data_ptr()
    rows, pos = plan.level([(None, p0, 3, B, 0, "user"), (None, p0 + 8, 3, B, 0, "item"), (None, p0 + 16, 3, B, 0, "item")])
    want = {"user": torch.unique(batch[:, 0]), "item": torch.unique(batch[:, 1:]), "tag": batch.new_empty(0)}
    for t in names:
        assert torch.equal(rows[t], want[t]), t
        ref = torch.zeros(n[t] + 1, dtype=torch.int32, device=DEV)
        ref[want[t] + 1] = torch.arange(1, want[t].numel() + 1, dtype=torch.int32, device=DEV)
        assert torch.equal(pos[t], ref), t
    trip = torch.empty(B, 3, dtype=torch.int64, device=DEV)
    for c, t in enumerate(("user", "item", "item")):
        PL.lookup(pos[t], p0 + 8 * c, stride=3, n=B, out=trip.data_ptr() + 8 * c, out_stride=3)
    assert torch.equal(trip[:, 0], torch.searchsorted(want["user"], batch[:, 0].contiguous()))
    assert torch.equal(trip[:, 2], torch.searchsorted(want["item"], batch[:, 2].contiguous()))
    # two levels down; at the second one the users are complete ("all rows")
    cur = dict(want)
    for level, all_types in ((0, ()), (1, ("user",))):
        if "user" in all_types:
            cur["user"] = None
        masks = {t: torch.zeros(n[t] + 1, dtype=torch.bool, device=DEV) for t in names}
        descs = []
        for t in names:
            if cur[t] is None:
                masks[t][1:] = True
            elif cur[t].numel():
                masks[t][cur[t] + 1] = True
                descs.append((None, cur[t], 1, cur[t].numel(), 0, t))
        for r, (src, nb) in enumerate(rel):
            if cur[src] is None:
                masks[nb][nbr[r].long().flatten()] = True
                descs.append((nbr[r], None, 1, n[src], n[src], nb))
            elif cur[src].numel():
                masks[nb][nbr[r].index_select(0, cur[src]).long().flatten()] = True
                descs.append((nbr[r], cur[src], 1, cur[src].numel(), n[src], nb))
        rows, pos = plan.level(descs, all_types=all_types)
        for t in names:
            w = torch.nonzero(masks[t][1:]).flatten()
            assert torch.equal(rows[t], w), (level, t)
            ref = torch.zeros(n[t] + 1, dtype=torch.int32, device=DEV)
            ref[w + 1] = torch.arange(1, w.numel() + 1, dtype=torch.int32, device=DEV)
            assert torch.equal(pos[t], ref), (level, t)
            assert torch.equal(PL.lookup(pos[t], w), torch.arange(w.numel(), device=DEV))
        cur = {t: rows[t].clone() for t in names}
    # an id outside the type is reported, not written
    bad = batch.clone()
    bad[3, 1] = n["item"]
    with pytest.raises(IndexError):
        plan.level([(None, bad.data_ptr() + 8, 3, B, 0, "item")])


def test_adam_fused_into_the_step_node_equals_the_separate_optimizer_pass():
    """`T.Adam(...).fuse_into(model)` on TGCN: the node tables' update rides in the epilogue of the product that forms the
    last term of their gradient (tagrec_tall_mm_adam_f32); `.grad` of the tables stays None and `step()` only counts.
    Four steps (BPR phase) + one TransTag step (which hands out gradients as usual) at 200 k nodes against the un-fused
    optimizer: same losses, the same parameters and Adam state up to the order in which the gradient's terms are added
    (Adam divides by sqrt(v): an element whose gradient is ~1e-7 may move by a fraction of lr either way)."""
    ds = T.synth.make_tripartite_device(50_000, 50_000, 100_000, 3_000_000, seed=6, device=DEV)
    cfg = T.get_config("tgcn", dim_latent=64, dim_layer_list=[64, 64, 64], device=DEV, train_batch=256, neighbor_k=25, reg=1e-4)
    prod = T.BPR_training_data(ds, config=cfg, seed=5)
    batches = [prod.all_train_data[i * 256:(i + 1) * 256] for i in range(4)]
    lr = 0.005
    runs = []
    for fuse in (False, True):
        torch.manual_seed(1)
        m = T.TGCN(ds, config=cfg)
        m.train()
        opt = T.Adam(m.parameters(), lr=lr)
        if fuse:
            opt.fuse_into(m)
        losses = []
        for b in batches:
            lossx = m.loss(b)
            opt.zero_grad()
            sum(lossx).backward()
            if fuse:
                assert all(p.grad is None for p in m.fused_tables()) and m.embed["weight"].grad is not None
                with pytest.raises(T.TagrecError):                    # a second fused backward before step()
                    sum(m.loss(b)).backward()
            opt.step()
            losses.append([float(v.detach()) for v in lossx])
        gq = torch.Generator(device=DEV).manual_seed(8)                # quad = (user, tag, positive item, negative item)
        tt = torch.stack([batches[0][:, 0], torch.randint(0, 100_000, (256,), device=DEV, generator=gq), batches[0][:, 1],
                          batches[0][:, 2]], dim=1)
        lt = m.transtag_loss(tt)
        opt.zero_grad()
        sum(lt).backward()
        assert all(p.grad is not None for p in m.fused_tables())
        opt.step()
        state = {k: p.detach().clone() for k, p in m.named_parameters()}
        adam = {k: (opt.state[id(p)]["m"].clone(), opt.state[id(p)]["v"].clone(), opt.state[id(p)]["t"]) for k, p in m.named_parameters()}
        runs.append((losses, state, adam))
    (l0, s0, a0), (l1, s1, a1) = runs
    np.testing.assert_allclose(l1, l0, rtol=2e-5)
    for k in s0:
        assert a0[k][2] == a1[k][2] == (5 if k in ("embed.user", "embed.item", "embed.tag") else 4), k   # (TransTag: the tables only)
        d = (s1[k] - s0[k]).abs()
        assert float(d.max()) <= 2 * 5 * lr, (k, float(d.max()))
        if k in ("embed.user", "embed.item", "embed.tag"):             # the fused tensors themselves: element by element
            assert float((d > 0.02 * lr).float().mean()) < 2e-3, (k, float((d > 0.02 * lr).float().mean()))
            np.testing.assert_allclose(a1[k][1].cpu().numpy(), a0[k][1].cpu().numpy(), rtol=1e-3, atol=1e-12, err_msg=k)
        # (the small tensors -- biases whose gradient is a sum that cancels to ~1e-8 -- take steps of +-lr on rounding noise:
        #  the two runs' tables differ in the last bit after the first step, which is enough to flip such a sign)
