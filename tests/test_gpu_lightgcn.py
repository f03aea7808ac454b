"""GPU parity: HIP path (through the C ABI) vs the CPU oracle and the reference's golden vectors.

Tolerances (fp32, summation order differs from torch's CPU CSR loop):
  activations / SpMM outputs   rtol 1e-5, atol 1e-6
  loss scalars                  rtol 1e-5
  gradients                     rtol 1e-3, atol 1e-7 * scale (sums of cancelling terms)
  parameters after Adam         atol 2e-4 (= 2 % of lr): Adam maps g -> lr*g/(|g|+1e-8), which amplifies
                                last-bit differences of near-zero gradients (see test_oracle_golden)
"""
import numpy as np
import pytest
import torch

from conftest import blocks_from_fixture

pytestmark = pytest.mark.gpu

import tagrec_amd as T
from tagrec_amd import graph as G
from tagrec_amd import help as H
from tagrec_amd.synth import Coo
from oracle import adj as oadj
from oracle import models as om

DEV = torch.device("cuda:0")


def _ds_from_fixture(fx):
    ds = T.synth.Dataset()
    use_tag = int(fx["use_tag"]) if "use_tag" in fx else 1
    ui, ut, it = blocks_from_fixture(fx, use_tag)
    ds.num = {"user": int(fx["n_user"]), "item": int(fx["n_item"])}
    ds.ui_adj = Coo(*ui)
    if use_tag:
        ds.num["tag"] = int(fx["n_tag"])
        ds.ut_adj, ds.it_adj = Coo(*ut), Coo(*it)
    return ds


def _oracle_csr(fx, norm=None, use_tag=None):
    use_tag = int(fx["use_tag"]) if use_tag is None else use_tag
    norm = str(fx["norm_type"]) if norm is None else norm
    return oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, use_tag)), norm)


def _graph(csr, symmetric=False):
    return T.Graph.from_host(csr.rowptr, csr.col, csr.val, csr.shape, DEV, symmetric=symmetric)


def _model(fx, name="lightgcn"):
    cfg = T.get_config(name, use_tag=bool(int(fx["use_tag"])), dim_layer_list=[int(x) for x in fx["layers"]],
                       dim_latent=int(fx["D"]), reg=float(fx["reg"]), device=DEV)
    m = T.LightGCN(_ds_from_fixture(fx), config=cfg)
    sd = {k[5:]: torch.from_numpy(fx[k]) for k in fx if k.startswith("init.")}
    m.load_state_dict(sd)
    return m


# ------------------------------------------------------------------ K1 SpMM
@pytest.mark.parametrize("D", [64, 32, 16, 128, 256, 10, 100, 4])
@pytest.mark.parametrize("norm", ["bi_norm", "ngcf"])
def test_spmm_matches_oracle(golden, D, norm):
    fx = golden("adj_toy")
    csr = _oracle_csr(fx, norm, 1)
    g = _graph(csr)
    torch.manual_seed(D)
    X = torch.randn(csr.shape[1], D)
    want = torch.sparse.mm(om.csr_to_torch(csr), X)
    got = g.spmm(X.to(DEV)).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    # transpose (backward operand)
    want_t = torch.sparse.mm(om.csr_to_torch(csr.transpose()), X)
    got_t = g.transpose().spmm(X.to(DEV)).cpu()
    np.testing.assert_allclose(got_t.numpy(), want_t.numpy(), rtol=1e-5, atol=1e-6)


def _star_graph(n, hub_deg, rng):
    """Rows of every length class: empty rows, short rows, a hub beyond the long-row threshold."""
    rows = [np.zeros(hub_deg, np.int64), rng.randint(1, n, size=4 * n)]
    cols = [rng.choice(n, hub_deg, replace=False), rng.randint(0, n, size=4 * n)]
    rows.append(np.full(1500, 7, np.int64)); cols.append(rng.choice(n, 1500, replace=False))   # 1024 < deg < 2*1024
    r, c = np.concatenate(rows), np.concatenate(cols)
    keep = r != 5                                       # row 5 stays empty
    v = rng.rand(len(r)).astype(np.float32)
    return oadj.coo_to_csr(r[keep], c[keep], v[keep], (n, n))


@pytest.mark.parametrize("D", [64, 128, 32])
def test_spmm_long_rows_and_empty_rows(D):
    rng = np.random.RandomState(0)
    csr = _star_graph(6000, 5000, rng)
    g = _graph(csr)
    info = g.info()
    assert info["n_long_rows"] == 2 and info["n_chunks"] == 10 + 3
    X = torch.randn(6000, D, generator=torch.Generator().manual_seed(1))
    want = torch.sparse.mm(om.csr_to_torch(csr), X)
    got = g.spmm(X.to(DEV)).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=1e-4)
    assert torch.all(got[5] == 0)
    # deterministic: the chunked reduction has a fixed order
    again = g.spmm(X.to(DEV)).cpu()
    assert torch.equal(got, again)


def test_graph_like_shares_the_long_row_list():
    """`Graph.like`: a second matrix over the same row pointer (own columns / values, another column count) gives
    the product of a graph created from scratch, long rows included, and outlives nothing it does not hold."""
    rng = np.random.RandomState(2)
    csr = _star_graph(6000, 5000, rng)
    g = _graph(csr)
    gen = torch.Generator().manual_seed(4)
    col2 = torch.randint(0, 777, (g.nnz,), generator=gen, dtype=torch.int32).to(DEV)
    val2 = torch.randn(g.nnz, generator=gen).to(DEV)
    twin = g.like(col2, val2, 777)
    fresh = T.Graph(g.rowptr, col2, val2, (6000, 777))
    assert twin.info()["n_long_rows"] == fresh.info()["n_long_rows"] == 2 and twin.shape == (6000, 777)
    X = torch.randn(777, 64, generator=gen).to(DEV)
    assert torch.equal(twin.spmm(X), fresh.spmm(X))
    with pytest.raises(T.TagrecError):
        g.like(col2[:-1], val2[:-1], 777)
    del g                                   # the twin keeps its parent (and the shared work list) alive
    assert torch.equal(twin.spmm(X), fresh.spmm(X))


def test_workspace_backed_graph_handles():
    """`Graph(..., workspace=True)` and its `like()` twins keep their device metadata (long-row work list, partial-sum slab)
    in a torch tensor instead of hipMalloc'ed memory -- the short-lived matrices of the TGCN attention backward -- and must
    give bit-identical products, long rows and every vector width included; an empty matrix works too."""
    rng = np.random.RandomState(3)
    csr = _star_graph(6000, 5000, rng)
    ref = _graph(csr)
    g = T.Graph(ref.rowptr, ref.col, ref.val, ref.shape, workspace=True)
    assert g.info() == ref.info() and g.info()["n_long_rows"] == 2
    gen = torch.Generator().manual_seed(5)
    for D in (8, 64, 256):
        X = torch.randn(ref.shape[1], D, generator=gen).to(DEV)
        assert torch.equal(g.spmm(X), ref.spmm(X))
    col2 = torch.randint(0, 333, (ref.nnz,), generator=gen, dtype=torch.int32).to(DEV)
    val2 = torch.randn(ref.nnz, generator=gen).to(DEV)
    twin, fresh = g.like(col2, val2, 333), T.Graph(ref.rowptr, col2, val2, (6000, 333))
    X = torch.randn(333, 32, generator=gen).to(DEV)
    assert torch.equal(twin.spmm(X), fresh.spmm(X)) and torch.equal(g.spmm(torch.ones(ref.shape[1], 8, device=DEV)),
                                                                    ref.spmm(torch.ones(ref.shape[1], 8, device=DEV)))
    empty = T.Graph(torch.zeros(5, dtype=torch.int64, device=DEV), torch.zeros(0, dtype=torch.int32, device=DEV),
                    torch.zeros(0, device=DEV), (4, 7), workspace=True)
    assert float(empty.spmm(torch.ones(7, 8, device=DEV)).abs().sum()) == 0.0


def test_deferred_graph_handles_need_no_host_read_and_give_the_same_products():
    """`Graph(..., workspace=True, deferred=True)`: the handle of a matrix built inside a training step without the host read
    of the long-row counters (the TGCN step's inverted tables) -- the work list is sized by its upper bounds and its unused
    slots are skipped.  Bit-identical products with the ordinary handle, long rows, every vector width, `like()` twins and the
    masked / flagged hops included."""
    rng = np.random.RandomState(3)
    csr = _star_graph(6000, 5000, rng)
    ref = _graph(csr)
    g = T.Graph(ref.rowptr, ref.col, ref.val, ref.shape, workspace=True, deferred=True)
    assert g.info()["n_long_rows"] >= 2                               # an upper bound, not a count
    gen = torch.Generator().manual_seed(5)
    for D in (8, 64, 256):
        X = torch.randn(ref.shape[1], D, generator=gen).to(DEV)
        assert torch.equal(g.spmm(X), ref.spmm(X))
    col2 = torch.randint(0, 333, (ref.nnz,), generator=gen, dtype=torch.int32).to(DEV)
    val2 = torch.randn(ref.nnz, generator=gen).to(DEV)
    twin, fresh = g.like(col2, val2, 333), T.Graph(ref.rowptr, col2, val2, (6000, 333))
    X = torch.randn(333, 32, generator=gen).to(DEV)
    assert torch.equal(twin.spmm(X), fresh.spmm(X))
    # a masked hop with a few flagged operand rows (the row-per-lane-group kernel + the chunked long rows)
    n_r, n_c = ref.shape
    X = torch.randn(n_c, 64, generator=gen).to(DEV)
    flags = torch.zeros(n_c, dtype=torch.uint8, device=DEV)
    flags[torch.randint(0, n_c, (40,), generator=gen).to(DEV)] = 1
    X = X * flags[:, None]
    mask = (torch.rand(n_r, generator=gen) < 0.5).to(torch.uint8).to(DEV)
    B = torch.randn(n_r, 64, generator=gen).to(DEV)
    outs = []
    for h in (g, ref):
        o = torch.zeros(n_r, 64, device=DEV)
        h.spmm_axpy_sparse(X, flags, None, B, 0.5, o, row_mask=mask)
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    want = (torch.sparse_csr_tensor(ref.rowptr, ref.col.long(), ref.val.double(), size=ref.shape) @ X.double() + 0.5 * B.double()).float()
    np.testing.assert_allclose((outs[0] * mask[:, None]).cpu().numpy(), (want * mask[:, None]).cpu().numpy(), rtol=2e-5, atol=2e-5)
    assert float((outs[0] * (1 - mask)[:, None].float()).abs().sum()) == 0.0          # rows outside the mask are not touched


def test_spmm_rejects_bad_arguments():
    rng = np.random.RandomState(1)
    csr = oadj.coo_to_csr(rng.randint(0, 100, 500), rng.randint(0, 100, 500), rng.rand(500), (100, 100))
    g = _graph(csr)
    with pytest.raises(T.TagrecError):
        g.spmm(torch.randn(100, 64))                       # CPU tensor
    with pytest.raises(T.TagrecError):
        g.spmm(torch.randn(99, 64, device=DEV))            # wrong rows
    with pytest.raises(T.TagrecError):
        g.spmm(torch.randn(100, 600, device=DEV))          # width beyond the scalar kernel
    x = torch.randn(100, 64, device=DEV)
    with pytest.raises(T.TagrecError):
        g.spmm(x, out=x)                                   # aliasing


# ------------------------------------------------------------------ fused layers vs golden
@pytest.mark.parametrize("name", ["lightgcn_toy", "lightgcn_med", "lightgcn_toy_d32", "lightgcn_toy_d256"])
def test_lightgcn_forward_golden(golden, name):
    fx = golden(name)
    m = _model(fx)
    m.eval()
    with torch.no_grad():
        outs = m.forward()
    for t, o in enumerate(outs):
        np.testing.assert_allclose(o.cpu().numpy(), fx[f"out.{t}"], rtol=1e-5, atol=1e-6)
    # raw per-layer products through the operator
    x = m.table.detach()
    for k in range(len(fx["layers"])):
        x = H.split_mm(m.norm_adj, x)
        np.testing.assert_allclose(x.cpu().numpy(), fx[f"raw.{k}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["lightgcn_toy", "lightgcn_med", "lightgcn_toy_d32", "lightgcn_toy_d256"])
def test_lightgcn_loss_and_grads_golden(golden, name):
    fx = golden(name)
    m = _model(fx)
    m.train()
    lossx = m.loss(torch.from_numpy(fx["batches"][0]).to(DEV))
    np.testing.assert_allclose([float(v) for v in lossx], fx["loss_parts"], rtol=1e-5, atol=1e-8)
    sum(lossx).backward()
    want = np.concatenate([fx[f"grad.embed.{t}"] for t in range(len(m.num_list))])
    got = m.table.grad.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-7 * np.abs(want).max() / 1e-4)


@pytest.mark.parametrize("fused_opt", [True, False])
@pytest.mark.parametrize("name", ["lightgcn_toy", "lightgcn_med", "lightgcn_toy_d256"])
def test_lightgcn_adam_steps_golden(golden, name, fused_opt):
    fx = golden(name)
    for n_steps in (1, 3):
        m = _model(fx)
        m.train()
        opt = T.Adam(m.parameters(), lr=float(fx["lr"])) if fused_opt else torch.optim.Adam(m.parameters(), lr=float(fx["lr"]))
        prod = T.Fixed_training_data([np.concatenate(fx["batches"][:n_steps])], fx["batches"].shape[1], DEV)
        # Fixed_training_data + the tail rule would merge batches; drive the step surface batch by batch
        prod.mini_batch = lambda: iter([torch.from_numpy(b).to(DEV) for b in fx["batches"][:n_steps]])
        losses = T.epoch_training(prod, m.loss, opt, verbose=False)
        np.testing.assert_allclose(losses, fx[f"step{n_steps}.losses"], rtol=2e-5)
        sd = m.state_dict()
        for t in range(len(m.num_list)):
            got, want = sd[f"embed.{t}"].cpu().numpy(), fx[f"step{n_steps}.embed.{t}"]
            assert np.abs(got - want).max() <= 2e-4
            assert np.mean(np.abs(got - want) <= 2e-5) >= 0.995


def test_lightgcn_unfused_path_matches_fused(golden):
    """Row folds (split_adj_k) go operator by operator through autograd; same numbers."""
    fx = golden("lightgcn_toy")
    m = _model(fx)
    cfg = T.get_config("lightgcn", use_tag=True, dim_layer_list=[64, 64], reg=float(fx["reg"]), device=DEV, split_adj_k=3)
    m2 = T.LightGCN(_ds_from_fixture(fx), config=cfg)
    m2.load_state_dict(m.state_dict())
    assert isinstance(m2.norm_adj, list) and len(m2.norm_adj) == 3
    b = torch.from_numpy(fx["batches"][0]).to(DEV)
    l1, l2 = m.loss(b), m2.loss(b)
    np.testing.assert_allclose([float(v) for v in l2], [float(v) for v in l1], rtol=1e-5)
    sum(l1).backward(); sum(l2).backward()
    np.testing.assert_allclose(m2.table.grad.cpu().numpy(), m.table.grad.cpu().numpy(), rtol=1e-3, atol=1e-9)


def test_predict_rating_golden(golden):
    fx = golden("lightgcn_toy")
    m = _model(fx)
    m.load_state_dict({k[6:]: torch.from_numpy(fx[k]) for k in fx if k.startswith("step3.embed.")})
    m.eval()
    r = m.predict_rating(torch.from_numpy(fx["predict.users"]).to(DEV))
    np.testing.assert_allclose(r.cpu().numpy(), fx["predict.rating"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ operators vs torch autograd
def test_operator_mul_loss_and_normalize():
    torch.manual_seed(0)
    for kind in ("softplus", "logsigmoid"):
        u, p, n = (torch.randn(77, 48) * 3 for _ in range(3))
        want_in = [t.clone().requires_grad_() for t in (u, p, n)]
        want = om.mul_loss(*want_in, kind)
        want.backward()
        got_in = [t.clone().to(DEV).requires_grad_() for t in (u, p, n)]
        got = H.mul_loss(*got_in, kind)
        got.backward()
        np.testing.assert_allclose(float(got), float(want), rtol=1e-5)
        for a, b in zip(got_in, want_in):
            np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-7)
    x = torch.randn(50, 40)
    x[3] = 0                                             # clamped row: eps branch
    xr = x.clone().requires_grad_()
    w = torch.randn(50, 40)
    (torch.nn.functional.normalize(xr, p=2, dim=1) * w).sum().backward()
    xg = x.clone().to(DEV).requires_grad_()
    (H.normalize_rows(xg) * w.to(DEV)).sum().backward()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)
    big = torch.randn(4, 8) * 60                         # softplus threshold / saturated sigmoid
    l = H.mul_loss(big.to(DEV), (-big).to(DEV), big.to(DEV), "softplus")
    np.testing.assert_allclose(float(l), float(om.mul_loss(big, -big, big, "softplus")), rtol=1e-6)


def test_adam_kernel_matches_torch():
    torch.manual_seed(1)
    p0 = torch.randn(1000, 64)
    pr = p0.clone().requires_grad_()
    pg = torch.nn.Parameter(p0.clone().to(DEV))
    o_ref, o_got = torch.optim.Adam([pr], lr=0.01), T.Adam([pg], lr=0.01)
    for it in range(5):
        g = torch.randn(1000, 64) * (10.0 ** -it)
        pr.grad = g.clone(); pg.grad = g.clone().to(DEV)
        o_ref.step(); o_got.step()
    np.testing.assert_allclose(pg.detach().cpu().numpy(), pr.detach().numpy(), rtol=1e-5, atol=1e-6)
    odd = torch.nn.Parameter(torch.randn(13, 7).to(DEV))   # length not a multiple of 4: tail kernel
    ref = odd.detach().cpu().clone().requires_grad_()
    g = torch.randn(13, 7)
    odd.grad, ref.grad = g.to(DEV), g.clone()
    T.Adam([odd], lr=0.1).step(); torch.optim.Adam([ref], lr=0.1).step()
    np.testing.assert_allclose(odd.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_sharded_model_with_hip_ops_world1(golden):
    """tagrec_amd.dist.ShardedLightGCN with the real HipOps on one rank (trivial collectives):
    must reproduce the single-GPU model; the multi-rank plumbing is covered on CPU in test_dist_gloo."""
    import os
    import torch.distributed as dist
    from tagrec_amd import dist as TD
    fx = golden("lightgcn_toy")
    m = _model(fx)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        csr = _oracle_csr(fx)
        cfg = T.get_config("lightgcn", use_tag=True, dim_layer_list=[64, 64], reg=float(fx["reg"]), device=DEV)
        ds = _ds_from_fixture(fx)
        sm = TD.ShardedLightGCN(ds, cfg, torch.from_numpy(csr.rowptr).to(DEV), torch.from_numpy(csr.col).to(DEV),
                                torch.from_numpy(csr.val).to(DEV), csr.shape[0])
        with torch.no_grad():
            sm.table.copy_(m.table)
        b = torch.from_numpy(fx["batches"][0]).to(DEV)
        l1, l2 = m.loss(b), sm.loss(b)
        np.testing.assert_allclose([float(v) for v in l2], [float(v) for v in l1], rtol=1e-6)
        sum(l1).backward(); sum(l2).backward()
        np.testing.assert_allclose(sm.table.grad.cpu().numpy(), m.table.grad.cpu().numpy(), rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose([float(v) for v in l2], fx["loss_parts"], rtol=1e-5)
    finally:
        dist.destroy_process_group()


def test_feature_sharded_model_with_hip_ops_world1(golden):
    """dist.FeatureShardedLightGCN with the real HipOps on one rank: the column-sharded kernel chain (spmm_ss ->
    row_scale_acc -> bpr_dots -> row_dot -> rownorm_bwd_dot -> spmm_normbwd_dot -> spmm_axpy) must reproduce the
    single-GPU fused model and the reference's fixture; the multi-rank reductions are covered in test_dist_gloo."""
    import os
    import torch.distributed as dist
    from tagrec_amd import dist as TD
    fx = golden("lightgcn_toy")
    m = _model(fx)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29573")
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        csr = _oracle_csr(fx)
        cfg = T.get_config("lightgcn", use_tag=True, dim_layer_list=[64, 64], reg=float(fx["reg"]), device=DEV)
        ds = _ds_from_fixture(fx)
        sm = TD.FeatureShardedLightGCN(ds, cfg, torch.from_numpy(csr.rowptr).to(DEV), torch.from_numpy(csr.col).to(DEV),
                                       torch.from_numpy(csr.val).to(DEV), csr.shape[0])
        with torch.no_grad():
            sm.table.copy_(m.table)
        b = torch.from_numpy(fx["batches"][0]).to(DEV)
        l1, l2 = m.loss(b), sm.loss(b)
        np.testing.assert_allclose([float(v) for v in l2], [float(v) for v in l1], rtol=1e-6)
        sum(l1).backward(); sum(l2).backward()
        np.testing.assert_allclose(sm.table.grad.cpu().numpy(), m.table.grad.cpu().numpy(), rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose([float(v) for v in l2], fx["loss_parts"], rtol=1e-5)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("G", [2, 4, 8])
def test_feature_shard_kernels_compose_to_full_width(golden, G):
    """The column-sharded kernels on D/G-wide slices (D/G = 32, 16, 8), reduced on the host the way the all-reduces
    do, equal the full-width fused kernels: forward layer (norm + layer mean), normalise-backward layer, scores."""
    from tagrec_amd import dist as TD
    fx = golden("lightgcn_toy")
    m = _model(fx)
    g, ops, D = m._graph(), TD.HipOps(), 64
    n = g.shape[0]
    gen = torch.Generator(device="cpu").manual_seed(G)
    x = torch.randn(n, D, generator=gen).to(DEV)
    dz = torch.randn(n, D, generator=gen).to(DEV)
    gin = torch.randn(n, D, generator=gen).to(DEV)
    s = 1.0 / 3
    # full width
    y_full, inv_full, acc_full = torch.empty_like(x), torch.empty(n, device=DEV), torch.zeros_like(x)
    g.spmm_norm_acc(x, y_full, inv_full, acc_full, s)
    out_full = torch.empty_like(x)
    g.spmm_normbwd(gin, y_full, inv_full, dz, s, out_full)
    last_full = torch.empty_like(x)
    ops.rownorm_bwd(y_full, inv_full, dz, s, last_full)
    # column slices
    Dl = D // G
    sl = [slice(k * Dl, (k + 1) * Dl) for k in range(G)]
    ys, ss = [], torch.zeros(n, device=DEV)
    for c in sl:
        y, p = torch.empty(n, Dl, device=DEV), torch.empty(n, device=DEV)
        ops.spmm_ss(g, x[:, c].contiguous(), y, p)
        ys.append(y)
        ss += p
    inv = 1.0 / torch.sqrt(ss).clamp_min(1e-12)
    np.testing.assert_allclose(inv.cpu().numpy(), inv_full.cpu().numpy(), rtol=2e-6)
    dot = torch.zeros(n, device=DEV)
    for k, c in enumerate(sl):
        acc = torch.zeros(n, Dl, device=DEV)
        ops.row_scale_acc(ys[k], inv, s, acc)
        np.testing.assert_allclose(acc.cpu().numpy(), acc_full[:, c].cpu().numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ys[k].cpu().numpy(), y_full[:, c].cpu().numpy(), rtol=1e-5, atol=1e-6)
        p = torch.empty(n, device=DEV)
        ops.row_dot(ys[k], inv, dz[:, c].contiguous(), s, p)
        dot += p
    scale = float(out_full.abs().max())
    for k, c in enumerate(sl):
        o = torch.empty(n, Dl, device=DEV)
        ops.spmm_normbwd_dot(g, gin[:, c].contiguous(), ys[k], inv, dz[:, c].contiguous(), dot, s, o)
        np.testing.assert_allclose(o.cpu().numpy(), out_full[:, c].cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)
        o2 = torch.empty(n, Dl, device=DEV)
        ops.rownorm_bwd_dot(ys[k], inv, dz[:, c].contiguous(), dot, s, o2)
        np.testing.assert_allclose(o2.cpu().numpy(), last_full[:, c].cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)
    # scores
    nu, ni = int(fx["n_user"]), int(fx["n_item"])
    trip = torch.from_numpy(fx["batches"][0]).to(DEV)
    dots = torch.zeros(trip.shape[0], 3, device=DEV)
    for c in sl:
        xs = x[:, c].contiguous()
        dots += ops.bpr_dots(xs[:nu], xs[nu:nu + ni], xs[:nu], xs[nu:nu + ni], trip)
    u, p_, n_ = x[:nu][trip[:, 0]], x[nu:nu + ni][trip[:, 1]], x[nu:nu + ni][trip[:, 2]]
    want = torch.stack([(u * p_).sum(1), (u * n_).sum(1), 0.5 * (u.pow(2).sum(1) + p_.pow(2).sum(1) + n_.pow(2).sum(1))], 1)
    np.testing.assert_allclose(dots.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n_layer", [3, 2, 1])
def test_restricted_forward_equals_full_forward_step(n_layer):
    """`LightGCN.loss` computes the top two layers only on the rows the batch's loss depends on (batch rows; their
    neighbours one layer down): same loss parts and the same table gradient as with every layer on all rows, on a graph
    with long rows (popular items) so that the chunked path is masked too."""
    ds = T.synth.make_bipartite_device(30_000, 20_000, 1_500_000, seed=3, device=DEV)
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], 30_000, 20_000, "bi_norm")
    g = T.Graph(rp, col, val, (n, n), symmetric=True)
    assert g.info()["n_long_rows"] > 0
    cfg = T.get_config("lightgcn", use_tag=False, dim_latent=64, dim_layer_list=[64] * n_layer, device=DEV, train_batch=128,
                       reg=1e-3)
    torch.manual_seed(2)
    m = T.LightGCN(ds, config=cfg, graph=g)
    m.train()
    batch = T.BPR_training_data(ds, config=cfg, seed=1).all_train_data[:128]
    res = []
    for restrict in (False, True):
        m.restrict_forward = restrict
        m.zero_grad()
        lossx = m.loss(batch)
        sum(lossx).backward()
        res.append(([float(v) for v in lossx], m.table.grad.clone()))
    (l0, g0), (l1, g1) = res
    np.testing.assert_allclose(l1, l0, rtol=2e-6)
    scale = float(g0.abs().max())
    # the restricted step sums in another order (top layer: four waves per batch row; flagged entries of a row are packed
    # before they are gathered), so entries that cancel to ~1e-6 of the largest gradient differ in their last digits
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=1e-3, atol=1e-5 * scale)


def test_feature_sharded_restricted_step_equals_single_gpu_model():
    """FeatureShardedLightGCN (one rank, real kernels) on a graph large enough for its restricted forward and the
    row-masked backward hop (B * 48 <= N): loss parts and table gradient of the single-GPU model, itself run with every
    layer on all rows."""
    import os
    import torch.distributed as dist
    from tagrec_amd import dist as TD
    ds = T.synth.make_bipartite_device(30_000, 20_000, 1_500_000, seed=3, device=DEV)
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], 30_000, 20_000, "bi_norm")
    cfg = T.get_config("lightgcn", use_tag=False, dim_latent=64, dim_layer_list=[64] * 3, device=DEV, train_batch=128, reg=1e-3)
    torch.manual_seed(2)
    m = T.LightGCN(ds, config=cfg, graph=T.Graph(rp, col, val, (n, n), symmetric=True))
    m.train()
    m.restrict_forward = False
    batch = T.BPR_training_data(ds, config=cfg, seed=1).all_train_data[:128]
    assert 3 * batch.shape[0] * 16 <= n
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29574")
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        sm = TD.FeatureShardedLightGCN(ds, cfg, rp, col, val, n)
        assert sm.ops.restrict_forward and sm.restrict_forward          # the restricted column-sharded step is what runs
        with torch.no_grad():
            sm.table.copy_(m.table)
        l1, l2 = m.loss(batch), sm.loss(batch)
        np.testing.assert_allclose([float(v) for v in l2], [float(v) for v in l1], rtol=2e-6)
        sum(l1).backward(); sum(l2).backward()
        g0, g1 = m.table.grad, sm.table.grad
        scale = float(g0.abs().max())
        np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=1e-4, atol=2e-6 * scale)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("D", [8, 64, 256, 20])
def test_row_flags_kernel(D):
    """`tagrec_row_flags_f32`: flag = row holds a non-zero, count = number of flagged rows; row count not a multiple of
    the rows a block covers."""
    from tagrec_amd import dist as TD
    n = 10_007
    gen = torch.Generator().manual_seed(D)
    x = torch.zeros(n, D)
    rows = torch.randperm(n, generator=gen)[:777]
    x[rows, torch.randint(0, D, (777,), generator=gen)] = 1.5
    x[n - 1, D - 1] = -0.0                                   # a negative zero is still a zero
    flags, count = TD.HipOps().row_flags(x.to(DEV))
    want = (x != 0).any(dim=1)
    assert torch.equal(flags.cpu().bool(), want) and int(count) == int(want.sum()) == 777


@pytest.mark.parametrize("shape", [(3000, 3000), (500, 40_000)])
def test_flagged_zero_rows_are_not_fetched(shape):
    """The row-flag contract of the backward products (include/tagrec.h, "ROW-SPARSE gradient"): while the flags cover less
    than 4/5 of the operand's rows, rows flagged zero are not read at all -- here they hold NaN, which any read would
    propagate.  The rectangular case is a row shard (few local rows, flags over the whole gathered table): the 4/5 rule is
    counted against the operand's rows, not the local ones.  Once the flags cover most rows they are ignored and the same
    call must read everything (operand cleaned first)."""
    n_r, n_c = shape
    gen = torch.Generator().manual_seed(n_c)
    deg = torch.randint(1, 40, (n_r,), generator=gen)
    deg[7] = 2500                                                   # one long row (chunked path)
    rp = torch.zeros(n_r + 1, dtype=torch.int64)
    torch.cumsum(deg, 0, out=rp[1:])
    col = torch.cat([torch.randperm(n_c, generator=gen)[:d].sort().values for d in deg.tolist()]).to(torch.int32)
    val = torch.rand(col.numel(), generator=gen) + 0.5
    g = T.Graph(rp.to(DEV), col.to(DEV), val.to(DEV), (n_r, n_c))
    D = 64
    for frac in (0.05, 0.95):
        keep = torch.rand(n_c, generator=gen) < frac
        x = torch.randn(n_c, D, generator=gen) * keep[:, None]
        b = torch.randn(n_r, D, generator=gen)
        want = (torch.sparse_csr_tensor(rp, col.long(), val.double(), size=shape) @ x.double() + 0.5 * b.double()).float()
        xd = x.clone()
        if frac < 0.8:
            xd[~keep] = float("nan")                                # must never be fetched
        flags = keep.to(torch.uint8).to(DEV)
        count = flags.sum(dtype=torch.int32).reshape(1)
        out = torch.empty(n_r, D, device=DEV)
        g.spmm_axpy_sparse(xd.to(DEV), flags, count, b.to(DEV), 0.5, out)
        got = out.cpu()
        assert torch.isfinite(got).all(), "a row flagged zero was fetched"
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-5)
