"""GPU: end-to-end Recall@20 parity at C1 scale, the evaluation path, the device-side producers, and
size-independent properties of the SpMM at the full C2 size (BASELINE.json configs[1])."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tagrec_amd as T
from oracle import data as odata

DEV = torch.device("cuda:0")


def test_recall_at_20_matches_reference_run(golden):
    """Same graph, same init (torch CPU seed 2020), same per-epoch triplets (numpy seed 2020+epoch),
    6 epochs of B=512 Adam steps: Recall@20 within +-1e-3 of the reference's own CPU run
    (BASELINE.json north_star), loss curve within 1e-3."""
    fx = golden("e2e_c1_lightgcn")
    ds = T.synth.make_cf_dataset()
    assert int(ds.edge_index["train"].astype(np.int64).sum()) == int(fx["edge_checksum"])
    cfg = T.get_config("lightgcn", use_tag=False, dim_layer_list=[64, 64], device=DEV, train_batch=int(fx["batch"]))
    torch.manual_seed(2020)
    model = T.LightGCN(ds, config=cfg)
    np.testing.assert_allclose(float(model.table.double().sum()), float(fx["init_sum"]), rtol=1e-9)
    opt = T.Adam(model.parameters(), lr=float(fx["lr"]))
    epochs = int(fx["epochs"])
    prod = T.Fixed_training_data([T.synth.sample_bpr_epoch(ds, 2020 + ep) for ep in range(epochs)], cfg["train_batch"], DEV)
    curve = []
    for _ in range(epochs):
        model.train()
        losses = T.epoch_training(prod, model.loss, opt, verbose=False)
        curve.append(float(np.mean(losses)))
    np.testing.assert_allclose(curve, fx["loss_curve"], atol=1e-3)
    res = T.Basic_test(ds, config=cfg).run(model)
    assert abs(res["recall"][1] - fx["res.recall"][1]) <= 1e-3, (res["recall"], fx["res.recall"])
    assert abs(res["recall"][0] - fx["res.recall"][0]) <= 1e-3
    np.testing.assert_allclose(res["ndcg"], fx["res.ndcg"], atol=2e-3)
    np.testing.assert_allclose(res["precision"], fx["res.precision"], atol=1e-3)
    np.testing.assert_allclose(res["hr"], fx["res.hr"], atol=5e-3)


class _FixedScores(torch.nn.Module):
    def __init__(self, rating):
        super().__init__()
        self.r = rating

    def predict_rating(self, users):
        return self.r[users].clone()


def test_basic_test_metrics_golden(golden):
    """Device-side mask -> top-k -> recall/precision/hr/ndcg against the reference's `test_users`
    output, and the rank-based AUC against a direct pair count."""
    fx = golden("metrics")
    nu, ni = fx["rating"].shape
    ds = T.synth.Dataset()
    ds.num = {"user": nu, "item": ni}
    ds.user_items = {"train": {u: fx[f"train.{u}"].tolist() for u in range(nu)},
                     "test": {u: fx[f"test.{u}"].tolist() for u in range(nu)}}
    cfg = T.get_config("lightgcn", device=DEV, test_batch=7)
    res = T.Basic_test(ds, config=cfg).run(_FixedScores(torch.from_numpy(fx["rating"]).to(DEV)))
    for k in ("recall", "precision", "hr", "ndcg"):
        np.testing.assert_allclose(res[k], fx["res." + k] / nu, rtol=1e-6, err_msg=k)
    # AUC: fraction of (test item, other unmasked item) pairs ranked correctly, mean over users
    auc = []
    for u in range(nu):
        r = fx["rating"][u].copy()
        keep = np.ones(ni, bool); keep[fx[f"train.{u}"]] = False
        pos = np.zeros(ni, bool); pos[fx[f"test.{u}"]] = True
        p, n = r[keep & pos], r[keep & ~pos]
        auc.append(((p[:, None] > n[None, :]).sum() + 0.5 * (p[:, None] == n[None, :]).sum()) / (len(p) * len(n)))
    np.testing.assert_allclose(res["auc"][0], np.mean(auc), rtol=1e-9)


def test_device_bpr_producer_properties():
    ds = T.synth.make_cf_dataset(300, 200, 6000, seed=5)
    cfg = T.get_config("lightgcn", device=DEV, train_batch=512)
    prod = T.BPR_training_data(ds, config=cfg, seed=1)
    a = prod.all_train_data.cpu().numpy()
    tr = ds.edge_index["train"]
    assert a.shape == (len(tr), 3)
    key = lambda x, y: x.astype(np.int64) * 200 + y
    assert np.array_equal(np.sort(key(a[:, 0], a[:, 1])), np.sort(key(tr[:, 0], tr[:, 1])))   # a permutation of the edges
    assert not np.isin(key(a[:, 0], a[:, 2]), key(tr[:, 0], tr[:, 1])).any()                   # negatives are never train items
    assert a[:, 2].min() >= 0 and a[:, 2].max() < 200
    prod.reset()
    b = prod.all_train_data.cpu().numpy()
    assert not np.array_equal(a, b)                                                            # re-sampled every epoch
    sizes = [x.shape[0] for x in prod.mini_batch()]
    assert sizes == [hi - lo for lo, hi in odata.mini_batch_bounds(len(tr), 512)]
    # negatives are uniform over the non-train items: Pearson chi-square on the busiest user
    u = np.bincount(tr[:, 0]).argmax()
    draws = np.concatenate([np.asarray(T.BPR_training_data(ds, config=cfg, seed=s).all_train_data.cpu())
                            for s in range(60)])
    neg = draws[draws[:, 0] == u][:, 2]
    free = np.setdiff1d(np.arange(200), tr[tr[:, 0] == u][:, 1])
    cnt = np.bincount(neg, minlength=200)[free].astype(np.float64)
    assert cnt.sum() == len(neg)
    exp = cnt.sum() / len(free)
    chi2 = ((cnt - exp) ** 2 / exp).sum()
    dof = len(free) - 1
    assert chi2 < dof + 6 * np.sqrt(2 * dof), (chi2, dof)


@pytest.fixture(scope="module")
def c2_graph():
    nu = ni = 1_000_000
    ds = T.synth.make_bipartite_device(nu, ni, 50_000_000, seed=1, device=DEV)
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], nu, ni)
    return T.Graph(rp, col, val, (n, n), symmetric=True), e


def test_c2_graph_shape(c2_graph):
    g, e = c2_graph
    assert g.shape == (2_000_000, 2_000_000) and g.nnz == 100_000_000
    assert e.shape == (50_000_000, 2)
    key = e[:, 0] * 1_000_000 + e[:, 1]
    assert bool((key[1:] > key[:-1]).all())                                   # distinct pairs
    deg = g.rowptr[1:] - g.rowptr[:-1]
    assert int(deg.min()) >= 1 and int(deg[1_000_000:].max()) <= 110_000     # degree >= 1, item cap
    assert g.info()["n_long_rows"] > 0                                        # the chunked path is exercised


def test_c2_spmm_properties(c2_graph):
    """At full size the oracle is too slow; check what must hold for ANY correct SpMM:
    linearity, self-adjointness of the symmetric bi_norm matrix, agreement with an independent
    torch formulation on sampled rows (incl. the longest), run-to-run determinism."""
    g, _ = c2_graph
    n, D = g.shape[0], 64
    gen = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(n, D, device=DEV, generator=gen)
    y = torch.randn(n, D, device=DEV, generator=gen)
    ax, ay = g.spmm(x), g.spmm(y)
    axy = g.spmm(2.0 * x - 0.5 * y)
    scale = float(ax.abs().mean())
    assert float((axy - (2.0 * ax - 0.5 * ay)).abs().max()) <= 2e-5 * max(1.0, scale * 50)
    lhs, rhs = float((ax.double() * y.double()).sum()), float((x.double() * ay.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * (abs(lhs) + abs(rhs)) + 1e-3
    deg = g.rowptr[1:] - g.rowptr[:-1]
    rows = torch.cat([torch.randint(0, n, (200,), device=DEV, generator=gen), torch.topk(deg, 3).indices])
    for r in rows.tolist():
        lo, hi = int(g.rowptr[r]), int(g.rowptr[r + 1])
        want = (g.val[lo:hi, None].double() * x[g.col[lo:hi].long()].double()).sum(0)
        np.testing.assert_allclose(ax[r].cpu().numpy(), want.float().cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert torch.equal(ax, g.spmm(x))


def test_c2_fused_layer_consistent_with_plain_spmm(c2_graph):
    g, _ = c2_graph
    n, D = g.shape[0], 64
    x = torch.randn(n, D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    y_raw, inv, acc = torch.empty_like(x), torch.empty(n, device=DEV), torch.zeros_like(x)
    g.spmm_norm_acc(x, y_raw, inv, acc, 0.25)
    plain = g.spmm(x)
    assert torch.equal(y_raw, plain)
    z = torch.nn.functional.normalize(plain, p=2, dim=1)
    assert float((acc - 0.25 * z).abs().max()) <= 1e-6
    np.testing.assert_allclose(inv.cpu().numpy(), (1.0 / plain.norm(dim=1).clamp_min(1e-12)).cpu().numpy(), rtol=1e-5)


def test_c5_width_restricted_step_equals_all_rows_step_at_full_c2_size(c2_graph):
    """C5's row width (dim 256, 3 layers) at the full C2 graph size (2 M nodes, nnz 100 M, rows of 1e5 entries): the
    restricted training step -- all-rows layer, row-masked layer, compact push-form top layer, flagged backward with
    packed entries, batch-row epilogue terms -- against the step that computes every layer on all rows: same loss parts,
    same table gradient.  (The whole C5 shape, 10 M x 10 M x 500 M edges, runs on one GPU too: tools/c5_one_gpu.py.)"""
    g, e = c2_graph
    nu = ni = 1_000_000
    ds = T.synth.Dataset()
    ds.num = {"user": nu, "item": ni}
    cfg = T.get_config("lightgcn", use_tag=False, dim_latent=256, dim_layer_list=[256] * 3, device=DEV, train_batch=512, reg=1e-3)
    torch.manual_seed(5)
    m = T.LightGCN(ds, config=cfg, graph=g)
    m.train()
    gen = torch.Generator(device=DEV).manual_seed(8)
    pick = torch.randint(0, e.shape[0], (512,), device=DEV, generator=gen)
    batch = torch.stack([e[pick, 0], e[pick, 1], torch.randint(0, ni, (512,), device=DEV, generator=gen)], 1)
    res = []
    for restrict in (False, True):
        m.restrict_forward = restrict
        m.zero_grad()
        lossx = m.loss(batch)
        sum(lossx).backward()
        res.append(([float(v) for v in lossx], m.table.grad.clone()))
        m.table.grad = None
    (l0, g0), (l1, g1) = res
    np.testing.assert_allclose(l1, l0, rtol=2e-6)
    scale = float(g0.abs().max())
    # |difference| <= 1e-3 |gradient| + 1e-5 of the largest entry, element by element (other summation order)
    assert float(((g1 - g0).abs() - 1e-3 * g0.abs()).max()) <= 1e-5 * scale


def test_c2_step_with_adam_in_the_last_hop_is_bit_identical(c2_graph):
    """BASELINE's configuration itself (C2: LightGCN L = 3, D = 64, 2 M nodes, nnz 100 M, B = 512): two training steps with
    the table's Adam update applied in the epilogue of the last backward product (`Adam.fuse_into`, what bench.py times)
    against the separate optimizer launch: table, exp_avg and exp_avg_sq bit-identical (batches without a repeated node,
    so that no atomic scatter reorders a sum between the two runs)."""
    g, e = c2_graph
    nu = ni = 1_000_000
    ds = T.synth.Dataset()
    ds.num = {"user": nu, "item": ni}
    cfg = T.get_config("lightgcn", use_tag=False, dim_latent=64, dim_layer_list=[64] * 3, device=DEV, train_batch=512)
    gen = torch.Generator(device=DEV).manual_seed(11)
    users = torch.randperm(nu, device=DEV, generator=gen)[:1024]
    items = torch.randperm(ni, device=DEV, generator=gen)[:2048]
    batches = [torch.stack([users[i * 512:(i + 1) * 512], items[i * 1024:i * 1024 + 512], items[i * 1024 + 512:(i + 1) * 1024]], 1)
               for i in range(2)]
    out = []
    for fuse in (False, True):
        torch.manual_seed(5)
        m = T.LightGCN(ds, config=cfg, graph=g)
        m.train()
        opt = T.Adam(m.parameters(), lr=0.01)
        if fuse:
            opt.fuse_into(m)
        losses = []
        for b in batches:
            lossx = m.loss(b)
            opt.zero_grad()
            sum(lossx).backward()
            assert (m.table.grad is None) == fuse
            opt.step()
            losses.append([float(v.detach()) for v in lossx])
        st = opt.state[id(m.table)]
        out.append((losses, m.table.detach().clone(), st["m"].clone(), st["v"].clone()))
        del m, opt
    (l0, t0, m0, v0), (l1, t1, m1, v1) = out
    assert l0 == l1
    assert torch.equal(t0, t1) and torch.equal(m0, m1) and torch.equal(v0, v1)


@pytest.fixture(scope="module")
def c3_graph(c2_graph):
    """C3 = the C2 graph under NGCF's normalisation D^-1 A + I (adj.py:82-83): not symmetric, so the backward multiplies by
    the transposed CSR built by Graph.transpose()."""
    _, e = c2_graph
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], 1_000_000, 1_000_000, "ngcf")
    return T.Graph(rp, col, val, (n, n))


def test_c3_transposed_graph_properties(c3_graph):
    """Full-size transposed 102 M-entry CSR: <A x, y> = <x, A^T y>, rows of A sum to 2 (D^-1 A is row-stochastic, + I),
    sampled rows of A^T y (the most popular columns included) against an fp64 sum over the matrix's own entries."""
    g = c3_graph
    gt = g.transpose()
    n, D = g.shape[0], 64
    assert g.nnz == 102_000_000 and gt.nnz == g.nnz and gt.shape == g.shape and not g.symmetric
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, D, device=DEV, generator=gen)
    y = torch.randn(n, D, device=DEV, generator=gen)
    ax, aty = g.spmm(x), gt.spmm(y)
    lhs, rhs = float((ax.double() * y.double()).sum()), float((x.double() * aty.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * (abs(lhs) + abs(rhs)) + 1e-3
    ones = g.spmm(torch.ones(n, 8, device=DEV))
    assert float((ones - 2.0).abs().max()) <= 1e-4
    deg_t = gt.rowptr[1:] - gt.rowptr[:-1]
    rows = torch.cat([torch.randint(0, n, (40,), device=DEV, generator=gen), torch.topk(deg_t, 2).indices])
    row_of = None
    for r in rows.tolist():
        idx = torch.nonzero(g.col == r).flatten()                       # entries of column r of A = row r of A^T
        src = torch.searchsorted(g.rowptr, idx, right=True) - 1
        want = (g.val[idx, None].double() * y[src].double()).sum(0)
        np.testing.assert_allclose(aty[r].cpu().numpy(), want.float().cpu().numpy(), rtol=1e-4, atol=1e-5)
        assert int(deg_t[r]) == idx.numel()


def test_c3_restricted_ngcf_step_equals_all_rows_step(c2_graph, c3_graph):
    """One NGCF training step at the full C3 size: the restricted step (top two layers' neighbour sums on the rows the
    loss depends on, top dense block on the batch rows, row-sparse backward through the transposed CSR) must give the
    loss and the gradients of the step that computes every layer on all rows."""
    from tagrec_amd import ngcf as NG
    _, e = c2_graph
    nu = ni = 1_000_000
    ds = T.synth.Dataset()
    ds.num = {"user": nu, "item": ni}
    cfg = T.get_config("ngcf", use_tag=False, dim_latent=64, dim_layer_list=[64, 64, 64], device=DEV, train_batch=512)
    torch.manual_seed(4)
    m = T.NGCF(ds, config=cfg, graph=c3_graph)
    m.train()
    gen = torch.Generator(device=DEV).manual_seed(6)
    pick = torch.randint(0, e.shape[0], (512,), device=DEV, generator=gen)
    batch = torch.stack([e[pick, 0], e[pick, 1], torch.randint(0, ni, (512,), device=DEV, generator=gen)], 1)
    res = []
    for restrict in (True, False):
        NG.RESTRICT_FORWARD = restrict
        try:
            m.zero_grad()
            lossx = m.loss(batch)
            sum(lossx).backward()
            res.append(([float(v) for v in lossx], {k: p.grad.clone() for k, p in m.named_parameters()}))
        finally:
            NG.RESTRICT_FORWARD = True
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=2e-6)
    for k, want in res[1][1].items():
        got = res[0][1][k]
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 1e-3 * scale + 1e-12, k
        assert float(want.abs().max()) > 0


@pytest.mark.parametrize("D,n_user,n_item,K", [(64, 300, 1000, 20), (256, 130, 515, 20), (16, 70, 33, 10), (192, 64, 200, 5)])
def test_fused_eval_topk_matches_torch_path(D, n_user, n_item, K):
    """csrc/eval.hip (score -> mask -> top-K in one pass) against sigmoid(U I^T) + mask + torch.topk."""
    from tagrec_amd import evaluate as EV
    gen = torch.Generator(device=DEV).manual_seed(D + n_item)
    U = torch.randn(n_user, D, device=DEV, generator=gen) * 0.3
    I = torch.randn(n_item, D, device=DEV, generator=gen) * 0.3
    mask = torch.rand(n_user, n_item, device=DEV, generator=gen) < 0.15          # train positives
    mask[5] = False
    mask[7, : n_item - max(1, K // 2)] = True                                    # fewer than K items left for user 7
    ptr = torch.zeros(n_user + 1, dtype=torch.int64, device=DEV)
    torch.cumsum(mask.sum(1), 0, out=ptr[1:])
    items = torch.nonzero(mask)[:, 1].to(torch.int32).contiguous()               # row-major -> sorted per user
    users = torch.randperm(n_user, device=DEV, generator=gen)[: n_user - 3]
    top, val = EV.fused_topk(U, I, users, ptr, items, K)
    rating = torch.sigmoid(U[users] @ I.t())
    rating[mask[users]] = -(1 << 10)
    wv, wi = torch.topk(rating, k=K)
    for row in range(users.numel()):
        n_free = int((~mask[users[row]]).sum())
        k_ok = min(K, n_free)
        np.testing.assert_allclose(val[row, :k_ok].cpu().numpy(), wv[row, :k_ok].cpu().numpy(), rtol=1e-5, atol=1e-6)
        got, want = top[row, :k_ok].tolist(), wi[row, :k_ok].tolist()
        if got != want:                              # only near-ties may differ in order
            assert sorted(got) == sorted(want) or np.allclose(val[row, :k_ok].cpu(), wv[row, :k_ok].cpu(), atol=1e-6)
        assert (top[row, k_ok:] == -1).all()         # nothing admitted beyond the un-masked items


def test_basic_test_fused_equals_batched_path():
    ds = T.synth.make_cf_dataset(400, 300, 9000, seed=12)
    cfg = T.get_config("lightgcn", use_tag=False, dim_layer_list=[64, 64], device=DEV, test_batch=128)
    torch.manual_seed(3)
    model = T.LightGCN(ds, config=cfg)
    opt = T.Adam(model.parameters(), lr=0.01)
    prod = T.BPR_training_data(ds, config=cfg, seed=1)
    model.train()
    T.epoch_training(prod, model.loss, opt, verbose=False)
    fused = T.Basic_test(ds, config=cfg, with_auc=False).run(model)
    slow = T.Basic_test(ds, config=dict(cfg, eval_fused=False), with_auc=False).run(model)
    for k in ("recall", "precision", "hr", "ndcg"):
        np.testing.assert_allclose(fused[k], slow[k], rtol=1e-9, atol=1e-12, err_msg=k)


def test_full_size_c4_training_step_restricted_equals_all_rows():
    """BASELINE.json configs[3] at FULL size (1 M users, 1 M items, 2 M tags, ~100 M assignments, D = 128, k = 25, 3 layers):
    one BPR step of the hand-derived step node (layers restricted to the rows the batch's loss depends on, compact tables,
    pull-form attention backward, own projections) against the all-rows step (`forward()` on 4 M nodes + autograd): same
    loss parts, same gradient of every parameter.  Sums run over different row sets / orders and the backward uses float
    atomics in places, so tensors are compared as a whole: ||a - b|| <= 2e-3 ||a|| + 1e-5 max ||.|| (the floor: the dense
    block's weight gradients of the upper layers are ~1e-8 here -- sums over up to 4 M rows of +-1e-3 terms that cancel --
    and both passes carry ~1e-9 of fp32 accumulation noise, measured 1.0e-9 .. 2.6e-9 against max ||.|| = 5e-4)."""
    torch.cuda.empty_cache()
    ds = T.synth.make_tripartite_device(1_000_000, 1_000_000, 2_000_000, 100_000_000, seed=2, device=DEV)
    cfg = T.get_config("tgcn", dim_latent=128, dim_layer_list=[128, 128, 128], device=DEV, train_batch=512, neighbor_k=25, reg=1e-4)
    torch.manual_seed(cfg["seed"])
    m = T.TGCN(ds, config=cfg)
    m.train()
    assert m.prune_forward and m.step_node and m._step_node_ok()
    batch = T.BPR_training_data(ds, config=cfg, seed=2020).all_train_data[:512]
    need = m._needed_rows(batch)
    assert need[1]["user"] is not None and 100_000 < need[1]["user"].numel() < 600_000      # the restricted step IS restricted
    res = []
    for restricted in (True, False):
        m.prune_forward = restricted
        m.zero_grad(set_to_none=True)
        lossx = m.loss(batch)
        sum(lossx).backward()
        res.append(([float(v.detach()) for v in lossx], {k: p.grad for k, p in m.named_parameters() if p.grad is not None}))
        for p in m.parameters():
            p.grad = None
        torch.cuda.synchronize()
    (l1, g1), (l0, g0) = res
    np.testing.assert_allclose(l1, l0, rtol=2e-6)
    assert set(g0) == set(g1) and len(g0) == 4 + 3 * 21
    norms = {k: float(g0[k].double().norm()) for k in g0}
    top = max(norms.values())
    bad = {}
    for k in g0:
        err = float((g0[k].double() - g1[k].double()).norm())
        if not err <= 2e-3 * norms[k] + 1e-5 * top:
            bad[k] = (err, norms[k])
    assert not bad, (top, bad)
    del m, ds, res, g0, g1
    torch.cuda.empty_cache()
