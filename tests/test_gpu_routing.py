"""GPU parity of the routed-propagation kernels (csrc/routing.hip) and of DGCF / DisenGCN against
torch restatements of the same arithmetic and against fixtures captured from the reference
(tests/golden/dgcf_*.npz, disengcn_toy.npz; oracle/make_golden.py `siblings`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import blocks_from_fixture

import tagrec_amd as T
from tagrec_amd import routing as R
from tagrec_amd.synth import Coo
from oracle import adj as oadj
from oracle import models as om

DEV = torch.device("cuda:0")


def _random_symmetric_graph(n, avg_deg, seed, hub=0):
    """Symmetric 0/1 structure; `hub` > 0 adds one node adjacent to `hub` others (a long row, > 1024 entries)."""
    rng = np.random.RandomState(seed)
    m = n * avg_deg // 2
    a, b = rng.randint(0, n, m), rng.randint(0, n, m)
    if hub:
        others = rng.choice(np.arange(1, n), hub, replace=False)
        a, b = np.concatenate([a, np.zeros(hub, np.int64)]), np.concatenate([b, others])
    keep = a != b
    a, b = a[keep], b[keep]
    rows, cols = np.concatenate([a, b]), np.concatenate([b, a])
    key = np.unique(rows.astype(np.int64) * n + cols)
    rows, cols = key // n, key % n
    csr = oadj.coo_to_csr(rows, cols, np.ones(len(rows), np.float32), (n, n))
    g = T.Graph.from_host(csr.rowptr, csr.col, csr.val, csr.shape, DEV, symmetric=True)
    return g, torch.from_numpy(rows).to(DEV), torch.from_numpy(cols).to(DEV)


def _slices(x, K):
    return x.view(x.shape[0], K, -1)


@pytest.mark.parametrize("D,K", [(64, 4), (32, 2), (128, 8), (64, 1), (32, 8), (256, 4)])
@pytest.mark.parametrize("hub", [0, 1500])
def test_routing_kernels_match_torch(D, K, hub):
    n = 2000
    g, rows, cols = _random_symmetric_graph(n, 12, seed=D + K + hub, hub=hub)
    rg = R.RoutingGraph(g)
    nnz = rg.nnz
    assert torch.equal(rows, rg.rows) and torch.equal(cols, rg.cols)
    gen = torch.Generator(device="cpu").manual_seed(D * K)
    logits = torch.randn(nnz, K, generator=gen).to(DEV)
    x = torch.randn(n, D, generator=gen).to(DEV)
    other = torch.randn(n, D, generator=gen).to(DEV)
    # softmax over factors
    w = rg.softmax(logits)
    np.testing.assert_allclose(w.cpu().numpy(), torch.softmax(logits, 1).cpu().numpy(), rtol=1e-5, atol=1e-7)
    # row sums -> 1/sqrt (inf -> 0 on empty rows)
    d = rg.rowsum_rsqrt(w)
    rs = torch.zeros(n, K, device=DEV).index_add_(0, rows, w)
    want_d = torch.where(rs > 0, 1.0 / torch.sqrt(rs), torch.zeros_like(rs))
    np.testing.assert_allclose(d.cpu().numpy(), want_d.cpu().numpy(), rtol=2e-5, atol=1e-7)
    # transposition permutation: wt[(c, r)] == w[(r, c)]
    wt = rg.permute(w)
    dense = torch.zeros(n, n, K, device=DEV)
    dense[rows, cols] = w
    np.testing.assert_array_equal(wt.cpu().numpy(), dense.transpose(0, 1)[rows, cols].cpu().numpy())
    # routed product with every epilogue term
    y, yn, inv = rg.spmm(w, x, post=d, self_add=other, b=x, b_scale=0.25, raw=True, normed=True)
    agg = torch.zeros(n, K, D // K, device=DEV).index_add_(0, rows, w[:, :, None] * _slices(x, K)[cols])
    want = (d[:, :, None] * agg).reshape(n, D) + other + 0.25 * x
    scale = float(want.abs().max())
    np.testing.assert_allclose(y.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)
    den = _slices(want, K).norm(dim=2).clamp_min(1e-12)
    np.testing.assert_allclose(yn.cpu().numpy(), (_slices(want, K) / den[:, :, None]).reshape(n, D).cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(inv.cpu().numpy(), (1.0 / den).cpu().numpy(), rtol=1e-4)
    # plain product, normalised output only
    _, yn2, _ = rg.spmm(w, x, raw=False, normed=True)
    p = agg.reshape(n, D)
    den2 = _slices(p, K).norm(dim=2).clamp_min(1e-12)
    np.testing.assert_allclose(yn2.cpu().numpy(), (_slices(p, K) / den2[:, :, None]).reshape(n, D).cpu().numpy(), rtol=1e-4, atol=1e-5)
    # scores: write, then accumulate
    sc = torch.empty(nnz, K, device=DEV)
    rg.score(x, other, sc, accumulate=False)
    want_sc = (_slices(x, K)[rows] * _slices(other, K)[cols]).sum(2)
    np.testing.assert_allclose(sc.cpu().numpy(), want_sc.cpu().numpy(), rtol=1e-4, atol=1e-4)
    rg.score(x, other, sc, accumulate=True)
    np.testing.assert_allclose(sc.cpu().numpy(), 2 * want_sc.cpu().numpy(), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("D,K", [(64, 4), (32, 8), (128, 2)])
def test_slice_ops_match_torch(D, K):
    n = 777
    gen = torch.Generator(device="cpu").manual_seed(D + K)
    x = torch.randn(n, D, generator=gen)
    x[5] = 0.0                                                   # a zero slice row: the clamp branch
    x = x.to(DEV)
    s = torch.rand(n, K, generator=gen).to(DEV)
    np.testing.assert_allclose(R.slice_scale(x, s).cpu().numpy(), (_slices(x, K) * s[:, :, None]).reshape(n, D).cpu().numpy(), rtol=1e-6)
    y, inv = R.slice_norm_fwd(x, K)
    want = torch.nn.functional.normalize(_slices(x, K), p=2, dim=2).reshape(n, D)
    np.testing.assert_allclose(y.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-7)
    yt, _ = R.slice_norm_fwd(x, K, tanh=True, want_inv=False)
    np.testing.assert_allclose(yt.cpu().numpy(), torch.tanh(want).cpu().numpy(), rtol=1e-5, atol=1e-7)
    # backward against autograd
    xr = x.clone().requires_grad_()
    dz = torch.randn(n, D, generator=gen).to(DEV)
    torch.nn.functional.normalize(_slices(xr, K), p=2, dim=2).reshape(n, D).backward(dz)
    got = R.slice_norm_bwd(x, inv, dz)
    np.testing.assert_allclose(got.cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
    # the autograd wrapper
    xr2 = x.clone().requires_grad_()
    R.slice_normalize(xr2, K).backward(dz)
    np.testing.assert_allclose(xr2.grad.cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_routing_rejects_bad_arguments():
    g, _, _ = _random_symmetric_graph(100, 6, seed=1)
    rg = R.RoutingGraph(g)
    w = torch.ones(rg.nnz, 4, device=DEV)
    with pytest.raises(T.TagrecError, match="D in"):
        rg.spmm(w, torch.zeros(100, 48, device=DEV))             # 48 columns: no kernel
    with pytest.raises(T.TagrecError, match="K must be|K in"):
        rg.softmax(torch.ones(rg.nnz, 3, device=DEV))
    with pytest.raises(T.TagrecError, match="aliases"):
        x = torch.zeros(100, 64, device=DEV)
        _lib = T._lib
        _lib.check(_lib.load().tagrec_route_spmm_f32(g.handle, _lib.ptr(w), 4, _lib.ptr(x), None, None, None, 0.0, _lib.ptr(x),
                                                     None, None, 64, _lib.stream_ptr()), "route_spmm")


def test_edge_value_operators_on_a_non_symmetric_graph():
    """edge_score / row_softmax / valued_spmm (the differentiable pieces KGAT is built from) against torch autograd on a
    directed graph with duplicate entries and a long row; exercises the transposed structure + permutation."""
    n, D = 1500, 32
    rng = np.random.RandomState(2)
    rows = np.concatenate([rng.randint(0, n, 9000), np.zeros(1200, np.int64), [3, 3, 3]])
    cols = np.concatenate([rng.randint(0, n, 9000), rng.choice(n, 1200, replace=False), [7, 7, 9]])       # (3,7) twice
    rg, order = R.RoutingGraph.from_edges(rows, cols, n, DEV)
    assert not rg.symmetric and rg.nnz == len(rows)
    r, c = rg.rows, rg.cols
    assert torch.equal(r, torch.from_numpy(rows).to(DEV)[order]) and torch.equal(c, torch.from_numpy(cols).to(DEV)[order])
    gen = torch.Generator(device="cpu").manual_seed(0)
    Hh = torch.randn(n, D, generator=gen).to(DEV).requires_grad_()
    Tt = torch.randn(n, D, generator=gen).to(DEV).requires_grad_()
    X = torch.randn(n, D, generator=gen).to(DEV).requires_grad_()
    gy = torch.randn(n, D, generator=gen).to(DEV)

    def pipeline(score, softmax, spmm):
        a = softmax(0.3 * score(Hh, Tt))
        return a, spmm(a, X)

    def ref_softmax(s):
        mx = torch.full((n,), -1e30, device=DEV).scatter_reduce(0, r, s.detach(), "amax")
        e = torch.exp(s - mx[r])
        return e / torch.zeros(n, device=DEV).index_add(0, r, e)[r]

    a0, y0 = pipeline(lambda h, t: (h[r] * t[c]).sum(1), ref_softmax,
                      lambda a, x: torch.zeros(n, D, device=DEV).index_add(0, r, a[:, None] * x[c]))
    y0.backward(gy)
    want = [t.grad.clone() for t in (Hh, Tt, X)]
    for t in (Hh, Tt, X):
        t.grad = None
    a1, y1 = pipeline(lambda h, t: R.edge_score(h, t, rg), lambda s: R.row_softmax(s, rg), lambda a, x: R.valued_spmm(a, x, rg))
    np.testing.assert_allclose(a1.detach().cpu().numpy(), a0.detach().cpu().numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(y1.detach().cpu().numpy(), y0.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    y1.backward(gy)
    for t, w in zip((Hh, Tt, X), want):
        np.testing.assert_allclose(t.grad.cpu().numpy(), w.cpu().numpy(), rtol=1e-3, atol=1e-4 * float(w.abs().max()))


def test_routing_identities_at_scale():
    """Size-independent properties on a graph the CPU oracle cannot finish in seconds (2 M nodes, 40 M entries, long rows
    included): (1) the routed product with K = 1 and the graph's own values IS the plain product; (2) product and score pass
    are adjoint: <H, A(w) T> = sum_j w_j <H[row_j], T[col_j]>, per factor; (3) A(w)^T via the permutation satisfies
    <x, A(w) y> = <A(w)^T x, y>; (4) softmax weights sum to one per entry and the row scaling normalises row sums."""
    ds = T.synth.make_bipartite_device(1_000_000, 1_000_000, 20_000_000, seed=4, device=DEV)
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], 1_000_000, 1_000_000, "bi_norm")
    g = T.Graph(rp, col, val, (n, n), symmetric=True)
    rg = R.RoutingGraph(g)
    assert g.info()["n_long_rows"] > 0
    D, K = 64, 4
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(n, D, device=DEV, generator=gen)
    h = torch.randn(n, D, device=DEV, generator=gen)
    # (1)
    y_plain = g.spmm(x)
    y_routed, _, _ = rg.spmm(val.view(-1, 1), x)
    assert float((y_routed - y_plain).abs().max()) <= 1e-5 * float(y_plain.abs().max())
    # (2), (3), (4)
    w = rg.softmax(torch.randn(rg.nnz, K, device=DEV, generator=gen))
    assert float((w.sum(1) - 1).abs().max()) < 1e-5
    y, _, _ = rg.spmm(w, x)
    sc = torch.empty(rg.nnz, K, device=DEV)
    rg.score(h, x, sc, accumulate=False)
    lhs = (h.double() * y.double()).view(n, K, -1).sum((0, 2))
    rhs = (w.double() * sc.double()).sum(0)
    # the sums cancel heavily (random signs): tolerance from the size of the terms, not of the total
    mag = float((h.double() * y.double()).abs().sum())
    np.testing.assert_allclose(lhs.cpu().numpy(), rhs.cpu().numpy(), rtol=0, atol=1e-7 * mag)
    yt, _, _ = rg.spmm(w, h, transposed=True)
    np.testing.assert_allclose(float((x.double() * yt.double()).sum()), float((h.double() * y.double()).sum()), rtol=0, atol=1e-7 * mag)
    d = rg.rowsum_rsqrt(w)
    ones = torch.ones(n, D, device=DEV)
    rs, _, _ = rg.spmm(w, ones)                                      # row sums of every factor, broadcast over its slice
    nz = (rp[1:] > rp[:-1])
    np.testing.assert_allclose((d[nz] ** 2 * rs.view(n, K, -1)[nz][:, :, 0]).cpu().numpy(), 1.0, rtol=1e-4)


# ------------------------------------------------------------------ models against the reference's fixtures
def _ds_from_fixture(fx):
    ds = T.synth.Dataset()
    use_tag = int(fx["use_tag"])
    ui, ut, it = blocks_from_fixture(fx, use_tag)
    ds.num = {"user": int(fx["n_user"]), "item": int(fx["n_item"]), "tag": int(fx["n_tag"])}
    ds.ui_adj = Coo(*ui)
    if use_tag:
        ds.ut_adj, ds.it_adj = Coo(*ut), Coo(*it)
    return ds


def _sibling(fx, name):
    cfg = T.get_config(name, use_tag=bool(int(fx["use_tag"])), dim_layer_list=[int(fx["D"])] * int(fx["n_layer"]),
                       dim_latent=int(fx["D"]), reg=float(fx["reg"]), factor_k=int(fx["factor_k"]),
                       iterate_k=int(fx["iterate_k"]), device=DEV)
    m = {"dgcf": T.DGCF, "disengcn": T.DisenGCN}[name](_ds_from_fixture(fx), config=cfg)
    m.load_state_dict({k[5:]: torch.from_numpy(fx[k]) for k in fx if k.startswith("init.")})
    return m


def _grad_check(got, want, rtol=1e-3):
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, rtol=rtol, atol=2e-5 * scale)


@pytest.mark.parametrize("name", ["dgcf_toy", "dgcf_med"])
def test_dgcf_golden(golden, name):
    fx = golden(name)
    m = _sibling(fx, "dgcf")
    m.train()
    rg = m.routing
    np.testing.assert_array_equal(torch.stack([rg.rows, rg.cols]).cpu().numpy(), fx["adj_idx"])      # same entry order
    with torch.no_grad():
        for t, o in enumerate(m.forward()):
            np.testing.assert_allclose(o.cpu().numpy(), fx[f"out.{t}"], rtol=1e-4, atol=1e-6)
        layer_a = m.forward(out_A=True)
    got_a = np.stack([np.stack([a._values().cpu().numpy() for a in la]) for la in layer_a])
    np.testing.assert_allclose(got_a, fx["out_A"], rtol=1e-4, atol=1e-6)
    cor = torch.zeros(2, 4, dtype=torch.long)
    lossx = m.loss((torch.from_numpy(fx["batches"][0]).to(DEV), cor))
    np.testing.assert_allclose([float(v) for v in lossx], fx["loss_parts"], rtol=1e-5, atol=1e-8)
    sum(lossx).backward()
    want = np.concatenate([fx[f"grad.embed.{t}"] for t in range(len(m.num_list))])
    _grad_check(m.table.grad.cpu().numpy(), want)
    # the step surface: 1 and 3 Adam steps through epoch_training
    for n_steps in (1, 3):
        m = _sibling(fx, "dgcf")
        m.train()
        opt = T.Adam(m.parameters(), lr=float(fx["lr"]))
        prod = T.Fixed_training_data([np.concatenate(fx["batches"][:n_steps])], fx["batches"].shape[1], DEV)
        prod.mini_batch = lambda: iter([(torch.from_numpy(b).to(DEV), cor) for b in fx["batches"][:n_steps]])
        losses = T.epoch_training(prod, m.loss, opt, verbose=False)
        np.testing.assert_allclose(losses, fx[f"step{n_steps}.losses"], rtol=5e-5)
        sd = m.state_dict()
        for t in range(len(m.num_list)):
            got, want = sd[f"embed.{t}"].cpu().numpy(), fx[f"step{n_steps}.embed.{t}"]
            assert np.abs(got - want).max() <= 3e-4
            assert np.mean(np.abs(got - want) <= 2e-5) >= 0.99
    m.eval()
    users = torch.from_numpy(fx["predict.users"]).to(DEV)
    m.load_state_dict({k[6:]: torch.from_numpy(fx[k]) for k in fx if k.startswith("step3.embed.")})
    np.testing.assert_allclose(m.predict_rating(users).cpu().numpy(), fx["predict.rating"], rtol=1e-4, atol=1e-5)


def test_dgcf_matches_oracle_on_a_larger_graph():
    """Seeded random graph beyond the fixtures (long row included): HIP DGCF vs the CPU oracle, forward + gradient."""
    n_u, n_i = 900, 700
    rng = np.random.RandomState(5)
    u = np.concatenate([rng.randint(0, n_u, 9000), np.arange(n_u)])
    i = np.concatenate([rng.randint(0, n_i, 9000), np.zeros(n_u, np.int64)])            # item 0 meets every user: 900+ entries
    key = np.unique(u.astype(np.int64) * n_i + i)
    u, i = key // n_i, key % n_i
    ds = T.synth.Dataset()
    ds.num = {"user": n_u, "item": n_i, "tag": 0}
    ds.ui_adj = Coo(u, i, np.ones(len(u), np.float32), (n_u, n_i))
    cfg = T.get_config("dgcf", use_tag=False, dim_layer_list=[64, 64], dim_latent=64, reg=1e-3, factor_k=4, iterate_k=2, device=DEV)
    torch.manual_seed(3)
    m = T.DGCF(ds, config=cfg)
    m.train()
    rg = m.routing
    tabs = [t.detach().cpu().clone().requires_grad_() for t in m.embed]
    rows, cols = rg.rows.cpu(), rg.cols.cpu()
    batch = torch.from_numpy(np.stack([rng.randint(0, n_u, 256), rng.randint(0, n_i, 256), rng.randint(0, n_i, 256)], 1))
    lo, lr_ = om.dgcf_loss(tabs, rows, cols, 2, 4, 2, batch, 1e-3, "softplus")
    (lo + lr_).backward()
    lossx = m.loss(batch.to(DEV))
    np.testing.assert_allclose([float(v) for v in lossx], [float(lo), float(lr_)], rtol=2e-5)
    sum(lossx).backward()
    _grad_check(m.table.grad.cpu().numpy(), np.concatenate([t.grad.numpy() for t in tabs]))


def test_disengcn_golden(golden):
    fx = golden("disengcn_toy")
    m = _sibling(fx, "disengcn")
    m.train()
    assert list(m.state_dict().keys()) == [k[5:] for k in fx if k.startswith("init.")]
    with torch.no_grad():
        for t, o in enumerate(m.forward()):
            np.testing.assert_allclose(o.cpu().numpy(), fx[f"out.{t}"], rtol=1e-4, atol=1e-6)
    cor = torch.zeros(2, 4, dtype=torch.long)
    lossx = m.loss((torch.from_numpy(fx["batches"][0]).to(DEV), cor))
    np.testing.assert_allclose([float(v) for v in lossx], fx["loss_parts"], rtol=1e-5, atol=1e-8)
    sum(lossx).backward()
    _grad_check(m.table.grad.cpu().numpy(), np.concatenate([fx[f"grad.embed.{t}"] for t in range(3)]))
    for k in range(int(fx["n_layer"])):
        _grad_check(m.layer[k].W.grad.cpu().numpy(), fx[f"grad.layer.{k}.W"])
        _grad_check(m.layer[k].b.grad.cpu().numpy(), fx[f"grad.layer.{k}.b"])
    for n_steps in (1, 3):
        m = _sibling(fx, "disengcn")
        m.train()
        opt = T.Adam(m.parameters(), lr=float(fx["lr"]))
        prod = T.Fixed_training_data([np.concatenate(fx["batches"][:n_steps])], fx["batches"].shape[1], DEV)
        prod.mini_batch = lambda: iter([(torch.from_numpy(b).to(DEV), cor) for b in fx["batches"][:n_steps]])
        losses = T.epoch_training(prod, m.loss, opt, verbose=False)
        np.testing.assert_allclose(losses, fx[f"step{n_steps}.losses"], rtol=5e-5)
        sd = m.state_dict()
        for key in sd:
            got, want = sd[key].cpu().numpy(), fx[f"step{n_steps}.{key}"]
            # Adam turns a gradient g into lr * g / (|g| + 1e-8): entries whose gradient is ~1e-8 move by a visible
            # fraction of lr on a last-bit difference, so bound the bulk tightly and the outliers by lr / 10
            assert np.mean(np.abs(got - want) <= 2e-5) >= 0.99, key
            assert np.abs(got - want).max() <= 1e-3, key


def test_dgcf_training_data_producer():
    """`DGCF_training_data` (bpr_training_data.py:47-83): E // B + 1 batches of (triplets, cor); users distinct when
    there are more users than the batch; positives are train items of their user, negatives are not."""
    ds = T.synth.make_cf_dataset(300, 200, 4000, seed=12, n_tag=40, n_assign=600)
    cfg = T.get_config("dgcf", device=DEV, train_batch=128, use_tag=True)
    prod = T.DGCF_training_data(ds, config=cfg, seed=4)
    prod.reset()
    batches = list(prod.mini_batch())
    assert len(batches) == len(ds.edge_index["train"]) // 128 + 1
    train = ds.user_items["train"]
    for data, cor in batches[:5]:
        d = data.cpu().numpy()
        assert d.shape == (128, 3) and cor.shape == (3, 40)      # cor_batch 100 capped by the 40 tags
        assert len(set(d[:, 0].tolist())) == 128
        assert all(p in train[u] and n not in train[u] for u, p, n in d.tolist())
    model = T.DGCF(ds, config=T.get_config("dgcf", device=DEV, train_batch=128, use_tag=True, dim_layer_list=[64]))
    losses = T.epoch_training(prod, model.loss, T.Adam(model.parameters(), lr=0.01), verbose=False)
    assert len(losses) == len(batches) and np.isfinite(losses).all() and losses[-1] < losses[0]


# ------------------------------------------------------------------ KGAT
class _KgatData:
    def __init__(self, fx):
        self.num = {"user": int(fx["n_user"]), "item": int(fx["n_item"]), "tag": int(fx["n_tag"])}
        self._edges = {int(k[6:]): fx[k] for k in fx if k.startswith("edges.")}

    def create_edge(self):
        return {k: self._edges[k] for k in sorted(self._edges)}


def _kgat(fx):
    cfg = T.get_config("kgat", dim_layer_list=[int(x) for x in fx["layers"]], dim_latent=int(fx["D"]),
                       dim_relation=int(fx["dim_relation"]), reg=float(fx["reg"]), agg_type=str(fx["agg_type"]),
                       cor_reg=float(fx["cor_reg"]), device=DEV)
    m = T.KGAT(_KgatData(fx), config=cfg)
    assert list(m.state_dict().keys()) == [k[5:] for k in fx if k.startswith("init.")]
    m.load_state_dict({k[5:]: torch.from_numpy(fx[k]) for k in fx if k.startswith("init.")})
    return m


@pytest.mark.parametrize("name", ["kgat_toy", "kgat_toy_wired", "kgat_toy_default"])
def test_kgat_golden(golden, name):
    fx = golden(name)
    m = _kgat(fx)
    m.train()
    with torch.no_grad():
        users, ents = m.forward()
    np.testing.assert_allclose(users.cpu().numpy(), fx["out.0"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ents.cpu().numpy(), fx["out.1"], rtol=1e-4, atol=1e-6)
    lossx = m.loss(torch.from_numpy(fx["batches"][0]).to(DEV))
    np.testing.assert_allclose([float(v) for v in lossx], fx["loss_parts"], rtol=1e-5, atol=1e-8)
    sum(lossx).backward()
    for k, p in m.named_parameters():
        want = fx["grad." + k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
        if np.abs(want).max() == 0:
            assert np.abs(got).max() <= 1e-12, k
        else:
            _grad_check(got, want)
    # TransE phase
    m.zero_grad()
    lt = m.transe_loss(torch.from_numpy(fx["transe_batch"]).to(DEV))
    np.testing.assert_allclose([float(v) for v in lt], fx["transe_loss_parts"], rtol=1e-5, atol=1e-8)
    sum(lt).backward()
    for k, p in m.named_parameters():
        if "transe_grad." + k in fx:
            _grad_check(p.grad.cpu().numpy(), fx["transe_grad." + k])
    # step surface
    for n_steps in (1, 3):
        m = _kgat(fx)
        m.train()
        opt = T.Adam(m.parameters(), lr=float(fx["lr"]))
        prod = T.Fixed_training_data([np.concatenate(fx["batches"][:n_steps])], fx["batches"].shape[1], DEV)
        prod.mini_batch = lambda: iter([torch.from_numpy(b).to(DEV) for b in fx["batches"][:n_steps]])
        losses = T.epoch_training(prod, m.loss, opt, verbose=False)
        np.testing.assert_allclose(losses, fx[f"step{n_steps}.losses"], rtol=5e-5)
        sd = m.state_dict()
        for key in sd:
            got, want = sd[key].cpu().numpy(), fx[f"step{n_steps}.{key}"]
            assert np.mean(np.abs(got - want) <= 2e-5) >= 0.99, key
            assert np.abs(got - want).max() <= 1e-3, key
    m.eval()
    m.load_state_dict({k[6:]: torch.from_numpy(fx[k]) for k in fx if k.startswith("step3.") and k != "step3.losses"})
    got = m.predict_rating(torch.from_numpy(fx["predict.users"]).to(DEV))
    assert got.shape == (len(fx["predict.users"]), int(fx["n_item"]) + int(fx["n_tag"]))
    np.testing.assert_allclose(got.cpu().numpy(), fx["predict.rating"], rtol=1e-4, atol=1e-5)


def test_kgat_training_data_producer(golden):
    """`KGAT_training_data` (transe_training_data.py:12-41): windows shifted by ONE row, negatives never a known tail."""
    fx = golden("kgat_toy_wired")
    data = _KgatData(fx)
    cfg = T.get_config("kgat", device=DEV, transe_batch=32)
    prod = T.KGAT_training_data(data, config=cfg, seed=1)
    n_tri = sum(v.shape[1] for v in data.create_edge().values())
    assert prod.all_triplet.shape == (n_tri, 3) and prod.tot_inter == n_tri // 32
    batches = list(prod.mini_batch())
    assert len(batches) == prod.tot_inter
    b0, b1 = batches[0].cpu().numpy(), batches[1].cpu().numpy()
    assert b0.shape == (32, 4) and np.array_equal(b0[1:, :3], b1[:-1, :3])          # the stride-1 windows
    known = set(map(tuple, prod.all_triplet.cpu().numpy().tolist()))
    assert all((h, r, n) not in known for h, r, _, n in np.concatenate([b.cpu().numpy() for b in batches[:20]]).tolist())
    # relation ids follow create_edge's keys; heads / tails its rows
    e0 = data.create_edge()[0]
    assert np.array_equal(prod.all_triplet[:e0.shape[1]].cpu().numpy(), np.stack([e0[0], np.zeros_like(e0[0]), e0[1]], 1))


@pytest.mark.parametrize("name", ["dgcf", "disengcn"])
def test_restricted_top_layer_equals_full_step(name):
    """DGCF / DisenGCN `loss` with the top layer's routing restricted to the rows the loss depends on, and the backward
    products skipping zero gradient rows, vs everything on all rows: same loss parts, same gradients."""
    ds = T.synth.make_cf_dataset(9000, 7000, 150_000, seed=13, n_tag=2000, n_assign=60_000)
    cfg = T.get_config(name, use_tag=True, dim_layer_list=[64, 64], dim_latent=64, reg=1e-3, factor_k=4, iterate_k=2, device=DEV,
                       train_batch=128)
    torch.manual_seed(6)
    m = {"dgcf": T.DGCF, "disengcn": T.DisenGCN}[name](ds, config=cfg)
    m.train()
    batch = torch.from_numpy(T.synth.sample_bpr_epoch(ds, 1)[:128]).to(DEV)
    res = []
    for restrict in (False, True):
        m.restrict_forward = restrict
        m.zero_grad()
        lossx = m.loss(batch)
        sum(lossx).backward()
        res.append(([float(v) for v in lossx], {k: p.grad.clone() for k, p in m.named_parameters()}))
    (l0, g0), (l1, g1) = res
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    top = max(float(v.double().norm()) for v in g0.values())
    for k in g0:
        a, b = g0[k].double(), g1[k].double()
        assert float((a - b).norm()) <= 1e-3 * float(a.norm()) + 1e-6 * top, k
