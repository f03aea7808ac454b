"""The CPU oracle against the golden vectors captured from the reference itself
(oracle/make_golden.py).  Pins the oracle; runs without /root/reference."""
import numpy as np
import pytest
import torch

from conftest import blocks_from_fixture
from oracle import adj as oadj
from oracle import data as odata
from oracle import models as om

torch.set_num_threads(4)


def _coo(csr):
    return np.stack([csr.rows(), csr.col.astype(np.int64)]), csr.val


# ------------------------------------------------------------------ A1-A3
@pytest.mark.parametrize("use_tag", [0, 1])
@pytest.mark.parametrize("norm", ["bi_norm", "ngcf", "si_norm", "si_norm_self", "plain"])
def test_adjacency_matches_reference(golden, use_tag, norm):
    fx = golden("adj_toy")
    csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, use_tag)), norm)
    idx, val = _coo(csr)
    assert np.array_equal(idx, fx[f"{norm}_{use_tag}_idx"])
    # same fp32 operation order as scipy's diag.dot(adj).dot(diag) -> bit-exact
    assert np.array_equal(val, fx[f"{norm}_{use_tag}_val"])


def test_row_folds_match_split_sp_mat(golden):
    fx = golden("adj_toy")
    csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, 1)), "bi_norm")
    for k, (lo, hi) in enumerate(oadj.row_folds(csr.shape[0], 3)):
        sub = oadj.slice_rows(csr, lo, hi)
        idx, val = _coo(sub)
        assert tuple(fx[f"fold3_{k}_shape"]) == sub.shape
        assert np.array_equal(idx, fx[f"fold3_{k}_idx"])
        assert np.array_equal(val, fx[f"fold3_{k}_val"])


def test_csr_transpose_roundtrip(golden):
    fx = golden("adj_toy")
    csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, 1)), "ngcf")
    assert np.array_equal(csr.transpose().to_dense(), csr.to_dense().T)


# ------------------------------------------------------------------ LightGCN / NGCF
def _tables(fx, prefix="init."):
    keys = sorted(k for k in fx if k.startswith(prefix + "embed."))
    return [torch.from_numpy(fx[k].copy()) for k in keys]


def _mats(fx, prefix="init."):
    return {k[len(prefix) + 4:]: torch.from_numpy(fx[k].copy()) for k in fx if k.startswith(prefix + "mat.")}


def _adj(fx):
    csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, int(fx["use_tag"]))), str(fx["norm_type"]))
    return om.csr_to_torch(csr)


@pytest.mark.parametrize("name", ["lightgcn_toy", "lightgcn_med", "lightgcn_toy_d32", "lightgcn_toy_d256"])
def test_lightgcn_oracle(golden, name):
    fx = golden(name)
    A, L = _adj(fx), len(fx["layers"])
    tabs = [t.requires_grad_() for t in _tables(fx)]
    trace = []
    out = om.lightgcn_propagate(torch.cat(tabs), A, L, trace)
    parts = torch.split(out, [t.shape[0] for t in tabs])
    for t, o in enumerate(parts):
        np.testing.assert_allclose(o.detach().numpy(), fx[f"out.{t}"], rtol=1e-6, atol=1e-7)
    for k, (x, _) in enumerate(trace):
        np.testing.assert_allclose(x.detach().numpy(), fx[f"raw.{k}"], rtol=1e-6, atol=1e-7)
    batch = torch.from_numpy(fx["batches"][0])
    loss, reg = om.lightgcn_loss(tabs, A, L, batch, float(fx["reg"]), str(fx["loss_kind"]))
    np.testing.assert_allclose([float(loss), float(reg)], fx["loss_parts"], rtol=1e-6)
    (loss + reg).backward()
    for t, p in enumerate(tabs):
        np.testing.assert_allclose(p.grad.numpy(), fx[f"grad.embed.{t}"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("name", ["lightgcn_toy", "lightgcn_med"])
@pytest.mark.parametrize("n_steps", [1, 3])
def test_lightgcn_adam_steps(golden, name, n_steps):
    fx = golden(name)
    A, L = _adj(fx), len(fx["layers"])
    tabs = [t.requires_grad_() for t in _tables(fx)]
    opt = torch.optim.Adam(tabs, lr=float(fx["lr"]))
    fn = lambda b: om.lightgcn_loss(tabs, A, L, b, float(fx["reg"]), str(fx["loss_kind"]))
    totals, _ = om.adam_epoch(tabs, fn, [torch.from_numpy(b) for b in fx["batches"][:n_steps]], opt)
    np.testing.assert_allclose(totals, fx[f"step{n_steps}.losses"], rtol=1e-6)
    for t, p in enumerate(tabs):
        # Adam turns a gradient g into lr*g/(|g|+1e-8): where |g| ~ 1e-8 a last-bit change of g
        # (thread count changes torch's reduction order) moves the update by a visible fraction of lr
        np.testing.assert_allclose(p.detach().numpy(), fx[f"step{n_steps}.embed.{t}"], rtol=1e-5, atol=5e-5)


@pytest.mark.parametrize("name", ["ngcf_toy", "ngcf_med"])
def test_ngcf_oracle(golden, name):
    fx = golden(name)
    A, L = _adj(fx), len(fx["layers"])
    tabs = [t.requires_grad_() for t in _tables(fx)]
    mats = {k: v.requires_grad_() for k, v in _mats(fx).items()}
    out = om.ngcf_propagate(torch.cat(tabs), mats, A, L)
    parts = torch.split(out, [t.shape[0] for t in tabs])
    for t, o in enumerate(parts):
        np.testing.assert_allclose(o.detach().numpy(), fx[f"out.{t}"], rtol=1e-5, atol=1e-7)
    batch = torch.from_numpy(fx["batches"][0])
    loss, reg = om.ngcf_loss(tabs, mats, A, L, batch, float(fx["reg"]), str(fx["loss_kind"]))
    np.testing.assert_allclose([float(loss), float(reg)], fx["loss_parts"], rtol=1e-6)
    (loss + reg).backward()
    for t, p in enumerate(tabs):
        np.testing.assert_allclose(p.grad.numpy(), fx[f"grad.embed.{t}"], rtol=1e-4, atol=1e-9)
    for k, p in mats.items():
        np.testing.assert_allclose(p.grad.numpy(), fx[f"grad.mat.{k}"], rtol=1e-4, atol=1e-8)
    # 3 Adam steps
    tabs = [t.requires_grad_() for t in _tables(fx)]
    mats = {k: v.requires_grad_() for k, v in _mats(fx).items()}
    prm = tabs + [mats[k] for k in sorted(mats)]
    opt = torch.optim.Adam(prm, lr=float(fx["lr"]))
    fn = lambda b: om.ngcf_loss(tabs, mats, A, L, b, float(fx["reg"]), str(fx["loss_kind"]))
    totals, _ = om.adam_epoch(prm, fn, [torch.from_numpy(b) for b in fx["batches"][:3]], opt)
    np.testing.assert_allclose(totals, fx["step3.losses"], rtol=1e-5)
    for t, p in enumerate(tabs):
        np.testing.assert_allclose(p.detach().numpy(), fx[f"step3.embed.{t}"], rtol=1e-4, atol=5e-5)
    for k, p in mats.items():
        np.testing.assert_allclose(p.detach().numpy(), fx[f"step3.mat.{k}"], rtol=1e-4, atol=5e-5)


# ------------------------------------------------------------------ N4 siblings: DGCF, DisenGCN
def _edge_index(fx):
    """`norm_adj._indices()` of the "plain" adjacency, rebuilt from the interaction blocks: row-major order."""
    csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, int(fx["use_tag"]))), "plain")
    rows, cols = csr.rows(), csr.col.astype(np.int64)
    assert np.array_equal(np.stack([rows, cols]), fx["adj_idx"])          # same entry order as the reference's tensor
    return torch.from_numpy(rows), torch.from_numpy(cols)


@pytest.mark.parametrize("name", ["dgcf_toy", "dgcf_med"])
def test_dgcf_oracle(golden, name):
    fx = golden(name)
    rows, cols = _edge_index(fx)
    L, K, T = int(fx["n_layer"]), int(fx["factor_k"]), int(fx["iterate_k"])
    tabs = [t.requires_grad_() for t in _tables(fx)]
    trace = []
    outs = om.dgcf_forward(tabs, rows, cols, L, K, T, trace)
    for t, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), fx[f"out.{t}"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(torch.stack(trace).numpy(), fx["out_A"], rtol=1e-5, atol=1e-7)
    batch = torch.from_numpy(fx["batches"][0])
    loss, reg = om.dgcf_loss(tabs, rows, cols, L, K, T, batch, float(fx["reg"]), str(fx["loss_kind"]))
    np.testing.assert_allclose([float(loss), float(reg)], fx["loss_parts"], rtol=1e-6)
    (loss + reg).backward()
    for t, p in enumerate(tabs):
        np.testing.assert_allclose(p.grad.numpy(), fx[f"grad.embed.{t}"], rtol=1e-5, atol=1e-9)
    # three Adam steps
    tabs = [t.requires_grad_() for t in _tables(fx)]
    opt = torch.optim.Adam(tabs, lr=float(fx["lr"]))
    fn = lambda b: om.dgcf_loss(tabs, rows, cols, L, K, T, b, float(fx["reg"]), str(fx["loss_kind"]))
    totals, _ = om.adam_epoch(tabs, fn, [torch.from_numpy(b) for b in fx["batches"][:3]], opt)
    np.testing.assert_allclose(totals, fx["step3.losses"], rtol=1e-5)
    for t, p in enumerate(tabs):
        # Adam amplifies last-bit differences of near-zero gradients (see test_lightgcn_adam_steps)
        np.testing.assert_allclose(p.detach().numpy(), fx[f"step3.embed.{t}"], rtol=1e-5, atol=2e-4)


def _disen_layers(fx, prefix="init."):
    n = len([k for k in fx if k.startswith(prefix + "layer.") and k.endswith(".W")])
    return [(torch.from_numpy(fx[f"{prefix}layer.{k}.W"].copy()), torch.from_numpy(fx[f"{prefix}layer.{k}.b"].copy()))
            for k in range(n)]


def test_disengcn_oracle(golden):
    fx = golden("disengcn_toy")
    rows, cols = _edge_index(fx)
    K, T = int(fx["factor_k"]), int(fx["iterate_k"])
    tabs = [t.requires_grad_() for t in _tables(fx)]
    layers = [(W.requires_grad_(), b.requires_grad_()) for W, b in _disen_layers(fx)]
    outs = om.disengcn_forward(tabs, layers, rows, cols, K, T)
    for t, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), fx[f"out.{t}"], rtol=1e-5, atol=1e-7)
    batch = torch.from_numpy(fx["batches"][0])
    loss, reg = om.disengcn_loss(tabs, layers, rows, cols, K, T, batch, float(fx["reg"]), str(fx["loss_kind"]))
    np.testing.assert_allclose([float(loss), float(reg)], fx["loss_parts"], rtol=1e-6)
    (loss + reg).backward()
    for t, p in enumerate(tabs):
        np.testing.assert_allclose(p.grad.numpy(), fx[f"grad.embed.{t}"], rtol=1e-4, atol=1e-9)
    for k, (W, b) in enumerate(layers):
        np.testing.assert_allclose(W.grad.numpy(), fx[f"grad.layer.{k}.W"], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(b.grad.numpy(), fx[f"grad.layer.{k}.b"], rtol=1e-4, atol=1e-9)


def kgat_edges(fx):
    """The relation -> edge-array dict the reference model was given (saved verbatim in the fixture)."""
    ks = sorted(int(k[6:]) for k in fx if k.startswith("edges."))
    return {k: torch.from_numpy(fx[f"edges.{k}"].astype(np.int64)) for k in ks}


def _kgat_params(fx, prefix="init."):
    return {k[len(prefix):]: torch.from_numpy(fx[k].copy()) for k in fx if k.startswith(prefix)}


@pytest.mark.parametrize("name", ["kgat_toy", "kgat_toy_wired", "kgat_toy_default"])
def test_kgat_oracle(golden, name):
    fx = golden(name)
    edges, nu, L, agg = kgat_edges(fx), int(fx["n_user"]), len(fx["layers"]), str(fx["agg_type"])
    prm = {k: v.requires_grad_() for k, v in _kgat_params(fx).items()}
    users, ents = om.kgat_forward(prm, edges, nu, L, agg)
    np.testing.assert_allclose(users.detach().numpy(), fx["out.0"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ents.detach().numpy(), fx["out.1"], rtol=1e-5, atol=1e-7)
    batch = torch.from_numpy(fx["batches"][0])
    loss, reg = om.kgat_loss(prm, edges, nu, L, batch, float(fx["reg"]), agg)
    np.testing.assert_allclose([float(loss), float(reg)], fx["loss_parts"], rtol=1e-6)
    (loss + reg).backward()
    for k, p in prm.items():
        want = fx["grad." + k]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(want)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-9 + 1e-6 * np.abs(want).max(), err_msg=k)
    prm = {k: v.requires_grad_() for k, v in _kgat_params(fx).items()}
    lt, rt = om.kgat_transe_loss(prm, torch.from_numpy(fx["transe_batch"]), float(fx["cor_reg"]))
    np.testing.assert_allclose([float(lt), float(rt)], fx["transe_loss_parts"], rtol=1e-6)
    (lt + rt).backward()
    for k, p in prm.items():
        if "transe_grad." + k in fx:
            np.testing.assert_allclose(p.grad.numpy(), fx["transe_grad." + k], rtol=1e-4, atol=1e-9, err_msg=k)


def test_predict_rating(golden):
    fx = golden("lightgcn_toy")
    A, L = _adj(fx), len(fx["layers"])
    tabs = _tables(fx, "step3.")        # the fixture's ratings were taken after the 3-step run
    out = om.lightgcn_propagate(torch.cat(tabs), A, L)
    nu, ni = tabs[0].shape[0], tabs[1].shape[0]
    r = om.predict_rating(out[:nu], out[nu:nu + ni], torch.from_numpy(fx["predict.users"]))
    np.testing.assert_allclose(r.numpy(), fx["predict.rating"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ TGCN
def _tgcn_params(fx):
    return {k[5:]: torch.from_numpy(fx[k].copy()) for k in fx if k.startswith("init.")}


def _tgcn_nbr(fx):
    return [(torch.from_numpy(fx[f"nbr{r}.ids"]), torch.from_numpy(fx[f"nbr{r}.wts"])) for r in range(6)]


def test_tgcn_oracle(golden):
    fx = golden("tgcn_toy")
    prm = {k: v.requires_grad_() for k, v in _tgcn_params(fx).items()}
    nbr, L = _tgcn_nbr(fx), len(fx["layers"])
    outs = om.tgcn_forward(prm, L, nbr)
    for t, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), fx[f"out.{t}"], rtol=2e-5, atol=1e-6)
    batch = torch.from_numpy(fx["batches"][0])
    loss, reg = om.tgcn_loss(prm, L, nbr, batch, float(fx["reg"]))
    np.testing.assert_allclose([float(loss), float(reg)], fx["loss_parts"], rtol=1e-5)
    (loss + reg).backward()
    for k, p in prm.items():
        g = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
        np.testing.assert_allclose(g, fx["grad." + k], rtol=2e-3, atol=2e-7, err_msg=k)


def test_tgcn_transtag_oracle(golden):
    fx = golden("tgcn_toy")
    prm = {k: v.requires_grad_() for k, v in _tgcn_params(fx).items()}
    tt = torch.from_numpy(fx["tt_batch"])
    loss, reg = om.tgcn_transtag_loss(prm, tt, float(fx["margin"]), float(fx["transtag_reg"]))
    np.testing.assert_allclose([float(loss), float(reg)], fx["tt_loss_parts"], rtol=1e-6)
    (loss + reg).backward()
    for k in ("embed.user", "embed.item", "embed.tag"):
        np.testing.assert_allclose(prm[k].grad.numpy(), fx["tt_grad." + k], rtol=1e-5, atol=1e-9)


# ------------------------------------------------------------------ producer / metrics
def test_mini_batch_bounds(golden):
    fx = golden("producer")
    for key in (k for k in fx if k.startswith("mb_")):
        _, n, B = key.split("_")
        assert odata.mini_batch_bounds(int(n), int(B)) == [tuple(r) for r in fx[key].tolist()], key


def test_sample_neg_item_stream(golden):
    fx = golden("producer")
    ui = {int(k): v.tolist() for k, v in golden("producer_user_items").items()}
    rng = np.random.RandomState(int(fx["neg_seed"]))
    got = odata.sample_neg_item(fx["neg_pos"], ui, 30, rng)
    assert np.array_equal(got, fx["neg_out"])
    assert [len(c) for c in odata.split_data(np.arange(103), 5)] == fx["split5"].tolist()


def test_rank_metrics(golden):
    fx = golden("metrics")
    nu = fx["rating"].shape[0]
    train = {u: fx[f"train.{u}"].tolist() for u in range(nu)}
    test = {u: fx[f"test.{u}"].tolist() for u in range(nu)}
    got = odata.rank_metrics(fx["rating"], train, test, list(range(nu)), fx["topks"].tolist())
    for k in ("recall", "precision", "hr", "ndcg"):
        np.testing.assert_allclose(got[k], fx["res." + k], rtol=1e-6, err_msg=k)
