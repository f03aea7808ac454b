"""GPU: the training-driver surface end to end -- Basic_train / Early_stop / Basic_test wiring, the
two-phase TGCN schedule with one shared Adam (com.py:65-74), node / message dropout paths, the TransTag producer."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tagrec_amd as T
from tagrec_amd import help as H

DEV = torch.device("cuda:0")


def test_basic_train_run_lightgcn(tmp_path):
    """lightgcn_comp (com.py:21-29) re-created with tagrec_amd: loss decreases, eval runs every
    test_interval epochs, the best state_dict is saved with the reference's key names."""
    ds = T.synth.make_cf_dataset(120, 90, 2500, seed=8)
    cfg = T.get_config("lightgcn", use_tag=False, dim_layer_list=[32, 32], dim_latent=32, device=DEV, epochs=4,
                       test_interval=2, train_batch=256, test_batch=50)
    torch.manual_seed(1)
    model = T.LightGCN(ds, config=cfg)
    train = T.Basic_train([T.BPR_training_data(ds, config=cfg, seed=3)], [model.loss],
                          [T.Adam(model.parameters(), lr=0.01)], T.Basic_test(ds, config=cfg),
                          types.SimpleNamespace(out_dir=str(tmp_path)), config=cfg)
    hist = train.run(model, verbose=False)
    assert len(hist) == 4
    first, last = np.mean(hist[0][2]), np.mean(hist[-1][2])
    assert last < first
    assert train.early_stop.best_result is not None and 0 <= train.early_stop.best_result["recall"][1] <= 1
    sd = torch.load(tmp_path / "model.pth.tar", weights_only=True)
    assert list(sd.keys()) == ["embed.0", "embed.1"] and sd["embed.0"].shape == (120, 32)
    # the saved state loads back into a fresh model and reproduces its predictions
    m2 = T.LightGCN(ds, config=cfg)
    m2.load_state_dict(sd)
    m2.eval()
    assert m2.predict_rating(torch.arange(5, device=DEV)).shape == (5, 90)


def test_tgcn_two_phase_epoch_shared_adam():
    """tgcn_comp: phase 0 = BPR through the propagation, phase 1 = TransTag on the ego tables, ONE Adam
    instance for both (its step counter advances in both phases)."""
    ds = T.synth.make_cf_dataset(60, 50, 700, seed=9, n_tag=20, n_assign=500)
    cfg = T.get_config("tgcn", dim_layer_list=[16], dim_latent=16, device=DEV, neighbor_k=5, train_batch=128,
                       transtag_batch=128, epochs=2, test_interval=10)
    torch.manual_seed(2)
    model = T.TGCN(ds, config=cfg)
    opt = T.Adam(model.parameters(), lr=0.01)
    bpr, tt = T.BPR_training_data(ds, config=cfg, seed=1), T.TransTag_training_data(ds, config=cfg, seed=1)
    # producer properties (transe_training_data.py:42-70): rows (u, t, i+, i-), i- never a tagged item of (u, t)
    a = tt.all_train_data.cpu().numpy()
    uit = np.asarray(ds.uit_data)
    assert a.shape == (len(uit), 4)
    have = set(map(tuple, uit[:, [0, 2, 1]].tolist()))
    assert all((int(u), int(t), int(n)) not in have for u, t, _, n in a)
    assert np.array_equal(a[:, :3], uit[:, [0, 2, 1]])                      # not shuffled
    train = T.Basic_train([bpr, tt], [model.loss, model.transtag_loss], [opt, opt], None, None, config=cfg)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    hist = train.run(model, verbose=False)
    assert [h[1] for h in hist] == [0, 1, 0, 1]
    n_steps = sum(len(h[2]) for h in hist)
    assert opt.step_count == n_steps and all(np.isfinite(h[2]).all() for h in hist)
    moved = [k for k, v in model.named_parameters() if not torch.equal(v, before[k])]
    assert "embed.tag" in moved and "layer.0.Wf" in moved


def test_node_drop_semantics():
    """`node_drop` (adj.py:170-191): identity at rate 0 or in eval; otherwise kept edges are divided by the keep
    rate and about that fraction survives."""
    ds = T.synth.make_cf_dataset(200, 150, 6000, seed=3)
    g = T.creat_adj(ds, False, "bi_norm", 1, DEV)
    assert H.node_drop(g, 0.0, True) is g and H.node_drop(g, 0.3, False) is g
    torch.manual_seed(0)
    d = H.node_drop(g, 0.25, True)
    frac = d.nnz / g.nnz
    assert 0.70 < frac < 0.80
    # surviving values = original / 0.75
    dense_g = torch.zeros(g.shape, device=DEV)
    rows = torch.repeat_interleave(torch.arange(g.shape[0], device=DEV), g.rowptr[1:] - g.rowptr[:-1])
    dense_g[rows, g.col.long()] = g.val
    rows_d = torch.repeat_interleave(torch.arange(d.shape[0], device=DEV), d.rowptr[1:] - d.rowptr[:-1])
    np.testing.assert_allclose(d.val.cpu().numpy(), (dense_g[rows_d, d.col.long()] / 0.75).cpu().numpy(), rtol=1e-6)
    with pytest.raises(AssertionError):
        H.node_drop(g, 1.0, True)


def test_dropout_paths_run_and_eval_is_deterministic():
    """message / node dropout take the operator-by-operator path in training; eval ignores them."""
    ds = T.synth.make_cf_dataset(80, 60, 1200, seed=4)
    cfg = T.get_config("lightgcn", use_tag=False, dim_layer_list=[32, 32], dim_latent=32, device=DEV,
                       message_drop_list=[0.2, 0.2], node_drop=0.1)
    torch.manual_seed(0)
    m = T.LightGCN(ds, config=cfg)
    b = torch.from_numpy(T.synth.sample_bpr_epoch(ds, 0)[:64]).to(DEV)
    m.train()
    l1 = m.loss(b)
    sum(l1).backward()
    assert torch.isfinite(m.table.grad).all() and float(m.table.grad.abs().sum()) > 0
    m.eval()
    u1, i1 = m.forward()
    u2, i2 = m.forward()
    assert torch.equal(u1, u2) and torch.equal(i1, i2)
    ref = T.LightGCN(ds, config=T.get_config("lightgcn", use_tag=False, dim_layer_list=[32, 32], dim_latent=32, device=DEV))
    ref.load_state_dict(m.state_dict())
    ref.eval()
    np.testing.assert_allclose(ref.forward()[0].detach().cpu().numpy(), u1.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_predict_rating_cache_is_dropped_on_train():
    ds = T.synth.make_cf_dataset(50, 40, 500, seed=6)
    cfg = T.get_config("lightgcn", use_tag=False, dim_layer_list=[16], dim_latent=16, device=DEV)
    m = T.LightGCN(ds, config=cfg)
    users = torch.arange(10, device=DEV)
    m.eval()
    r1 = m.predict_rating(users)
    assert m._eval_cache is not None
    m.train()
    assert m._eval_cache is None
    opt = T.Adam(m.parameters(), lr=0.1)
    lossx = m.loss(torch.from_numpy(T.synth.sample_bpr_epoch(ds, 0)[:64]).to(DEV))
    sum(lossx).backward()
    opt.step()
    m.eval()
    r2 = m.predict_rating(users)
    assert not torch.equal(r1, r2)


def test_capturable_adam_matches_adam():
    """Adam(capturable=True) (step counter + bias corrections on the device) equals the host-stepped kernel."""
    torch.manual_seed(0)
    a = torch.randn(1000, 33, device=DEV).requires_grad_()
    b = a.detach().clone().requires_grad_()
    oa, ob = T.Adam([a], lr=0.01), T.Adam([b], lr=0.01, capturable=True)
    for k in range(5):
        g = torch.randn(1000, 33, device=DEV, generator=torch.Generator(device=DEV).manual_seed(k))
        a.grad, b.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=1e-6, atol=1e-8)
    assert int(ob.state[id(b)]["t_dev"]) == 5


@pytest.mark.parametrize("name", ["lightgcn", "ngcf", "tgcn"])
def test_graphed_epoch_matches_eager(name):
    """epoch_training(graphs={}) -- two eager steps, then every full-size batch a replay of ONE captured HIP graph --
    gives the losses and parameters of the eager loop (float atomics in the BPR scatter aside)."""
    ds = T.synth.make_cf_dataset(150, 120, 3000, seed=5, n_tag=30, n_assign=900)
    kw = dict(dim_layer_list=[32, 32], dim_latent=32, device=DEV, train_batch=256, neighbor_k=5)
    if name != "tgcn":
        kw["use_tag"] = False
    cfg = T.get_config(name, **kw)
    cls = {"lightgcn": T.LightGCN, "ngcf": T.NGCF, "tgcn": T.TGCN}[name]
    out = []
    for use_graph in (False, True):
        torch.manual_seed(3)
        m = cls(ds, config=cfg)
        m.train()
        opt = T.Adam(m.parameters(), lr=0.01, capturable=use_graph)
        prod = T.BPR_training_data(ds, config=cfg, seed=9)
        graphs = {} if use_graph else None
        losses = []
        for _ in range(2):
            losses += T.epoch_training(prod, m.loss, opt, verbose=False, graphs=graphs)
        if use_graph:
            assert not graphs.get("errors"), graphs.get("errors")
            assert sum(isinstance(v, T.GraphedStep) for v in graphs.values()) == 1
            assert opt.step_count == len(losses)
        out.append((np.array(losses), {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}))
    (l0, s0), (l1, s1) = out
    assert len(l0) == len(l1) and len(l0) >= 16
    # the first steps agree to rounding; later ones drift apart at the rate two EAGER runs do (float atomics in the
    # BPR scatter reorder sums, Adam and the ReLUs amplify it -- fastest for TGCN)
    np.testing.assert_allclose(l1[:8], l0[:8], rtol=2e-5)
    np.testing.assert_allclose(l1, l0, rtol=2e-4 if name != "tgcn" else 1e-2)
    if name != "tgcn":
        for k in s0:
            assert np.mean(np.abs(s1[k] - s0[k]) <= 2e-4) >= 0.95, k
            assert np.abs(s1[k] - s0[k]).max() <= 2e-2, k


def test_basic_train_with_hip_graph_config(tmp_path):
    ds = T.synth.make_cf_dataset(120, 90, 2500, seed=8)
    cfg = T.get_config("lightgcn", use_tag=False, dim_layer_list=[32, 32], dim_latent=32, device=DEV, epochs=3,
                       test_interval=2, train_batch=256, test_batch=50, hip_graph=True)
    torch.manual_seed(1)
    model = T.LightGCN(ds, config=cfg)
    train = T.Basic_train([T.BPR_training_data(ds, config=cfg, seed=3)], [model.loss],
                          [T.Adam(model.parameters(), lr=0.01, capturable=True)], T.Basic_test(ds, config=cfg),
                          types.SimpleNamespace(out_dir=str(tmp_path)), config=cfg)
    hist = train.run(model, verbose=False)
    assert not train.graphs.get("errors"), train.graphs.get("errors")
    assert any(isinstance(v, T.GraphedStep) for v in train.graphs.values())
    assert np.mean(hist[-1][2]) < np.mean(hist[0][2])


def test_fused_message_dropout_matches_masked_operator_form():
    """LightGCN with message dropout runs the fused layer kernels (mask drawn in the epilogue from (seed, layer, element));
    the same forward pass assembled from the unfused operators with the library's mask applied explicitly must give the
    same output and the same gradient.  The keep fraction is 1 - p."""
    from tagrec_amd import lightgcn as LG
    ds = T.synth.make_cf_dataset(300, 250, 6000, seed=4)
    p = [0.3, 0.5, 0.0]
    cfg = T.get_config("lightgcn", use_tag=False, dim_layer_list=[64, 64, 64], dim_latent=64, device=DEV, message_drop_list=p)
    torch.manual_seed(0)
    m = T.LightGCN(ds, config=cfg)
    m.train()
    g = m.norm_adj
    n, D = m.table.shape
    ones = torch.ones(n, D, device=DEV)
    keep = H.message_drop(ones, 0.3, 12345)
    frac = float((keep > 0).float().mean())
    assert abs(frac - 0.7) < 0.01 and torch.allclose(keep[keep > 0], torch.tensor(1 / 0.7, device=DEV))
    assert torch.equal(keep, H.message_drop(ones, 0.3, 12345)) and not torch.equal(keep, H.message_drop(ones, 0.3, 12346))
    # fused
    out = m._propagate()
    seed = (int(m.drop_seed) << 24) + m._drop_calls
    w = torch.randn(n, D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    (out * w).sum().backward()
    got_out, got_grad = out.detach().clone(), m.table.grad.clone()
    # operator form with explicit masks
    m.table.grad = None
    x = m.table
    layers = [x]
    for k in range(3):
        x = H.split_mm(g, x)
        if p[k] > 0:
            x = x * H.message_drop(ones, p[k], LG._layer_seed(seed, k))
        layers.append(H.normalize_rows(x))
    ref = torch.mean(torch.stack(layers, dim=1), dim=1)
    (ref * w).sum().backward()
    np.testing.assert_allclose(got_out.cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    scale = float(m.table.grad.abs().max())
    np.testing.assert_allclose(got_grad.cpu().numpy(), m.table.grad.cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)
    # a second training-mode pass draws different masks; the loss path (BPR fused behind it) is finite and differentiable
    out2 = m._propagate()
    assert not torch.allclose(out2, got_out)
    b = torch.from_numpy(T.synth.sample_bpr_epoch(ds, 0)[:128]).to(DEV)
    m.table.grad = None
    sum(m.loss(b)).backward()
    assert torch.isfinite(m.table.grad).all() and float(m.table.grad.abs().sum()) > 0


def test_ngcf_message_dropout_keeps_the_fused_kernels_and_matches_masked_operator_form():
    """NGCF with message dropout (ngcf.py:85) stays on the MFMA dense kernels: the mask is the library's counter-based one,
    applied between the dense kernel and the re-normalisation.  The same pass assembled from the unfused operators with
    that mask applied explicitly must give the same output and the same gradients (table and W / b)."""
    from tagrec_amd import ngcf as NG
    ds = T.synth.make_cf_dataset(300, 250, 6000, seed=4)
    p = [0.3, 0.0, 0.5]
    cfg = T.get_config("ngcf", use_tag=False, dim_layer_list=[64, 32, 64], dim_latent=64, device=DEV, message_drop_list=p)
    torch.manual_seed(0)
    m = T.NGCF(ds, config=cfg)
    m.train()
    assert m._fused_ok()
    n = m.table.shape[0]
    out = m._propagate()
    seed = (int(m.drop_seed) << 24) + m._drop_calls
    w = torch.randn(out.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    (out * w).sum().backward()
    got_out = out.detach().clone()
    got = {k: v.grad.clone() for k, v in m.named_parameters()}
    m.zero_grad()
    x = m.table
    outs = [x]
    for k in range(3):
        nei = H.split_mm(m.norm_adj, x)
        s_ = torch.nn.functional.leaky_relu(torch.matmul(nei + x, m.mat[f"W1_{k}"] + m.mat[f"b1_{k}"]), 0.2)
        b_ = torch.nn.functional.leaky_relu(torch.matmul(nei * x, m.mat[f"W2_{k}"] + m.mat[f"b2_{k}"]), 0.2)
        x = s_ + b_
        if p[k] > 0:
            x = x * H.message_drop(torch.ones_like(x), p[k], NG._layer_seed(seed, k))
        outs.append(H.normalize_rows(x))
    ref = torch.cat(outs, dim=1)
    (ref * w).sum().backward()
    np.testing.assert_allclose(got_out.cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    scale = max(float(v.grad.abs().max()) for v in m.parameters())
    for k, v in m.named_parameters():
        np.testing.assert_allclose(got[k].cpu().numpy(), v.grad.cpu().numpy(), rtol=2e-3, atol=2e-5 * scale, err_msg=k)
    # the loss path (restricted top layers, batch rows compact) with dropout: finite, differentiable, a new mask per pass
    b = torch.from_numpy(T.synth.sample_bpr_epoch(ds, 0)[:16]).to(DEV)
    m.zero_grad()
    l1 = m.loss(b)
    sum(l1).backward()
    assert all(torch.isfinite(v.grad).all() for v in m.parameters()) and float(m.table.grad.abs().sum()) > 0
    assert float(sum(m.loss(b))) != float(sum(l1))


def test_ngcf_dropout_loss_path_with_repeated_batch_nodes_matches_masked_operator_form():
    """The loss path with message dropout runs the top layer on the batch rows alone (ngcf.propagate_forward, "rows" form).
    A batch that names a node several times (popular positive items do) must give every slot of that node the node's ONE
    mask -- the reference drops the full [N, d] layer output (ngcf.py:85) -- so that the forward value and the gradient
    (sent through the node's first slot) belong to the same realisation.  Loss, table gradient and W / b gradients against
    the same pass assembled from unfused operators on ALL rows with the full-table mask; the forward is run-to-run deterministic."""
    from tagrec_amd import ngcf as NG
    ds = T.synth.make_cf_dataset(3000, 2500, 60000, seed=5)
    p = [0.2, 0.0, 0.4]
    cfg = T.get_config("ngcf", use_tag=False, dim_layer_list=[64, 64, 32], dim_latent=64, device=DEV, message_drop_list=p,
                       reg=1e-3)
    torch.manual_seed(1)
    m = T.NGCF(ds, config=cfg)
    m.train()
    b = torch.from_numpy(T.synth.sample_bpr_epoch(ds, 0)[:64].copy())
    b[8:24, 0] = b[0, 0]            # one user 17 times, one positive item 12 times, one negative 9 times (and as a positive)
    b[30:41, 1] = b[3, 1]
    b[44:52, 2] = b[3, 1]
    b = b.to(DEV)
    assert 3 * b.shape[0] * 16 <= m.table.shape[0]          # the compact top layer is taken

    def run(calls):
        m._drop_calls = calls
        m.zero_grad()
        l = m.loss(b)
        sum(l).backward()
        return [float(x) for x in l], {k: v.grad.clone() for k, v in m.named_parameters()}

    l1, g1 = run(6)
    l2, g2 = run(6)
    assert l1 == l2, "same seed, same batch: the forward pass is deterministic (every slot of a node carries one mask)"
    for k in g1:        # (the BPR backward adds repeated rows with float atomics: last-bit differences between runs)
        np.testing.assert_allclose(g2[k].cpu().numpy(), g1[k].cpu().numpy(), rtol=1e-4, atol=1e-6 * float(g1[k].abs().max()))
    seed = (int(m.drop_seed) << 24) + m._drop_calls
    m.zero_grad()
    x = m.table
    outs = [x]
    for k in range(3):
        nei = H.split_mm(m.norm_adj, x)
        s_ = torch.nn.functional.leaky_relu(torch.matmul(nei + x, m.mat[f"W1_{k}"] + m.mat[f"b1_{k}"]), 0.2)
        b_ = torch.nn.functional.leaky_relu(torch.matmul(nei * x, m.mat[f"W2_{k}"] + m.mat[f"b2_{k}"]), 0.2)
        x = s_ + b_
        if p[k] > 0:
            x = x * H.message_drop(torch.ones_like(x), p[k], NG._layer_seed(seed, k))
        outs.append(H.normalize_rows(x))
    ref = torch.cat(outs, dim=1)
    nu = ds.num["user"]
    U, I = ref[:nu], ref[nu:]
    loss, reg = H.triplet_loss(U, I, U, I, b, cfg["mul_loss_func"])
    (loss + cfg["reg"] * reg).backward()
    np.testing.assert_allclose(l1, [float(loss), float(cfg["reg"] * reg)], rtol=2e-5)
    scale = max(float(v.grad.abs().max()) for v in m.parameters())
    for k, v in m.named_parameters():
        np.testing.assert_allclose(g1[k].cpu().numpy(), v.grad.cpu().numpy(), rtol=2e-3, atol=2e-5 * scale, err_msg=k)


@pytest.mark.parametrize("name", ["lightgcn", "ngcf"])
def test_graphed_compact_restricted_step_matches_eager(name):
    """The COMPACT restricted step (layers only on the rows the batch's loss depends on: row-masked kernels, spmm_listed,
    batch-row gradient flags) makes no host read, so it replays as one captured HIP graph: same losses and parameters as
    the eager loop on a graph large enough for the compact path (3 B * 16 <= N)."""
    ds = T.synth.make_bipartite_device(6000, 5000, 200_000, seed=11, device=DEV)
    cfg = T.get_config(name, use_tag=False, dim_layer_list=[32, 32, 32], dim_latent=32, device=DEV, train_batch=64)
    cls = {"lightgcn": T.LightGCN, "ngcf": T.NGCF}[name]
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], 6000, 5000, cfg["norm_type"])
    g = T.Graph(rp, col, val, (n, n), symmetric=(cfg["norm_type"] in ("bi_norm", "plain")))
    g.transpose()
    out = []
    for use_graph, fuse in ((False, False), (True, False), (True, True)):
        # (True, True): the table's Adam update inside the last backward product AND the step captured -- the optimizer's step
        # counter and step-dependent factors live in device memory (tagrec_spmm_axpy_adam_graph_f32)
        torch.manual_seed(3)
        m = cls(ds, config=cfg, graph=g)
        m.train()
        opt = T.Adam(m.parameters(), lr=0.01, capturable=use_graph)
        if fuse:
            opt.fuse_into(m)
        prod = T.BPR_training_data(ds, config=cfg, seed=9)
        batches = [prod.all_train_data[i * 64:(i + 1) * 64] for i in range(12)]

        class _P:                                               # a producer that yields the same 12 full batches
            def reset(self):
                pass

            def mini_batch(self):
                return iter(batches)
        graphs = {} if use_graph else None
        losses = T.epoch_training(_P(), m.loss, opt, verbose=False, graphs=graphs)
        if use_graph:
            assert not graphs.get("errors"), graphs.get("errors")
            assert sum(isinstance(v, T.GraphedStep) for v in graphs.values()) == 1
        out.append((np.array(losses), {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}))
    (l0, s0) = out[0]
    for l1, s1 in out[1:]:
        assert len(l0) == len(l1) == 12
        np.testing.assert_allclose(l1, l0, rtol=5e-5)
        for k in s0:
            assert np.mean(np.abs(s1[k] - s0[k]) <= 2e-4) >= 0.99, k


@pytest.mark.parametrize("name", ["lightgcn", "ngcf"])
def test_adam_fused_into_the_last_backward_hop_is_bit_identical(name):
    """`Adam.fuse_into(model)`: the compact restricted LightGCN / NGCF step applies the table's Adam update in the epilogue
    of the backward product that lands on the table (no gradient tensor, no separate launch).  Same arithmetic in the same
    order: every parameter, exp_avg and exp_avg_sq after four steps are bit-identical to the separate optimizer launch;
    LightGCN with reg != 0 (an extra term lands on the gradient after the last hop) and the all-rows paths hand over a
    gradient as usual."""
    ds = T.synth.make_bipartite_device(6000, 5000, 200_000, seed=11, device=DEV)
    e = ds.edge_index["train"]
    norm = "bi_norm" if name == "lightgcn" else "ngcf"
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], 6000, 5000, norm)
    g = T.Graph(rp, col, val, (n, n), symmetric=(name == "lightgcn"))
    g.transpose()
    cls = {"lightgcn": T.LightGCN, "ngcf": T.NGCF}[name]
    for reg, layers, restrict in ((0.0, [32, 32, 32], True), (1e-3, [32, 32, 32], True), (0.0, [32], True), (0.0, [32, 16], True),
                                 (0.0, [32, 32], False)):           # restrict False: LightGCN's all-rows path fuses too
        if name == "ngcf" and not restrict:
            continue
        cfg = T.get_config(name, use_tag=False, dim_layer_list=layers, dim_latent=32, device=DEV, train_batch=64, reg=reg,
                           restrict_forward=restrict)
        out = []
        for fuse in (False, True):
            torch.manual_seed(3)
            m = cls(ds, config=cfg, graph=g)
            m.train()
            opt = T.Adam(m.parameters(), lr=0.01)
            if fuse:
                opt.fuse_into(m)
            prod = T.BPR_training_data(ds, config=cfg, seed=9)
            # batches without a repeated node: index_add on repeated rows is an atomic scatter whose order (hence the last
            # bit of a sum) changes from run to run, which would hide what is compared here
            pool = prod.all_train_data[:4096].cpu().numpy()
            batches, used_u, used_i, cur = [], set(), set(), []
            for u, i_, j_ in pool:
                if u in used_u or i_ in used_i or j_ in used_i or i_ == j_:
                    continue
                used_u.add(u); used_i.update((i_, j_)); cur.append((u, i_, j_))
                if len(cur) == 64:
                    batches.append(torch.tensor(cur, device=DEV)); cur = []; used_u, used_i = set(), set()
                if len(batches) == 4:
                    break
            assert len(batches) == 4
            losses = []
            for i in range(4):
                lossx = m.loss(batches[i])
                opt.zero_grad()
                sum(lossx).backward()
                used_fused = m.table.grad is None
                opt.step()
                losses.append([float(v.detach()) for v in lossx])
            assert all(opt.state[id(p)]["t"] == 4 for p in m.parameters())
            out.append((losses, [p.detach().clone() for p in m.parameters()],
                        [opt.state[id(p)][k].clone() for p in m.parameters() for k in ("m", "v")], used_fused))
        (l0, p0, s0, f0), (l1, p1, s1, f1) = out
        assert not f0 and f1 == (name == "ngcf" or reg == 0.0)
        assert l0 == l1
        assert all(torch.equal(a, b) for a, b in zip(p0, p1)) and all(torch.equal(a, b) for a, b in zip(s0, s1))


def test_fused_adam_ownership_and_commit():
    """`Adam.fuse_into` hygiene: (1) a second optimizer built over the same model revokes the first one's fusion -- the step
    then hands a gradient to the NEW optimizer instead of updating through the stale one's lr and state; (2) `fuse_into`
    refuses a model whose table it does not own; (3) a second fused backward() before step() raises, and the mark is set by
    `fused_commit` (after the launch), not by `fused_state`."""
    ds = T.synth.make_bipartite_device(6000, 5000, 200_000, seed=11, device=DEV)
    cfg = T.get_config("lightgcn", use_tag=False, dim_layer_list=[32, 32], dim_latent=32, device=DEV, train_batch=64)
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], 6000, 5000, "bi_norm")
    g = T.Graph(rp, col, val, (n, n), symmetric=True)
    torch.manual_seed(3)
    m = T.LightGCN(ds, config=cfg, graph=g)
    m.train()
    b = T.BPR_training_data(ds, config=cfg, seed=9).all_train_data[:64]
    old = T.Adam(m.parameters(), lr=0.5).fuse_into(m)
    sum(m.loss(b)).backward()
    assert m.table.grad is None                       # fused: no gradient tensor
    with pytest.raises(T._lib.TagrecError, match="second backward"):
        sum(m.loss(b)).backward()
    old.step()
    before = m.table.detach().clone()
    new = T.Adam(m.parameters(), lr=0.01)             # new run, same model: the old fusion must be gone
    lossx = m.loss(b)
    new.zero_grad()
    sum(lossx).backward()
    assert m.table.grad is not None and torch.equal(m.table.detach(), before)
    new.step()
    assert float((m.table.detach() - before).abs().max()) <= 0.0100001        # moved with the NEW lr (Adam: |step| <= lr)
    other = T.LightGCN(ds, config=cfg, graph=g)
    with pytest.raises(T._lib.TagrecError, match="not one of this optimizer"):
        new.fuse_into(other)
    # fused_state alone leaves no mark
    new.fuse_into(m)
    new.fused_state(m.table)
    new.fused_state(m.table)
    new.fused_commit(m.table)
    with pytest.raises(T._lib.TagrecError, match="second backward"):
        new.fused_state(m.table)


@pytest.mark.parametrize("name", ["lightgcn", "ngcf"])
def test_restricted_step_reads_no_unwritten_row_model_level_poison(name, monkeypatch):
    """The single-GPU restricted LightGCN / NGCF step leaves the rows a layer does not compute UNWRITTEN in torch.empty
    buffers and tells every later reader which rows are valid (row masks, always-consulted operand flags, dz_flags).  Here
    every float buffer the step allocates with torch.empty / empty_like / Tensor.new_empty is pre-filled with NaN (the
    allocator hands back blocks with stale finite values otherwise, which would hide a stale read): at 50 k nodes the loss,
    the gradients -- and with the optimizer fused into the last hop the updated table and Adam state -- must be finite and
    equal to the all-rows step's."""
    ds = T.synth.make_bipartite_device(25_000, 25_000, 1_000_000, seed=13, device=DEV)
    e = ds.edge_index["train"]
    norm = "bi_norm" if name == "lightgcn" else "ngcf"
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], 25_000, 25_000, norm)
    g = T.Graph(rp, col, val, (n, n), symmetric=(name == "lightgcn"))
    g.transpose()
    cls = {"lightgcn": T.LightGCN, "ngcf": T.NGCF}[name]
    cfg = T.get_config(name, use_tag=False, dim_layer_list=[64, 64, 64], dim_latent=64, device=DEV, train_batch=128)
    batch = T.BPR_training_data(ds, config=cfg, seed=3).all_train_data[:128]
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def poisoned_empty(*a, **kw):
        t = real_empty(*a, **kw)
        return t.fill_(float("nan")) if t.is_floating_point() and t.is_cuda else t

    def poisoned_empty_like(*a, **kw):
        t = real_empty_like(*a, **kw)
        return t.fill_(float("nan")) if t.is_floating_point() and t.is_cuda else t

    def run(poison, restrict, fuse):
        torch.manual_seed(5)
        if name == "ngcf":
            from tagrec_amd import ngcf as NG
            NG.RESTRICT_FORWARD = restrict
        m = cls(ds, config=dict(cfg, restrict_forward=restrict), graph=g)
        m.train()
        opt = T.Adam(m.parameters(), lr=0.01)
        if fuse:
            opt.fuse_into(m)
        if poison:
            monkeypatch.setattr(torch, "empty", poisoned_empty)
            monkeypatch.setattr(torch, "empty_like", poisoned_empty_like)
        try:
            lossx = m.loss(batch)
            opt.zero_grad()
            sum(lossx).backward()
            grads = {k: (None if p.grad is None else p.grad.clone()) for k, p in m.named_parameters()}
            opt.step()
        finally:
            monkeypatch.undo()
        state = {k: p.detach().clone() for k, p in m.named_parameters()}
        adam = [opt.state[id(p)][kk].clone() for p in m.parameters() for kk in ("m", "v")]
        return [float(v.detach()) for v in lossx], grads, state, adam

    try:
        base = run(False, False, False)                  # all-rows step, nothing poisoned
        for fuse in (False, True):
            got = run(True, True, fuse)                  # restricted step, every torch.empty poisoned
            np.testing.assert_allclose(got[0], base[0], rtol=1e-5)
            for k in base[2]:
                assert torch.isfinite(got[2][k]).all(), k
                # (the first Adam step moves an element by lr * g / (|g| + 1e-8): last-bit differences of a near-zero gradient
                # become visible, so a handful of elements may differ by more than 2e-4 -- never by more than 2 lr)
                diff = (got[2][k] - base[2][k]).abs()
                assert float((diff <= 2e-4).float().mean()) >= 0.999 and float(diff.max()) <= 0.0201, k
            for a, b in zip(got[3], base[3]):
                assert torch.isfinite(a).all()
            if not fuse:
                for k, gb in base[1].items():
                    assert got[1][k] is not None and torch.isfinite(got[1][k]).all(), k
                    scale = float(gb.abs().max()) + 1e-30
                    np.testing.assert_allclose(got[1][k].cpu().numpy(), gb.cpu().numpy(), rtol=2e-3, atol=2e-5 * scale, err_msg=k)
    finally:
        if name == "ngcf":
            from tagrec_amd import ngcf as NG
            NG.RESTRICT_FORWARD = True


def test_multi_tensor_adam_is_bit_identical_to_one_launch_per_tensor():
    """`Adam.step` updates the small tensors of a model in one launch (tagrec_adam_multi_f32); three steps on tensors of odd
    sizes / alignments against the same optimizer with one launch per tensor (MULTI_MAX_NUMEL = 0), bit for bit, and against
    torch.optim.Adam within float rounding."""
    g = torch.Generator().manual_seed(4)
    shapes = [(1,), (3,), (17, 5), (128, 32), (4129,), (64, 64), (2, 3, 7), (1000, 130)] + [(33,)] * 70
    base = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(3)]

    def run(kind):
        ps = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]
        if kind == "torch":
            opt = torch.optim.Adam(ps, lr=0.01)
        else:
            opt = T.Adam(ps, lr=0.01)
            if kind == "single":
                opt.MULTI_MAX_NUMEL = 0
        for gs in grads:
            for p, gg in zip(ps, gs):
                p.grad = gg.to(DEV)
            ps[5].grad = ps[5].grad if gs is not grads[1] else None         # a tensor that skips a step keeps its own count
            opt.step()
        return [p.detach().clone() for p in ps]

    multi, single, ref = run("multi"), run("single"), run("torch")
    for a, b, c in zip(multi, single, ref):
        assert torch.equal(a, b)
        np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=2e-6, atol=1e-7)
