"""Generate tests/golden/*.npz by running the REFERENCE itself (this container only).

    python oracle/make_golden.py            # writes every fixture

The reference is imported unmodified from /root/reference with the three
in-process shims of SURVEY.md section 8c (collections.Iterable, a stub
tensorboardX module, np.int).  Nothing of the reference is copied: fixtures
hold inputs and the reference's outputs only.  `/root/reference` does not exist
on the GPU box, so this script never runs there; the fixtures travel instead.
"""
import collections
import collections.abc
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    collections.Iterable = collections.abc.Iterable                    # utility/utils.py:6
    tb = types.ModuleType("tensorboardX")
    tb.SummaryWriter = object
    sys.modules["tensorboardX"] = tb                                   # utility/word.py:1
    if not hasattr(np, "int"):
        np.int = int                                                   # data/utils.py:73-74,91-92
    sys.path.insert(0, REF)
    sys.argv = ["main.py", "--model", "lightgcn"]
    from utility.word import CFG                                       # parses argv at import
    from utility import config as refcfg
    import model as M
    import model.help as H
    import data.utils as data_utils
    import train_data.utils as td_utils
    import train_data.abstract as td_abs
    import training.utils as tr_utils
    import training.basic_test as basic_test
    import training.basic_train as basic_train
    return dict(CFG=CFG, cfg=refcfg, M=M, H=H, data_utils=data_utils, td_utils=td_utils, td_abs=td_abs,
                tr_utils=tr_utils, basic_test=basic_test, basic_train=basic_train)


def main(only=None):
    import scipy.sparse as sp
    import torch
    sys.path.insert(0, ROOT)
    import tagrec_amd
    synth = tagrec_amd.synth
    R = import_reference()
    CFG, M, H = R["CFG"], R["M"], R["H"]
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)

    def scipy_data(ds):
        """Object shaped like TGCN_load for the model constructors."""
        o = types.SimpleNamespace()
        o.num = dict(ds.num)
        mk = lambda c: sp.coo_matrix((c.data, (c.row, c.col)), shape=c.shape, dtype=np.float32)
        o.ui_adj = mk(ds.ui_adj)
        if ds.ut_adj is not None:
            o.ut_adj, o.it_adj = mk(ds.ut_adj), mk(ds.it_adj)
        return o

    def blocks(ds):
        d = {"n_user": ds.num["user"], "n_item": ds.num["item"], "n_tag": ds.num.get("tag", 0),
             "ui_row": ds.ui_adj.row, "ui_col": ds.ui_adj.col}
        if ds.ut_adj is not None:
            d.update(ut_row=ds.ut_adj.row, ut_col=ds.ut_adj.col, it_row=ds.it_adj.row, it_col=ds.it_adj.col)
        return d

    def set_cfg(model, **kw):
        CFG.update(R["cfg"].dict_map[model])
        CFG.update(model=model, device=torch.device("cpu"), split_adj_k=1, node_drop=0.0,
                   message_drop_list=[0.0] * 4, reg=0.0)
        CFG.update(kw)

    def coalesced(adj):
        a = adj.coalesce()
        return a.indices().numpy(), a.values().numpy()

    toy = synth.make_cf_dataset(40, 30, 300, seed=1, n_tag=12, n_assign=200)
    med = synth.make_cf_dataset(200, 300, 5000, seed=2)

    # ------------------------------------------------------------------ loaders (N3): reference TGCN_load on toy files
    if only in (None, "loader"):
        import tempfile
        import data as ref_data
        with tempfile.TemporaryDirectory() as tmp:
            tagrec_amd.data.write_dataset(toy, tmp, "toyset")
            # a repeated user line and a repeated item exercise the merge / de-dup rules of data/utils.py:23-46
            with open(os.path.join(tmp, "toyset", "train.txt"), "a") as f:
                f.write("3 1 1 2\n")
            CFG.update(data_root=tmp, dataset="toyset", has_val=False, cpu_core=1)
            ld = ref_data.TGCN_load(types.SimpleNamespace(pool=None))
            fx = {"files." + n: np.frombuffer(open(os.path.join(tmp, "toyset", n), "rb").read(), dtype=np.uint8)
                  for n in ("train.txt", "test.txt", "user_item_tag.txt")}
            fx.update({"num." + k: int(v) for k, v in ld.num.items()})
            for split in ("train", "test"):
                e = np.asarray(ld.edge_index[split])
                fx["edges." + split] = e[np.lexsort((e[:, 1], e[:, 0]))]
            for nm in ("ui_adj", "ut_adj", "it_adj"):
                m = getattr(ld, nm).tocsr().tocoo()          # duplicates summed, canonical order
                fx[nm + ".row"], fx[nm + ".col"], fx[nm + ".data"] = m.row, m.col, m.data
                fx[nm + ".shape"] = np.array(m.shape)
            fx["uit_data"] = np.asarray(ld.uit_data)
            np.savez_compressed(os.path.join(OUT, "loader_toy.npz"), **fx)
            print("wrote loader_toy", ld.num)
        if only == "loader":
            return

    def batches_for(ds, n_batch, B, seed):
        tri = synth.sample_bpr_epoch(ds, seed)
        return [tri[k * B:(k + 1) * B] for k in range(n_batch)]

    def run_steps(model, loss_fn, batches, lr, n_steps):
        """zero_grad / backward / Adam.step exactly as basic_train.epoch_training does,
        via the reference's own epoch_training on a tiny producer object."""
        prod = types.SimpleNamespace(reset=lambda: None,
                                     mini_batch=lambda: iter([torch.from_numpy(b) for b in batches[:n_steps]]))
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        return R["basic_train"].epoch_training(prod, loss_fn, opt)

    # ------------------------------------------------------------------ N4 siblings: DGCF, DisenGCN (dynamic edge values)
    def sibling_case(name, ds, model_name, use_tag, n_layer, D, K, T, reg, B, seed):
        set_cfg(model_name, use_tag=use_tag, dim_layer_list=[D] * n_layer, dim_latent=D, reg=reg, factor_k=K, iterate_k=T)
        torch.manual_seed(2020)
        model = {"dgcf": M.DGCF, "disengcn": M.DisenGCN}[model_name](scipy_data(ds))
        model.train()
        fx = blocks(ds)
        fx.update(n_layer=n_layer, D=D, factor_k=K, iterate_k=T, reg=reg, loss_kind=CFG["mul_loss_func"],
                  norm_type=CFG["norm_type"], use_tag=int(use_tag), lr=0.01)
        idx = model.norm_adj._indices().numpy()
        fx["adj_idx"] = idx.copy()
        for k, v in model.state_dict().items():
            fx["init." + k] = v.numpy().copy()
        bs = batches_for(ds, 3, B, seed)
        fx["batches"] = np.stack(bs)
        cor = torch.zeros(2, 4, dtype=torch.long)                       # the second half of a DGCF batch; unused by loss()
        with torch.no_grad():
            for t, o in enumerate(model.forward()):
                fx[f"out.{t}"] = o.numpy().copy()
            if model_name == "dgcf":
                layer_a = model.forward(out_A=True)                     # per layer: K sparse matrices of routing weights
                fx["out_A"] = np.stack([np.stack([a._values().numpy() for a in la]) for la in layer_a])
        lx = model.loss((torch.from_numpy(bs[0]), cor))
        fx["loss_parts"] = np.array([float(v) for v in lx], dtype=np.float64)
        model.zero_grad()
        sum(lx).backward()
        for k, p in model.named_parameters():
            fx["grad." + k] = p.grad.numpy().copy()
        init = {k: v.clone() for k, v in model.state_dict().items()}
        for n in (1, 3):
            model.load_state_dict(init)
            prod = types.SimpleNamespace(reset=lambda: None,
                                         mini_batch=lambda: iter([(torch.from_numpy(b), cor) for b in bs[:n]]))
            opt = torch.optim.Adam(model.parameters(), lr=0.01)
            losses = R["basic_train"].epoch_training(prod, model.loss, opt)
            fx[f"step{n}.losses"] = np.array(losses, dtype=np.float64)
            for k, v in model.state_dict().items():
                fx[f"step{n}." + k] = v.numpy().copy()
        model.eval()
        with torch.no_grad():
            users = torch.arange(0, min(ds.num["user"], 16))
            fx["predict.users"] = users.numpy()
            fx["predict.rating"] = model.predict_rating(users).numpy().copy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
        print("wrote", name, {k: v for k, v in fx.items() if np.ndim(v) == 0})

    if only in (None, "siblings"):
        sibling_case("dgcf_toy", toy, "dgcf", True, 2, 64, 4, 2, 1e-3, 64, 31)
        sibling_case("dgcf_med", med, "dgcf", False, 1, 32, 2, 3, 0.0, 256, 32)
        sibling_case("disengcn_toy", toy, "disengcn", True, 2, 64, 4, 2, 1e-3, 64, 33)
        # KGAT: the data object carries the duplicate (user, tag) / (item, tag) entries TGCN_load's COO matrices have
        # (one per (u, i, t) assignment, data/tgcn_load.py:21-22), which torch.sparse.softmax sums before normalising
        import data as ref_data

        def kgat_data(ds, layout):
            o = scipy_data(ds)
            for nm in ("ut_adj", "it_adj"):
                c = getattr(ds, nm)
                cnt = c.data.astype(np.int64)
                setattr(o, nm, sp.coo_matrix((np.ones(int(cnt.sum()), np.float32), (np.repeat(c.row, cnt), np.repeat(c.col, cnt))),
                                             shape=c.shape))
            wired = ref_data.TGCN_load.create_edge(o)                      # dict k -> [2, E] (what com.py:78-79 passes)
            o.create_edge = (lambda: wired) if layout == "wired" else (lambda: {k: v.T.copy() for k, v in wired.items()})
            return o

        def kgat_case(name, ds, agg_type, layers, D, Dr, reg, B, seed, layout="pairs"):
            set_cfg("kgat", use_tag=True, dim_layer_list=list(layers), dim_latent=D, dim_relation=Dr, reg=reg, agg_type=agg_type)
            torch.manual_seed(2020)
            model = M.KGAT(kgat_data(ds, layout))
            model.train()
            fx = blocks(ds)
            fx.update(layers=np.array(layers), D=D, dim_relation=Dr, reg=reg, agg_type=agg_type, lr=0.01,
                      transe_reg=CFG["transe_reg"], cor_reg=CFG["cor_reg"])
            fx["layout"] = layout
            for k, e in model.edge_index_dict.items():
                fx[f"edges.{k}"] = e.numpy().copy()
            for k, v in model.state_dict().items():
                fx["init." + k] = v.numpy().copy()
            bs = batches_for(ds, 3, B, seed)
            fx["batches"] = np.stack(bs)
            with torch.no_grad():
                for t, o in enumerate(model.forward()):
                    fx[f"out.{t}"] = o.numpy().copy()
            lx = model.loss(torch.from_numpy(bs[0]))
            fx["loss_parts"] = np.array([float(v) for v in lx], dtype=np.float64)
            model.zero_grad()
            sum(lx).backward()
            for k, p in model.named_parameters():
                fx["grad." + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
            # TransE phase on (head, relation, pos tail, neg tail) rows drawn from the model's own edge lists
            rng = np.random.RandomState(seed)
            n_all = ds.num["user"] + ds.num["item"] + ds.num["tag"]
            quads = []
            for k, e in model.edge_index_dict.items():
                e = e.numpy() if layout == "pairs" else e.numpy().T
                pick = rng.randint(0, len(e), 12)
                quads.append(np.stack([e[pick, 0], np.full(12, k), e[pick, 1], rng.randint(0, n_all, 12)], 1))
            tb = np.concatenate(quads).astype(np.int64)
            fx["transe_batch"] = tb
            lt = model.transe_loss(torch.from_numpy(tb))
            fx["transe_loss_parts"] = np.array([float(v) for v in lt], dtype=np.float64)
            model.zero_grad()
            sum(lt).backward()
            for k, p in model.named_parameters():
                if p.grad is not None:
                    fx["transe_grad." + k] = p.grad.numpy().copy()
            init = {k: v.clone() for k, v in model.state_dict().items()}
            for n in (1, 3):
                model.load_state_dict(init)
                losses = run_steps(model, model.loss, bs, 0.01, n)
                fx[f"step{n}.losses"] = np.array(losses, dtype=np.float64)
                for k, v in model.state_dict().items():
                    fx[f"step{n}." + k] = v.numpy().copy()
            model.eval()
            with torch.no_grad():
                users = torch.arange(0, min(ds.num["user"], 16))
                fx["predict.users"] = users.numpy()
                fx["predict.rating"] = model.predict_rating(users).numpy().copy()
            np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
            print("wrote", name, {k: v for k, v in fx.items() if np.ndim(v) == 0})

        kgat_case("kgat_toy", toy, "bi_inter", (64, 32, 16), 64, 64, 1e-3, 64, 34)                 # [E, 2] edge arrays
        kgat_case("kgat_toy_wired", toy, "bi_inter", (64, 32), 64, 32, 1e-3, 64, 36, "wired")     # [2, E], as com.py wires it
        kgat_case("kgat_toy_default", toy, "bi_agg", (64,), 64, 32, 1e-3, 64, 35, "wired")         # default agg_type: no propagation
        if only == "siblings":
            return

    # ------------------------------------------------------------------ adjacency (A1-A3)
    fx = blocks(toy)
    for use_tag in (False, True):
        for norm in ("bi_norm", "ngcf", "si_norm", "si_norm_self", "plain"):
            adj = H.creat_adj(scipy_data(toy), use_tag, norm, 1, torch.device("cpu"))
            idx, val = coalesced(adj)
            fx[f"{norm}_{int(use_tag)}_idx"], fx[f"{norm}_{int(use_tag)}_val"] = idx, val
    folds = H.creat_adj(scipy_data(toy), True, "bi_norm", 3, torch.device("cpu"))
    for k, f in enumerate(folds):
        fx[f"fold3_{k}_idx"], fx[f"fold3_{k}_val"] = coalesced(f)
        fx[f"fold3_{k}_shape"] = np.array(f.shape)
    np.savez_compressed(os.path.join(OUT, "adj_toy.npz"), **fx)

    # ------------------------------------------------------------------ helpers for model cases
    def model_case(name, ds, model_name, use_tag, layers, D, reg, B, seed):
        set_cfg(model_name, use_tag=use_tag, dim_layer_list=list(layers), dim_latent=D, reg=reg)
        torch.manual_seed(2020)                                          # init_seed (utility/utils.py:10-15)
        cls = {"lightgcn": M.LightGCN, "ngcf": M.NGCF}[model_name]
        model = cls(scipy_data(ds))
        model.train()
        fx = blocks(ds)
        fx.update(layers=np.array(layers), D=D, reg=reg, loss_kind=CFG["mul_loss_func"], norm_type=CFG["norm_type"],
                  use_tag=int(use_tag), lr=0.01)
        for k, v in model.state_dict().items():
            fx["init." + k] = v.numpy().copy()
        bs = batches_for(ds, 3, B, seed)
        fx["batches"] = np.stack(bs)
        # forward + per-layer operator trace (reference operators only)
        with torch.no_grad():
            outs = model.forward()
            for t, o in enumerate(outs):
                fx[f"out.{t}"] = o.numpy().copy()
            x = torch.cat(list(model.embed), dim=0)
            if model_name == "lightgcn":
                for k in range(len(layers)):
                    x = H.split_mm(model.norm_adj, x)
                    fx[f"raw.{k}"] = x.numpy().copy()
        # loss parts + grads at the init point
        lx = model.loss(torch.from_numpy(bs[0]))
        fx["loss_parts"] = np.array([float(v) for v in lx], dtype=np.float64)
        model.zero_grad()
        sum(lx).backward()
        for k, p in model.named_parameters():
            fx["grad." + k] = p.grad.numpy().copy()
        # 1 and 3 Adam steps from the same init, the reference's own epoch_training
        init = {k: v.clone() for k, v in model.state_dict().items()}
        for n in (1, 3):
            model.load_state_dict(init)
            losses = run_steps(model, model.loss, bs, 0.01, n)
            fx[f"step{n}.losses"] = np.array(losses, dtype=np.float64)
            for k, v in model.state_dict().items():
                fx[f"step{n}." + k] = v.numpy().copy()
        model.eval()
        with torch.no_grad():
            users = torch.arange(0, min(ds.num["user"], 16))
            fx["predict.users"] = users.numpy()
            fx["predict.rating"] = model.predict_rating(users).numpy().copy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
        print("wrote", name, {k: v for k, v in fx.items() if np.ndim(v) == 0})

    if only in (None, "d256"):
        # C5's row width (dim 256, 3 layers) on the toy graph: the widest vector kernel, forward / loss / grads / Adam
        model_case("lightgcn_toy_d256", toy, "lightgcn", True, (256, 256, 256), 256, 1e-3, 64, 16)
        if only == "d256":
            return
    model_case("lightgcn_toy", toy, "lightgcn", True, (64, 64), 64, 1e-3, 64, 11)
    model_case("lightgcn_med", med, "lightgcn", False, (64, 64, 64), 64, 0.0, 512, 12)
    model_case("lightgcn_toy_d32", toy, "lightgcn", False, (32,), 32, 1e-2, 48, 13)
    model_case("ngcf_toy", toy, "ngcf", True, (64, 32, 16), 64, 1e-3, 64, 14)
    model_case("ngcf_med", med, "ngcf", False, (64, 64, 64), 64, 0.0, 512, 15)

    # ------------------------------------------------------------------ TGCN toy (T1-T6)
    set_cfg("tgcn", use_tag=True, dim_layer_list=[16, 16], dim_latent=16, reg=1e-3, neighbor_k=5)
    sd = scipy_data(toy)
    sd.num["weight"] = toy.num["weight"]
    np.random.seed(7)
    mats = [sd.ui_adj, sd.ut_adj, sd.ui_adj.transpose(), sd.it_adj, sd.ut_adj.transpose(), sd.it_adj.transpose()]
    tables = [R["data_utils"].all_neighbor_sample((m, int(max(m.tocsr().getnnz(1))))) for m in mats]
    sd.get_all_neighbor = lambda: tables
    torch.manual_seed(2020)
    tg = M.TGCN(sd)
    tg.train()
    fx = blocks(toy)
    fx.update(layers=np.array([16, 16]), D=16, reg=1e-3, neighbor_k=5, margin=CFG["margin"],
              transtag_reg=CFG["transtag_reg"], n_weight=toy.num["weight"], lr=0.01)
    for r, (ids, wts) in enumerate(tables):
        fx[f"nbr{r}.ids"], fx[f"nbr{r}.wts"] = ids[:, :5].copy(), wts[:, :5].copy()
    for k, v in tg.state_dict().items():
        fx["init." + k] = v.numpy().copy()
    bs = batches_for(toy, 3, 64, 21)
    fx["batches"] = np.stack(bs)
    rng = np.random.RandomState(5)
    uit = toy.uit_data
    tt = np.stack([uit[:64, 0], uit[:64, 2], uit[:64, 1], rng.randint(0, toy.num["item"], 64)], axis=1).astype(np.int64)
    fx["tt_batch"] = tt
    with torch.no_grad():
        for t, o in enumerate(tg.forward()):
            fx[f"out.{t}"] = o.numpy().copy()
    lx = tg.loss(torch.from_numpy(bs[0]))
    fx["loss_parts"] = np.array([float(v) for v in lx])
    tg.zero_grad(); sum(lx).backward()
    for k, p in tg.named_parameters():
        fx["grad." + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    lt = tg.transtag_loss(torch.from_numpy(tt))
    fx["tt_loss_parts"] = np.array([float(v) for v in lt])
    tg.zero_grad(); sum(lt).backward()
    for k in ("embed.user", "embed.item", "embed.tag"):
        fx["tt_grad." + k] = dict(tg.named_parameters())[k].grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "tgcn_toy.npz"), **fx)
    print("wrote tgcn_toy")

    # ------------------------------------------------------------------ producer (A16, A17)
    fx = {}
    prod = types.SimpleNamespace()
    cases = [(1000, 512), (1024, 512), (1536, 512), (80000, 512), (100, 512), (1025, 512), (2047, 1024)]
    for n, B in cases:
        prod.all_train_data = np.arange(n)
        prod.batch_size = B
        got = [(int(b[0]), int(b[-1]) + 1) for b in R["td_abs"].Abstract_training_data.mini_batch(prod)]
        fx[f"mb_{n}_{B}"] = np.array(got)
    np.random.seed(2020)
    tr = toy.edge_index["train"]
    fx["neg_seed"] = 2020
    fx["neg_pos"] = tr
    fx["neg_out"] = R["td_utils"].sample_neg_item(tr, toy.user_items["train"], toy.num["item"])
    fx["split5"] = np.array([len(c) for c in R["td_utils"].split_data(np.arange(103), 5)])
    np.savez_compressed(os.path.join(OUT, "producer.npz"), **fx)
    fx_users = {str(u): np.array(v) for u, v in toy.user_items["train"].items()}
    np.savez_compressed(os.path.join(OUT, "producer_user_items.npz"), **fx_users)

    # ------------------------------------------------------------------ metrics (P1 + training/utils.py)
    rng = np.random.RandomState(3)
    nu, ni = 24, 60
    rating = rng.rand(nu, ni).astype(np.float32)
    train_items = {u: sorted(rng.choice(ni, rng.randint(1, 8), replace=False).tolist()) for u in range(nu)}
    test_items = {u: sorted(rng.choice(ni, rng.randint(1, 6), replace=False).tolist()) for u in range(nu)}
    CFG["topks"] = [10, 20]
    masked = torch.from_numpy(rating.copy())
    for u in range(nu):
        masked[u, train_items[u]] = -(1 << 10)                          # basic_test.py:47
    _, top = torch.topk(masked, k=20)
    res = R["basic_test"].test_users([test_items[u] for u in range(nu)], top.numpy())
    fx = {"rating": rating, "top": top.numpy(), "topks": np.array([10, 20])}
    for u in range(nu):
        fx[f"train.{u}"], fx[f"test.{u}"] = np.array(train_items[u]), np.array(test_items[u])
    for k, v in res.items():
        fx["res." + k] = np.array(v, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **fx)

    # ------------------------------------------------------------------ end-to-end Recall@20 at C1 scale
    c1 = synth.make_cf_dataset()                                        # 943 x 1682, 100k edges, seed 0
    set_cfg("lightgcn", use_tag=False, dim_layer_list=[64, 64], dim_latent=64, reg=0.0)
    torch.manual_seed(2020)
    model = M.LightGCN(scipy_data(c1))
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    init_sum = float(sum(p.double().sum() for p in model.parameters()))
    EPOCHS, B = 6, 512
    loss_curve = []
    for ep in range(EPOCHS):
        model.train()
        tri = torch.from_numpy(synth.sample_bpr_epoch(c1, 2020 + ep))
        prod = types.SimpleNamespace(all_train_data=tri, batch_size=B, reset=lambda: None)
        prod.mini_batch = lambda: R["td_abs"].Abstract_training_data.mini_batch(prod)
        losses = R["basic_train"].epoch_training(prod, model.loss, opt)
        loss_curve.append(float(np.mean(losses)))
        print("e2e epoch", ep, loss_curve[-1], flush=True)
    model.eval()
    users = sorted(c1.user_items["test"].keys())
    tot = collections.defaultdict(lambda: np.zeros(2))
    with torch.no_grad():
        for ub in R["tr_utils"].minibatch(users, 512):
            if not len(ub):
                continue
            rating = model.predict_rating(torch.tensor(ub))
            for r, u in enumerate(ub):
                rating[r, c1.user_items["train"].get(u, [])] = -(1 << 10)
            _, top = torch.topk(rating, k=20)
            res = R["basic_test"].test_users([c1.user_items["test"][u] for u in ub], top.numpy())
            for k, v in res.items():
                tot[k] += np.array(v)
    fx = {"epochs": EPOCHS, "batch": B, "lr": 0.01, "init_sum": init_sum, "loss_curve": np.array(loss_curve),
          "n_test_users": len(users), "n_train_edges": len(c1.edge_index["train"]),
          "edge_checksum": int(c1.edge_index["train"].astype(np.int64).sum())}
    for k, v in tot.items():
        fx["res." + k] = v / len(users)
    np.savez_compressed(os.path.join(OUT, "e2e_c1_lightgcn.npz"), **fx)
    print("e2e recall@[10,20]", fx["res.recall"], "ndcg", fx["res.ndcg"])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)      # optional: a single section ("loader", "siblings")
