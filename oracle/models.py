"""Oracle: model forward / loss restatement on PyTorch CPU.  TEST INFRASTRUCTURE.

Functional (no nn.Module) restatement of the reference's hot path.  Parameters
travel as plain dicts keyed by the reference's `state_dict` names so golden
fixtures interchange.  Gradients come from autograd over the same op sequence
the reference runs, so they are the reference's gradients.

Citations are into /root/reference.
"""
import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- adjacency
def csr_to_torch(csr):
    """`sp2tensor` (model/help/adj.py:144-150): COO, int64 indices, fp32 values,
    not coalesced (the entries are already row-major sorted)."""
    idx = torch.stack([torch.from_numpy(csr.rows()), torch.from_numpy(csr.col.astype(np.int64))])
    return torch.sparse_coo_tensor(idx, torch.from_numpy(csr.val), csr.shape)


def split_mm(adj, x):
    """`split_mm` (adj.py:158-167): one sparse tensor, or a list of row folds whose
    products are concatenated along dim 0."""
    if isinstance(adj, (list, tuple)):
        return torch.cat([torch.sparse.mm(a, x) for a in adj], dim=0)
    return torch.sparse.mm(adj, x)


def node_drop_mask(nnz, drop, rand):
    """`node_drop` (adj.py:170-191): keep edge iff int(rand + (1-drop)) != 0, kept
    values are divided by (1-drop).  `rand` is the uniform[0,1) draw per edge."""
    keep = 1.0 - drop
    return (rand + keep).to(torch.int32).bool(), keep


# --------------------------------------------------------------------------- losses
def mul_loss(u, p, n, kind):
    """`mul_loss` (model/help/loss.py:4-12)."""
    pos = (u * p).sum(dim=1)
    neg = (u * n).sum(dim=1)
    if kind == "logsigmoid":
        return -F.logsigmoid(pos - neg).mean()
    return F.softplus(neg - pos).mean()


def l2reg_loss(*embs):
    """`l2reg_loss` (loss.py:27-32): 0.5 * sum ||e||_F^2 / rows(first)."""
    tot = 0
    for e in embs:
        tot = tot + e.norm(2).pow(2)
    return 0.5 * tot / float(embs[0].shape[0])


def transtag_loss(h, r, tp, tn, margin):
    """`transtag_loss` (loss.py:35-41)."""
    ps = torch.norm(h + r - tp, p=2, dim=1)
    ns = torch.norm(h + r - tn, p=2, dim=1)
    return torch.relu(margin + ps - ns).mean()


# --------------------------------------------------------------------------- init
def xavier_tables(shapes, seed):
    """`init_seed` (utility/utils.py:10-15) then `xavier_uniform_` on every
    parameter in registration order (lightgcn.py:37-47, ngcf.py:39-60)."""
    torch.manual_seed(seed)
    out = []
    for shp in shapes:
        t = torch.empty(*shp)
        torch.nn.init.xavier_uniform_(t)
        out.append(t)
    return out


# --------------------------------------------------------------------------- LightGCN
def lightgcn_propagate(x0, adj, n_layer, trace=None):
    """`LightGCN.forward` (model/lightgcn.py:49-63) at dropout 0: the raw product
    feeds the next layer, its L2-normalised copy (eps 1e-12) goes into the mean,
    layer 0 enters the mean un-normalised."""
    x = x0
    layers = [x0]
    for _ in range(n_layer):
        x = split_mm(adj, x)
        z = F.normalize(x, p=2, dim=1)
        layers.append(z)
        if trace is not None:
            trace.append((x, z))
    return torch.stack(layers, dim=1).mean(dim=1)


def lightgcn_loss(tables, adj, n_layer, batch, reg, kind="softplus"):
    """`LightGCN.loss` (lightgcn.py:68-82): BPR on propagated rows, L2 on EGO rows.
    tables = [user, item(, tag)] tensors; batch LongTensor[B,3]."""
    nums = [t.shape[0] for t in tables]
    out = lightgcn_propagate(torch.cat(list(tables), dim=0), adj, n_layer)
    parts = torch.split(out, nums, dim=0)
    u, p, n = batch[:, 0], batch[:, 1], batch[:, 2]
    loss = mul_loss(parts[0][u], parts[1][p], parts[1][n], kind)
    regl = l2reg_loss(tables[0][u], tables[1][p], tables[1][n])
    return loss, reg * regl


# --------------------------------------------------------------------------- NGCF
def ngcf_propagate(x0, mats, adj, n_layer, trace=None):
    """`NGCF.bi_inter_embed` (model/ngcf.py:73-90).  Note `W + b`: the 1 x D_out bias
    is broadcast-added to the WEIGHT (ngcf.py:78,82); outputs are concatenated."""
    x = x0
    outs = [x0]
    for k in range(n_layer):
        nei = split_mm(adj, x)
        s = F.leaky_relu(torch.matmul(nei + x, mats[f"W1_{k}"] + mats[f"b1_{k}"]), 0.2)
        b = F.leaky_relu(torch.matmul(nei * x, mats[f"W2_{k}"] + mats[f"b2_{k}"]), 0.2)
        x = s + b
        z = F.normalize(x, p=2, dim=1)
        outs.append(z)
        if trace is not None:
            trace.append((nei, x, z))
    return torch.cat(outs, dim=1)


def ngcf_loss(tables, mats, adj, n_layer, batch, reg, kind="logsigmoid"):
    """`NGCF.loss` (ngcf.py:95-105): L2 on the PROPAGATED rows."""
    nums = [t.shape[0] for t in tables]
    out = ngcf_propagate(torch.cat(list(tables), dim=0), mats, adj, n_layer)
    parts = torch.split(out, nums, dim=0)
    u, p, n = batch[:, 0], batch[:, 1], batch[:, 2]
    ue, pe, ne = parts[0][u], parts[1][p], parts[1][n]
    return mul_loss(ue, pe, ne, kind), reg * l2reg_loss(ue, pe, ne)


# --------------------------------------------------------------------------- TGCN
def tgcn_attention1(prm, pre, ev, ej, ew, idx_j, idx_w):
    """`Attention1.forward` (model/tgcn.py:20-37).  Index 0 is the pad row (a zero
    vector is prepended to ej and ew); the softmax is NOT masked."""
    W1, W2, b, v = prm[pre + "W_1"], prm[pre + "W_2"], prm[pre + "b"], prm[pre + "v"]
    d = ev.shape[1]
    ejp = torch.cat([ej.new_zeros(1, ej.shape[1]), ej])
    ewp = torch.cat([ew.new_zeros(1, ew.shape[1]), ew])
    nj = ejp[idx_j]                                   # (N,k,D)
    nw = ewp[idx_w]                                   # (N,k,dw)
    # [ev || nw] W1 == ev W1[:D] + nw W1[D:]  (same sum, different association)
    cat = torch.cat([ev.unsqueeze(1).expand(-1, idx_j.shape[1], -1), nw], dim=-1)
    a = torch.matmul(cat, W1) + torch.matmul(nj, W2) + b
    s = torch.matmul(torch.relu(a), v.t())            # (N,k,1)
    w = torch.softmax(s, dim=1)
    return (w * nj).sum(dim=1)


def tgcn_atten2(prm, pre, u, i, t):
    """`BasicLayer._atten2` (tgcn.py:78-84): type-level attention, NOT summed."""
    st = torch.stack([u, i, t], dim=1)
    x = torch.matmul(torch.relu(torch.matmul(st, prm[pre + "U"]) + prm[pre + "q"]), prm[pre + "p"].t())
    return torch.softmax(x, dim=1) * st


def tgcn_conv(prm, pre, e3):
    """`BasicLayer._conv` (tgcn.py:86-101) written without Conv2d.
    bit-level Conv2d(1,C,(3,1)): out[n,c,d] = sum_j w[c,j] e3[n,j,d]  -> (N, C*D)
    vec-level Conv2d(1,V,(j,D)), j=1..3: out[n,c,h] = sum_{a<j,d} w[c,a,d] e3[n,h+a,d] -> (N, V*(4-j))"""
    wb = prm[pre + "conv.bit_level.weight"][:, 0, :, 0]                  # (C,3)
    bit = torch.relu(torch.einsum("cj,njd->ncd", wb, e3)).reshape(e3.shape[0], -1)
    vecs = []
    for j in (1, 2, 3):
        w = prm[pre + f"conv.vec_level.conv_{j}.weight"][:, 0]           # (V,j,D)
        win = torch.stack([e3[:, h:h + j, :] for h in range(4 - j)], dim=1)  # (N,4-j,j,D)
        y = torch.relu(torch.einsum("cad,nhad->nch", w, win))
        vecs.append(y.reshape(e3.shape[0], -1))
    return torch.cat([bit] + vecs, dim=1)


def tgcn_layer(prm, k, eu, ei, et, ew, nbr):
    """`BasicLayer.forward` (tgcn.py:108-137).  nbr = 6 pairs (idx, widx) in the order
    u<-i, u<-t, i<-u, i<-t, t<-u, t<-i; attention modules are shared by NEIGHBOUR type."""
    pre = f"layer.{k}."
    (u_i, u_t, i_u, i_t, t_u, t_i) = nbr
    a = lambda typ, ev, ej, pair: tgcn_attention1(prm, pre + f"atten1.{typ}.", ev, ej, ew, pair[0], pair[1])
    eu_i, eu_t = a("item", eu, ei, u_i), a("tag", eu, et, u_t)
    ei_u, ei_t = a("user", ei, eu, i_u), a("tag", ei, et, i_t)
    et_u, et_i = a("user", et, eu, t_u), a("item", et, ei, t_i)
    outs = []
    for trip in ((eu, eu_i, eu_t), (ei_u, ei, ei_t), (et_u, et_i, et)):
        c = tgcn_conv(prm, pre, tgcn_atten2(prm, pre, *trip))
        outs.append(torch.relu(torch.matmul(c, prm[pre + "Wf"]) + prm[pre + "bf"]))
    return outs


def tgcn_forward(prm, n_layer, nbr, trace=None):
    """`TGCN.forward` (tgcn.py:204-230) at dropout 0; `sample()` (:194-202) always
    returns the first neighbor_k columns, so `nbr` is fixed."""
    eu, ei, et, ew = prm["embed.user"], prm["embed.item"], prm["embed.tag"], prm["embed.weight"]
    cu, ci, ct = [eu], [ei], [et]
    for k in range(n_layer):
        eu, ei, et = tgcn_layer(prm, k, eu, ei, et, ew, nbr)
        if trace is not None:
            trace.append((eu, ei, et))
        cu.append(F.normalize(eu, p=2, dim=1))
        ci.append(F.normalize(ei, p=2, dim=1))
        ct.append(F.normalize(et, p=2, dim=1))
    return torch.cat(cu, 1), torch.cat(ci, 1), torch.cat(ct, 1)


def tgcn_loss(prm, n_layer, nbr, batch, reg, kind="logsigmoid"):
    """`TGCN.loss` (tgcn.py:235-249): reg on propagated rows."""
    au, ai, _ = tgcn_forward(prm, n_layer, nbr)
    ue, pe, ne = au[batch[:, 0]], ai[batch[:, 1]], ai[batch[:, 2]]
    return mul_loss(ue, pe, ne, kind), reg * l2reg_loss(ue, pe, ne)


def tgcn_transtag_loss(prm, batch, margin, treg):
    """`TGCN.transtag_loss` (tgcn.py:251-261): batch = (user, tag, pos_item, neg_item), EGO rows."""
    ue = prm["embed.user"][batch[:, 0]]
    te = prm["embed.tag"][batch[:, 1]]
    pe = prm["embed.item"][batch[:, 2]]
    ne = prm["embed.item"][batch[:, 3]]
    return transtag_loss(ue, te, pe, ne, margin), treg * l2reg_loss(ue, te, pe, ne)


# --------------------------------------------------------------------------- predict / step
# --------------------------------------------------------------------------- DGCF / DisenGCN
def _diag_inv_sqrt_rowsum(rows, a, n):
    """`torch.sparse.sum(adj, dim=1)` -> 1/sqrt, inf -> 0 (model/dgcf.py:96-99); rows without entries get 0
    (the reference's sparse diagonal has no entry there)."""
    rs = torch.zeros(n, dtype=a.dtype).index_add_(0, rows, a)
    d = 1.0 / torch.sqrt(rs)
    d[torch.isinf(d)] = 0.0
    return d


def dgcf_factor_update(rows, cols, a_factor, ego_split, n):
    """`DGCF.factor_update` (model/dgcf.py:92-110): D^-1/2 A_f D^-1/2 x with the (detached) routing weights as
    edge values, then the edge score  normalize(f[head]) . tanh(normalize(x[tail]))."""
    a = a_factor.detach()
    d = _diag_inv_sqrt_rowsum(rows, a, n)
    adj = torch.sparse_coo_tensor(torch.stack([rows, cols]), a, (n, n))
    f = d[:, None] * ego_split
    f = torch.sparse.mm(adj, f)
    f = d[:, None] * f
    h = F.normalize(f[rows], p=2, dim=1)
    t = F.normalize(ego_split[cols], p=2, dim=1)
    return a, f, torch.sum(h * torch.tanh(t), dim=1)


def dgcf_forward(tables, rows, cols, n_layer, factor_k, iterate_k, trace=None):
    """`DGCF.forward` + `iterate_update` (model/dgcf.py:51-90).  rows/cols = `norm_adj._indices()` (int64)."""
    ego = torch.cat(list(tables), dim=0)
    n, dk = ego.shape[0], ego.shape[1] // factor_k
    a_values = torch.ones(factor_k, rows.numel())
    all_embed = [ego]
    for _ in range(n_layer):
        split = torch.split(ego, dk, dim=1)
        layer_emb, layer_a = [], []
        for t in range(iterate_k):
            a_factor = torch.softmax(a_values, dim=0)
            scores = []
            for i in range(factor_k):
                a, f, sc = dgcf_factor_update(rows, cols, a_factor[i], split[i], n)
                scores.append(sc)
                if t == iterate_k - 1:
                    layer_emb.append(f)
                    layer_a.append(a)
            a_values = a_values + torch.stack(scores, dim=0)
        layer_emb = F.normalize(torch.stack(layer_emb), p=2, dim=2)
        ego = torch.cat(list(layer_emb), dim=1)
        all_embed.append(ego)
        if trace is not None:
            trace.append(torch.stack(layer_a).detach())
    out = torch.mean(torch.stack(all_embed, dim=1), dim=1)
    return torch.split(out, [t.shape[0] for t in tables], dim=0)


def dgcf_loss(tables, rows, cols, n_layer, factor_k, iterate_k, batch, reg, kind="softplus"):
    """`DGCF.loss` (model/dgcf.py:115-145): BPR on propagated rows, L2 on the EGO rows; the `cor` half of the
    batch is unused (the correlation loss is commented out in the reference)."""
    outs = dgcf_forward(tables, rows, cols, n_layer, factor_k, iterate_k)
    u, p, ng = batch[:, 0], batch[:, 1], batch[:, 2]
    loss = mul_loss(outs[0][u], outs[1][p], outs[1][ng], kind)
    return loss, reg * l2reg_loss(tables[0][u], tables[1][p], tables[1][ng])


def disengcn_layer(W, b, rows, cols, x, factor_k, iterate_k, n):
    """`disengcn.Layer.forward` (model/disengcn.py:23-46): per-factor projection `x (W + b)`, LeakyReLU 0.2,
    normalise; then `iterate_k` rounds of neighbourhood routing: p = softmax over factors of
    <new_f[head], f[tail]>, f_i + A(p_i) f_i, normalise (p detached as edge values)."""
    f = F.normalize(F.leaky_relu(torch.matmul(x, W + b), 0.2), p=2, dim=2)           # [K, n, dk]
    new_f = f
    idx = torch.stack([rows, cols])
    for _ in range(iterate_k):
        p_uv = torch.softmax(torch.sum(new_f[:, rows] * f[:, cols], dim=2), dim=0)  # [K, nnz]
        embs = []
        for i in range(factor_k):
            adj = torch.sparse_coo_tensor(idx, p_uv[i].detach(), (n, n))
            embs.append(F.normalize(f[i] + torch.sparse.mm(adj, f[i]), p=2, dim=1))
        new_f = torch.stack(embs)
    return torch.cat(list(new_f), dim=1)


def disengcn_forward(tables, layers, rows, cols, factor_k, iterate_k):
    """`DisenGCN.forward` (model/disengcn.py:90-103): only the LAST layer's output is returned (the layer list /
    mean are commented out in the reference).  layers = [(W [K,D,dk], b [K,1,dk]), ...]."""
    x = torch.cat(list(tables), dim=0)
    for W, b in layers:
        x = disengcn_layer(W, b, rows, cols, x, factor_k, iterate_k, x.shape[0])
    return torch.split(x, [t.shape[0] for t in tables], dim=0)


def disengcn_loss(tables, layers, rows, cols, factor_k, iterate_k, batch, reg, kind="softplus"):
    """`DisenGCN.loss` (model/disengcn.py:105-131): L2 term on the PROPAGATED rows."""
    outs = disengcn_forward(tables, layers, rows, cols, factor_k, iterate_k)
    u, p, ng = batch[:, 0], batch[:, 1], batch[:, 2]
    ue, pe, ne = outs[0][u], outs[1][p], outs[1][ng]
    return mul_loss(ue, pe, ne, kind), reg * l2reg_loss(ue, pe, ne)


# --------------------------------------------------------------------------- KGAT
def kgat_attention(all_embed, trans_e, relation, edges):
    """Attention logits and the row softmax of `KGAT.forward` (model/kgat.py:63-104, split_adj_k == 1 branch):
    per relation k:  pai = sum((e_tail W_k) * tanh(e_head W_k + r_k)); the edge lists are concatenated into ONE
    sparse tensor and `torch.sparse.softmax(dim=1)` normalises each row (duplicate entries are summed by its
    coalesce first).  `edges[k]` is a 2-D integer tensor read exactly as the reference reads it -- head =
    `e[:, 0]`, tail = `e[:, 1]` (kgat.py:71-72) -- so an [E, 2] array (the layout of
    `KGAT_load.get_relation_dict`, data/kgat_load.py:53-63) gives E edges, while the [2, E] arrays of
    `TGCN_load.create_edge` (data/tgcn_load.py:55-71), which is what com.py:78-79 passes in, give two."""
    pai, rows, cols = [], [], []
    for k in sorted(edges.keys()):
        row, col = edges[k][:, 0].long(), edges[k][:, 1].long()
        tr = torch.matmul(all_embed[col], trans_e[k])
        hr = torch.matmul(all_embed[row], trans_e[k]) + relation[k]
        pai.append(torch.sum(tr * torch.tanh(hr), dim=1))
        rows.append(row)
        cols.append(col)
    n = all_embed.shape[0]
    adj = torch.sparse_coo_tensor(torch.stack([torch.cat(rows), torch.cat(cols)]), torch.cat(pai), (n, n))
    return torch.sparse.softmax(adj, dim=1)


def kgat_forward(prm, edges, n_user, n_layer, agg_type="bi_inter"):
    """`KGAT.forward` + `bi_inter_embed` (kgat.py:63-126).  prm keys as the reference's state_dict
    (`embed.user`, `embed.entity`, `embed.relation`, `mat.transE`, `mat.W1_k`, ...).  With any `agg_type` other
    than "bi_inter" -- including the reference's own default "bi_agg" -- no propagation happens (kgat.py:100-101)."""
    all_embed = torch.cat([prm["embed.user"], prm["embed.entity"]], dim=0)
    if agg_type == "bi_inter":
        adj = kgat_attention(all_embed, prm["mat.transE"], prm["embed.relation"], edges)
        outs = [all_embed]
        for k in range(n_layer):
            nei = torch.sparse.mm(adj, all_embed)
            s = F.leaky_relu(torch.matmul(nei + all_embed, prm[f"mat.W1_{k}"] + prm[f"mat.b1_{k}"]), 0.2)
            b = F.leaky_relu(torch.matmul(nei * all_embed, prm[f"mat.W2_{k}"] + prm[f"mat.b2_{k}"]), 0.2)
            all_embed = s + b
            outs.append(F.normalize(all_embed, p=2, dim=1))
        all_embed = torch.cat(outs, dim=1)
    return all_embed[:n_user], all_embed[n_user:]


def kgat_loss(prm, edges, n_user, n_layer, batch, reg, agg_type="bi_inter", kind="softplus"):
    """`KGAT.loss` (kgat.py:143-153): items index the ENTITY table; L2 on the propagated rows."""
    users, ents = kgat_forward(prm, edges, n_user, n_layer, agg_type)
    ue, pe, ne = users[batch[:, 0]], ents[batch[:, 1]], ents[batch[:, 2]]
    return mul_loss(ue, pe, ne, kind), reg * l2reg_loss(ue, pe, ne)


def kgat_transe_loss(prm, batch, cor_reg):
    """`KGAT.get_embed` + `transe_loss` (kgat.py:128-162): rows (head, relation, pos tail, neg tail)."""
    all_embed = torch.cat([prm["embed.user"], prm["embed.entity"]], dim=0)
    head, rela, pt, nt = batch[:, 0], batch[:, 1], batch[:, 2], batch[:, 3]
    r_e = prm["embed.relation"][rela]
    W = prm["mat.transE"][rela]
    h_e = torch.matmul(all_embed[head].unsqueeze(1), W).squeeze()
    p_e = torch.matmul(all_embed[pt].unsqueeze(1), W).squeeze()
    n_e = torch.matmul(all_embed[nt].unsqueeze(1), W).squeeze()
    pos = torch.norm(h_e + r_e - p_e, p=2, dim=1).pow(2)
    neg = torch.norm(h_e + r_e - n_e, p=2, dim=1).pow(2)
    return torch.mean(F.softplus(pos - neg)), cor_reg * l2reg_loss(h_e, r_e, p_e, n_e)


def predict_rating(user_out, item_out, users):
    """`predict_rating` (lightgcn.py:84-89): sigmoid(U_b I^T)."""
    return torch.sigmoid(torch.matmul(user_out[users], item_out.t()))


def adam_epoch(params, loss_fn, batches, opt):
    """`epoch_training` (training/basic_train.py:10-30) for one producer:
    per batch -> loss parts to host floats, sum, zero_grad/backward/step.
    Returns (list of per-batch totals, list of per-batch parts)."""
    totals, parts = [], []
    for b in batches:
        lx = loss_fn(b)
        parts.append([float(x.detach()) for x in lx])
        tot = sum(lx)
        opt.zero_grad()
        tot.backward()
        opt.step()
        totals.append(float(tot.detach()))
    return totals, parts
