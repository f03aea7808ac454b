"""Operator surface of /root/reference/model/help (adj.py, loss.py), same names and
argument meaning, backed by the HIP library through `torch.autograd.Function`s.

    split_mm(norm_adj, all_embed)            adj.py:158-167
    node_drop(graph, keep_prob, training)    adj.py:170-191
    mul_loss(u, p, n, loss_func)             loss.py:4-12
    l2reg_loss(*embs)                        loss.py:27-32
    transtag_loss / transe_loss              loss.py:35-50
    creat_adj(...)                           adj.py:38-46   (re-exported from graph.py)

Inputs must be GPU tensors; there is no CPU path.
"""
import torch

from . import _lib
from .graph import Graph, coalesce_device, creat_adj  # noqa: F401  (creat_adj re-exported)


def _new_like_rows(g, D, ref):
    return torch.empty(g.shape[0], D, dtype=torch.float32, device=ref.device)


class _SpMM(torch.autograd.Function):
    """Y = A @ X; backward dX = A^T @ dY (A carries no gradient, as in the reference
    where norm_adj is a constant sparse tensor)."""

    @staticmethod
    def forward(ctx, X, graph):
        ctx.graph = graph
        return graph.spmm(X.contiguous())

    @staticmethod
    def backward(ctx, dY):
        return ctx.graph.transpose().spmm(dY.contiguous()), None


def split_mm(norm_adj, all_embed):
    """One `Graph`, or a list of row-fold Graphs whose products are concatenated
    along dim 0 -- the reference's `split_adj_k` memory workaround."""
    if isinstance(norm_adj, (list, tuple)):
        return torch.cat([_SpMM.apply(all_embed, g) for g in norm_adj], dim=0)
    return _SpMM.apply(all_embed, norm_adj)


class _RowNormalize(torch.autograd.Function):
    """F.normalize(x, p=2, dim=1) (eps 1e-12) with its analytic backward."""

    @staticmethod
    def forward(ctx, X):
        X = _lib.require_gpu_tensor(X.contiguous(), torch.float32, "normalize_rows input")
        n, D = X.shape
        Z = torch.empty_like(X)
        inv = torch.empty(n, dtype=torch.float32, device=X.device)
        _lib.check(_lib.load().tagrec_rownorm_fwd_f32(_lib.ptr(X), _lib.ptr(Z), D, _lib.ptr(inv), n, D,
                                                      _lib.stream_ptr()), "rownorm_fwd")
        ctx.save_for_backward(X, inv)
        return Z

    @staticmethod
    def backward(ctx, dZ):
        X, inv = ctx.saved_tensors
        dZ = dZ.contiguous()
        n, D = X.shape
        dX = torch.empty_like(X)
        _lib.check(_lib.load().tagrec_rownorm_bwd_f32(_lib.ptr(X), _lib.ptr(inv), _lib.ptr(dZ), D, 1.0, _lib.ptr(dX),
                                                      0, n, D, _lib.stream_ptr()), "rownorm_bwd")
        return dX


def normalize_rows(x):
    return _RowNormalize.apply(x)


class _TripletLoss(torch.autograd.Function):
    """BPR loss on tables + triplet indices (the gather is part of the kernel).
    Returns a 2-vector: [mul_loss, l2reg_loss (unweighted)]."""

    @staticmethod
    def forward(ctx, U, I, Ureg, Ireg, trip, loss_kind):
        lib = _lib.load()
        for t, nm in ((U, "U"), (I, "I")):
            _lib.require_gpu_tensor(t, torch.float32, "bpr " + nm)
        trip = _lib.require_gpu_tensor(trip.contiguous(), torch.int64, "bpr triplets")
        B, D = trip.shape[0], U.shape[1]
        has_reg = Ureg is not None
        coef = torch.empty(B, dtype=torch.float32, device=U.device)
        partials = torch.empty(2 * ((B + 3) // 4), dtype=torch.float32, device=U.device)
        out = torch.empty(2, dtype=torch.float32, device=U.device)
        _lib.check(lib.tagrec_bpr_fwd_f32(_lib.ptr(U), _lib.ptr(I), U.stride(0), D,
                                          _lib.ptr(Ureg), _lib.ptr(Ireg), Ureg.stride(0) if has_reg else 0,
                                          Ureg.shape[1] if has_reg else 0, _lib.ptr(trip), B, loss_kind,
                                          _lib.ptr(coef), _lib.ptr(partials), _lib.ptr(out), _lib.stream_ptr()),
                   "bpr_fwd")
        ctx.save_for_backward(U, I, Ureg if has_reg else U.new_empty(0), Ireg if has_reg else U.new_empty(0), trip, coef)
        ctx.has_reg = has_reg
        ctx.same = has_reg and Ureg.data_ptr() == U.data_ptr() and Ireg.data_ptr() == I.data_ptr()
        return out

    @staticmethod
    def backward(ctx, g):
        U, I, Ureg, Ireg, trip, coef = ctx.saved_tensors
        g = g.contiguous()
        dU, dI = torch.zeros_like(U), torch.zeros_like(I)
        if ctx.has_reg and not ctx.same:
            dUr, dIr = torch.zeros_like(Ureg), torch.zeros_like(Ireg)
        elif ctx.has_reg:
            dUr, dIr = dU, dI
        else:
            dUr = dIr = None
        _lib.check(_lib.load().tagrec_bpr_bwd_f32(
            _lib.ptr(U), _lib.ptr(I), U.stride(0), U.shape[1],
            _lib.ptr(Ureg if ctx.has_reg else None), _lib.ptr(Ireg if ctx.has_reg else None),
            Ureg.stride(0) if ctx.has_reg else 0, Ureg.shape[1] if ctx.has_reg else 0,
            _lib.ptr(trip), trip.shape[0], _lib.ptr(coef), _lib.ptr(g), 1.0,
            _lib.ptr(dU), _lib.ptr(dI), _lib.ptr(dUr), _lib.ptr(dIr), _lib.stream_ptr()), "bpr_bwd")
        if ctx.has_reg and ctx.same:
            return dU, dI, None, None, None, None
        return dU, dI, (dUr if ctx.has_reg else None), (dIr if ctx.has_reg else None), None, None


def loss_kind_id(loss_func):
    return _lib.LOSS_LOGSIGMOID if loss_func == "logsigmoid" else _lib.LOSS_SOFTPLUS


def triplet_loss(U, I, Ureg, Ireg, trip, loss_func):
    """(mul_loss, l2reg_loss) of a [B,3] triplet batch against user/item tables."""
    if Ureg is not None and Ureg.data_ptr() == U.data_ptr() and Ireg.data_ptr() == I.data_ptr():
        Ureg, Ireg = U, I          # reg on the same rows: one gradient buffer
    out = _TripletLoss.apply(U, I, Ureg, Ireg, trip, loss_kind_id(loss_func))
    return out[0], out[1]


def mul_loss(users_emb, pos_emb, neg_emb, loss_func):
    """`mul_loss` on already gathered rows (loss.py:4-12)."""
    B = users_emb.shape[0]
    ar = torch.arange(B, device=users_emb.device)
    trip = torch.stack([ar, ar, ar + B], dim=1)
    items = torch.cat([pos_emb, neg_emb], dim=0)
    return _TripletLoss.apply(users_emb.contiguous(), items, None, None, trip, loss_kind_id(loss_func))[0]


def l2reg_loss(*embs):
    """0.5 * sum ||e||_F^2 / rows(first)  (loss.py:27-32); torch reductions on the GPU."""
    for e in embs:
        if not e.is_cuda:
            raise _lib.TagrecError("l2reg_loss: expected GPU tensors")
    tot = 0
    for e in embs:
        tot = tot + e.norm(2).pow(2)
    return 0.5 * tot / float(embs[0].shape[0])


class _TransTagLoss(torch.autograd.Function):
    """TransTag margin loss + L2 on table rows selected by a [B,4] (user, tag, pos_item, neg_item) batch;
    the gathers are part of the kernel.  Returns [loss, l2reg_loss (unweighted)]."""

    @staticmethod
    def forward(ctx, Eu, Ei, Et, quad, margin):
        for t, nm in ((Eu, "Eu"), (Ei, "Ei"), (Et, "Et")):
            _lib.require_gpu_tensor(t, torch.float32, "transtag " + nm)
        quad = _lib.require_gpu_tensor(quad.contiguous(), torch.int64, "transtag batch")
        B, D = quad.shape[0], Eu.shape[1]
        dist = torch.empty(B, 2, dtype=torch.float32, device=Eu.device)
        partials = torch.empty(2 * ((B + 3) // 4), dtype=torch.float32, device=Eu.device)
        out = torch.empty(2, dtype=torch.float32, device=Eu.device)
        _lib.check(_lib.load().tagrec_transtag_fwd_f32(_lib.ptr(Eu), _lib.ptr(Ei), _lib.ptr(Et), Eu.stride(0), D,
                                                       _lib.ptr(quad), B, float(margin), _lib.ptr(dist), _lib.ptr(partials),
                                                       _lib.ptr(out), _lib.stream_ptr()), "transtag_fwd")
        ctx.save_for_backward(Eu, Ei, Et, quad, dist)
        ctx.margin = float(margin)
        return out

    @staticmethod
    def backward(ctx, g):
        Eu, Ei, Et, quad, dist = ctx.saved_tensors
        dEu, dEi, dEt = torch.zeros_like(Eu), torch.zeros_like(Ei), torch.zeros_like(Et)
        _lib.check(_lib.load().tagrec_transtag_bwd_f32(_lib.ptr(Eu), _lib.ptr(Ei), _lib.ptr(Et), Eu.stride(0), Eu.shape[1],
                                                       _lib.ptr(quad), quad.shape[0], ctx.margin, _lib.ptr(dist),
                                                       _lib.ptr(g.contiguous()), _lib.ptr(dEu), _lib.ptr(dEi), _lib.ptr(dEt),
                                                       _lib.stream_ptr()), "transtag_bwd")
        return dEu, dEi, dEt, None, None


def transtag_batch_loss(Eu, Ei, Et, quad, margin):
    """(transtag_loss, l2reg_loss) of a [B,4] batch against the user / item / tag tables (tgcn.py:251-261)."""
    if not (Eu.stride(0) == Ei.stride(0) == Et.stride(0) and Eu.shape[1] == Ei.shape[1] == Et.shape[1]):
        raise _lib.TagrecError("transtag_batch_loss: tables must share width and row stride")
    out = _TransTagLoss.apply(Eu, Ei, Et, quad, margin)
    return out[0], out[1]


def transtag_loss(head_e, rela_e, pos_tail_e, neg_tail_e, margin=0):
    """mean relu(margin + ||h+r-t+|| - ||h+r-t-||)  (loss.py:35-41)."""
    if not head_e.is_cuda:
        raise _lib.TagrecError("transtag_loss: expected GPU tensors")
    ps = torch.norm(head_e + rela_e - pos_tail_e, p=2, dim=1)
    ns = torch.norm(head_e + rela_e - neg_tail_e, p=2, dim=1)
    return torch.relu(margin + ps - ns).mean()


def transe_loss(head_e, rela_e, pos_tail_e, neg_tail_e):
    """mean softplus(||h+r-t+|| - ||h+r-t-||)  (loss.py:44-50)."""
    ps = torch.norm(head_e + rela_e - pos_tail_e, p=2, dim=1)
    ns = torch.norm(head_e + rela_e - neg_tail_e, p=2, dim=1)
    return torch.nn.functional.softplus(ps - ns).mean()


def message_drop(x, p, seed, out=None, rows=None):
    """mask(seed) * x / (1 - p) with the library's counter-based mask (the one the fused layer kernels apply);
    p = 0 returns x.  rows (int64 [T], x is [T, d]): x holds rows `rows` of a full [N, d] tensor and element (j, c) takes
    the draw of element (rows[j], c) of that tensor -- repeated ids get the same mask."""
    if p <= 0:
        return x
    x = x.contiguous()
    out = torch.empty_like(x) if out is None else out
    if rows is not None:
        _lib.check(_lib.load().tagrec_dropout_rows_f32(_lib.ptr(x), _lib.ptr(out), _lib.ptr(rows), x.shape[0], x.shape[1],
                                                       float(p), int(seed), _lib.stream_ptr()), "dropout_rows")
        return out
    _lib.check(_lib.load().tagrec_dropout_f32(_lib.ptr(x), _lib.ptr(out), x.numel(), float(p), int(seed), _lib.stream_ptr()),
               "dropout")
    return out


class _MessageDrop(torch.autograd.Function):
    """`message_drop` with its backward (the same mask on the gradient)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return message_drop(x, p, seed)

    @staticmethod
    def backward(ctx, g):
        return message_drop(g.contiguous(), ctx.p, ctx.seed), None, None


def message_dropout(x, p, seed):
    """F.dropout(x, p, training=True) with the library's counter-based mask (a function of seed and element index) instead
    of torch's generator: differentiable, and reproducible by every other path that is handed the same seed."""
    return x if p <= 0 else _MessageDrop.apply(x, float(p), int(seed))


def node_drop(graph, keep_prob, training=False):
    """Edge dropout (adj.py:170-191).  As in the reference the argument called
    `keep_prob` is the DROP rate: an edge survives iff int(rand + (1-drop)) != 0
    and survivors are divided by (1-drop).  The mask is drawn on the GPU (the
    reference draws it on the CPU), so parity is statistical."""
    assert 0 <= keep_prob < 1
    if keep_prob == 0 or not training:
        return graph
    keep = 1.0 - keep_prob

    def drop(g):
        mask = (torch.rand(g.nnz, device=g.device) + keep).to(torch.int32).bool()
        deg = g.rowptr[1:] - g.rowptr[:-1]
        rows = torch.repeat_interleave(torch.arange(g.shape[0], device=g.device), deg)[mask]
        rp, c, v = coalesce_device(rows, g.col.long()[mask], g.val[mask] / keep, g.shape[0], g.shape[1])
        return Graph(rp, c, v, g.shape)          # the mask breaks symmetry: transpose is rebuilt

    if isinstance(graph, (list, tuple)):
        return [drop(g) for g in graph]
    return drop(graph)
