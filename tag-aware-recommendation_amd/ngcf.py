"""NGCF behind the reference's model surface (/root/reference/model/ngcf.py).

    NGCF(data)               embed.* tables + mat.W1_k / b1_k / W2_k / b2_k                      (:9-60)
    .forward()               -> tuple of [n_type, sum(dims)] concatenated layer outputs          (:62-90)
    .loss(batch[B,3])        -> (mul_loss, reg * l2reg_loss on the PROPAGATED rows)              (:95-105)
    .predict_rating(users)                                                                       (:107-112)

Per layer: one HIP SpMM (N = A X, A = D^-1 A + I is NOT symmetric, so backward multiplies by the
transposed CSR), one MFMA kernel for both dense transforms + LeakyReLU + add + L2-normalise that
writes its slot of the concatenated output in place, and a hand-derived backward
(normalise-bwd -> activation/dA MFMA kernel -> weight-gradient MFMA kernel -> transposed SpMM).
The reference's `W + b` quirk (bias broadcast-added to the WEIGHT, :78,:82) is kept: W' = W + b is
formed on the device, dW = dW', db = column sums of dW'.
"""
import torch
import torch.nn as nn

from . import _lib, help as H
from .base import TableModel
from .config import CFG as _GLOBAL_CFG
from .graph import Graph, creat_adj


def dense_forward(nei, x, w1p, w2p, xp, inv, z_slot, ldz):
    n, din = x.shape
    _lib.check(_lib.load().tagrec_ngcf_dense_fwd_f32(_lib.ptr(nei), _lib.ptr(x), _lib.ptr(w1p), _lib.ptr(w2p), n, din,
                                                     w1p.shape[1], _lib.ptr(xp), _lib.ptr(inv), _lib.ptr(z_slot), ldz,
                                                     _lib.stream_ptr()), "ngcf_dense_fwd")


def dense_backward(dxp, nei, x, w1p, w2p, norm=None):
    """norm = (xp, inv, dz, ldz): the gradient w.r.t. Xp is dxp (may be None) + normalize-backward of the layer's slot
    dz of the concat gradient, formed inside the kernel."""
    n, din = x.shape
    dout = w1p.shape[1]
    d_nei, d_xd = torch.empty_like(x), torch.empty_like(x)
    dp1, dp2 = torch.empty(n, dout, device=x.device), torch.empty(n, dout, device=x.device)
    lib = _lib.load()
    if norm is not None:
        xp, inv, dz, ldz = norm
        _lib.check(lib.tagrec_ngcf_dense_bwd_norm_f32(_lib.ptr(dxp), _lib.ptr(xp), _lib.ptr(inv), _lib.ptr(dz), ldz, _lib.ptr(nei),
                                                      _lib.ptr(x), _lib.ptr(w1p), _lib.ptr(w2p), n, din, dout, _lib.ptr(d_nei),
                                                      _lib.ptr(d_xd), _lib.ptr(dp1), _lib.ptr(dp2), _lib.stream_ptr()),
                   "ngcf_dense_bwd_norm")
    else:
        _lib.check(lib.tagrec_ngcf_dense_bwd_f32(_lib.ptr(dxp), _lib.ptr(nei), _lib.ptr(x), _lib.ptr(w1p), _lib.ptr(w2p), n,
                                                 din, dout, _lib.ptr(d_nei), _lib.ptr(d_xd), _lib.ptr(dp1), _lib.ptr(dp2),
                                                 _lib.stream_ptr()), "ngcf_dense_bwd")
    ws_n = lib.tagrec_ngcf_wgrad_workspace(din, dout)
    ws = torch.empty(ws_n, dtype=torch.float32, device=x.device)
    dw1, dw2 = torch.empty_like(w1p), torch.empty_like(w2p)
    _lib.check(lib.tagrec_ngcf_wgrad_f32(_lib.ptr(nei), _lib.ptr(x), _lib.ptr(dp1), _lib.ptr(dp2), n, din, dout,
                                         _lib.ptr(dw1), _lib.ptr(dw2), _lib.ptr(ws), ws_n, _lib.stream_ptr()),
               "ngcf_wgrad")
    return d_nei, d_xd, dw1, dw2


RESTRICT_FORWARD = True     # loss(): form the top two layers' neighbour sums only on the rows the batch's loss depends on


def _layer_seed(seed, k):
    return (int(seed) * 64 + k) & 0xFFFFFFFFFFFFFFFF


def _drop_renorm(xp, inv, z_slot, ldz, p, seed):
    """Message dropout of a layer's output (ngcf.py:85) behind the fused dense kernel: Xp <- mask(seed) Xp / (1 - p) with the
    library's counter-based mask, then the slot of the concatenated output and the inverse norms are recomputed from it
    (two element-wise passes; the MFMA kernels stay in use)."""
    H.message_drop(xp, p, seed, out=xp)
    n, d = xp.shape
    _lib.check(_lib.load().tagrec_rownorm_fwd_f32(_lib.ptr(xp), _lib.ptr(z_slot), ldz, _lib.ptr(inv), n, d, _lib.stream_ptr()),
               "rownorm_fwd")


def _dxp_through_dropout(dx_next, xp, inv, dz, ldz, p, seed):
    """d loss / d (pre-dropout Xp) = mask / (1 - p) * (dx_next + normalize-backward(Xp_dropped, inv, dz))."""
    n, d = xp.shape
    g = torch.zeros_like(xp) if dx_next is None else dx_next.contiguous().clone()
    _lib.check(_lib.load().tagrec_rownorm_bwd_f32(_lib.ptr(xp), _lib.ptr(inv), _lib.ptr(dz), ldz, 1.0, _lib.ptr(g), 1, n, d,
                                                  _lib.stream_ptr()), "rownorm_bwd")
    return H.message_drop(g, p, seed, out=g)


def propagate_forward(graph, x0, wps, dims, loss_rows=None, drops=None, seed=0):
    """x0 [N, dims[0]] -> out [N, sum(dims)] = cat(x0, z1..zL) and the per-layer state for backward.

    loss_rows (int64 node ids): `out` will be read at these rows only (the batch rows).  Then the last layer's
    neighbour sum is formed for them alone and the one below it for their neighbours (marked through the adjacency
    rows; further down every row is needed); the other rows of those layers carry values that nothing reads."""
    n = x0.shape[0]
    dtot = sum(dims)
    out = torch.empty(n, dtot, dtype=torch.float32, device=x0.device)
    out[:, :dims[0]] = x0
    saved, x, off = [], x0, dims[0]
    L = len(wps)
    masks = {}
    if (loss_rows is not None and L >= 1 and loss_rows.numel() * 16 <= n and graph.shape[0] == graph.shape[1]
            and all(d in (8, 16, 32, 64, 128, 256) for d in dims[:-1])):
        top = torch.zeros(n, dtype=torch.uint8, device=x0.device)
        top.index_fill_(0, loss_rows, 1)
        masks[L - 1] = top
        if L >= 2:
            masks[L - 2] = graph.mark_rows(loss_rows, torch.zeros_like(top))
    for k, (w1p, w2p) in enumerate(wps):
        nei = graph.spmm_rows(x, torch.zeros_like(x), masks[k]) if k in masks else graph.spmm(x)
        if k in masks and k == L - 1:
            # top layer: the dense block runs on the batch rows alone.  A node named by several batch slots is computed
            # once per slot (identical values) and only its FIRST slot carries the upstream gradient in the backward
            # pass, so the row list keeps its static length 3 B and nothing is read back to the host.
            rows, _ = torch.sort(loss_rows)
            first = torch.ones_like(rows, dtype=torch.bool)
            first[1:] = rows[1:] != rows[:-1]
            xc, nc = x.index_select(0, rows), nei.index_select(0, rows)
            d = dims[k + 1]
            xpc = torch.empty(rows.numel(), d, dtype=torch.float32, device=x0.device)
            invc = torch.empty(rows.numel(), dtype=torch.float32, device=x0.device)
            zc = torch.empty(rows.numel(), d, dtype=torch.float32, device=x0.device)
            dense_forward(nc, xc, w1p, w2p, xpc, invc, zc, d)
            pk = drops[k] if drops else 0.0
            if pk > 0:                                             # (the mask is indexed by the position in this row list)
                _drop_renorm(xpc, invc, zc, d, pk, _layer_seed(seed, k))
            out[:, off:off + d].index_copy_(0, rows, zc)          # the other rows of this slot are never read
            saved.append(("rows", (rows, first), masks[k], masks.get(k - 1), xc, nc, xpc, invc, w1p, w2p, pk, _layer_seed(seed, k)))
            break
        xp = torch.empty(n, dims[k + 1], dtype=torch.float32, device=x0.device)
        inv = torch.empty(n, dtype=torch.float32, device=x0.device)
        dense_forward(nei, x, w1p, w2p, xp, inv, out[:, off:], dtot)
        pk = drops[k] if drops else 0.0
        if pk > 0:
            _drop_renorm(xp, inv, out[:, off:], dtot, pk, _layer_seed(seed, k))
        saved.append((x, nei, xp, inv, w1p, w2p, masks.get(k), bool(masks), pk, _layer_seed(seed, k)))
        x, off = xp, off + dims[k + 1]
    return out, saved


def propagate_backward(graph_t, d_out, saved, dims):
    """d_out [N, sum(dims)] -> (d_x0, [(dW1', dW2')] per layer)."""
    n, dtot = d_out.shape
    lib = _lib.load()
    offs = [0]
    for d in dims:
        offs.append(offs[-1] + d)
    dws = [None] * len(saved)
    dx_next = None
    for k in range(len(saved) - 1, -1, -1):
        if isinstance(saved[k][0], str):                       # the top layer of a restricted forward pass: batch rows only
            _, (rows, first), mask, reach, xc, nc, xpc, invc, w1p, w2p, pk, sk = saved[k]
            d = dims[k + 1]
            dzc = d_out[:, offs[k + 1]:offs[k + 1] + d].index_select(0, rows) * first[:, None]     # one slot per node
            if pk > 0:
                d_nei_c, d_xd_c, dw1, dw2 = dense_backward(_dxp_through_dropout(None, xpc, invc, dzc, d, pk, sk), nc, xc, w1p, w2p)
            else:
                d_nei_c, d_xd_c, dw1, dw2 = dense_backward(None, nc, xc, w1p, w2p, norm=(xpc, invc, dzc, d))
            dws[k] = (dw1, dw2)
            din = xc.shape[1]
            d_nei = torch.zeros(n, din, dtype=torch.float32, device=xc.device).index_add_(0, rows, d_nei_c)
            d_xd = torch.zeros(n, din, dtype=torch.float32, device=xc.device).index_add_(0, rows, d_xd_c)
            count = torch.full((1,), rows.numel(), dtype=torch.int32, device=xc.device)
            # d_nei and d_xd live on the batch rows, so dx is zero outside `reach` = those rows and their neighbours (the
            # mask the layer below was computed on): only they are visited
            dx = (torch.empty if reach is None else torch.zeros)(n, din, dtype=torch.float32, device=xc.device)
            graph_t.spmm_axpy_sparse(d_nei, mask, count, d_xd, 1.0, dx, reach)
            dx_next = dx
            continue
        x, nei, xp, inv, w1p, w2p, mask_k, restricted, pk, sk = saved[k]
        # d Xp = (what layer k+1 sent back) + normalize-backward of this layer's concat slot, formed inside the kernel
        # (with message dropout: formed outside, masked, and handed to the kernel as dXp)
        if pk > 0:
            d_nei, d_xd, dw1, dw2 = dense_backward(_dxp_through_dropout(dx_next, xp, inv, d_out[:, offs[k + 1]:], dtot, pk, sk),
                                                   nei, x, w1p, w2p)
        else:
            d_nei, d_xd, dw1, dw2 = dense_backward(dx_next, nei, x, w1p, w2p, norm=(xp, inv, d_out[:, offs[k + 1]:], dtot))
        dws[k] = (dw1, dw2)
        dx = torch.empty_like(x)
        if restricted and x.shape[1] in (8, 16, 32, 64, 128, 256):
            # restricted forward: where d_nei can be non-zero is known without looking at it.  The layer computed on a
            # row mask receives gradient on that mask only (the slot's dz lives on the batch rows, the layer above sends
            # back onto `reach` = this mask): the mask serves as the row flags.  Below that nearly every row is reached
            # through the popular items, and the plain product is used.
            if mask_k is not None:
                graph_t.spmm_axpy_sparse(d_nei, mask_k, None, d_xd, 1.0, dx)
            else:
                graph_t.spmm_axpy(d_nei, d_xd, 1.0, dx)
        elif x.shape[1] in (8, 16, 32, 64, 128, 256):
            # d_nei is non-zero on the rows the batch gradient has reached so far (the batch rows in the last layer,
            # their neighbours one layer down): the product does not fetch the rows flagged zero (same result)
            flags = torch.empty(n, dtype=torch.uint8, device=x.device)
            count = torch.zeros(1, dtype=torch.int32, device=x.device)
            _lib.check(lib.tagrec_row_flags_f32(_lib.ptr(d_nei), n, x.shape[1], _lib.ptr(flags), _lib.ptr(count),
                                                _lib.stream_ptr()), "row_flags")
            graph_t.spmm_axpy_sparse(d_nei, flags, count, d_xd, 1.0, dx)
        else:
            graph_t.spmm_axpy(d_nei, d_xd, 1.0, dx)
        dx_next = dx
    d0 = d_out[:, :dims[0]]
    return (dx_next + d0) if dx_next is not None else d0.contiguous(), dws


def _mat_grads(dws):
    out = []
    for dw1, dw2 in dws:                  # order W1_k, b1_k, W2_k, b2_k (the ParameterDict's order)
        out += [dw1, dw1.sum(0, keepdim=True), dw2, dw2.sum(0, keepdim=True)]
    return out


def _wps(mats):
    return [(mats[4 * k] + mats[4 * k + 1], mats[4 * k + 2] + mats[4 * k + 3]) for k in range(len(mats) // 4)]


class _Propagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, dims, drops, seed, table, *mats):
        out, saved = propagate_forward(graph, table.detach(), _wps([m.detach() for m in mats]), dims, None, drops, seed)
        ctx.graph, ctx.dims, ctx.saved = graph, dims, saved
        return out

    @staticmethod
    def backward(ctx, d_out):
        d0, dws = propagate_backward(ctx.graph.transpose(), d_out.contiguous(), ctx.saved, ctx.dims)
        ctx.saved = None
        return (None, None, None, None, d0, *_mat_grads(dws))


class _PropagateBprLoss(torch.autograd.Function):
    """(table, mats) -> [mul_loss, l2reg_loss(propagated rows)] in one autograd node."""

    @staticmethod
    def forward(ctx, graph, dims, n_user, n_item, trip, loss_kind, drops, seed, table, *mats):
        loss_rows = torch.cat([trip[:, 0], trip[:, 1] + n_user, trip[:, 2] + n_user]) if RESTRICT_FORWARD else None
        out, saved = propagate_forward(graph, table.detach(), _wps([m.detach() for m in mats]), dims, loss_rows, drops, seed)
        B, dtot = trip.shape[0], out.shape[1]
        coef = torch.empty(B, dtype=torch.float32, device=out.device)
        partials = torch.empty(2 * ((B + 3) // 4), dtype=torch.float32, device=out.device)
        res = torch.empty(2, dtype=torch.float32, device=out.device)
        U, I = out[:n_user], out[n_user:n_user + n_item]
        _lib.check(_lib.load().tagrec_bpr_fwd_f32(_lib.ptr(U), _lib.ptr(I), dtot, dtot, _lib.ptr(U), _lib.ptr(I), dtot,
                                                  dtot, _lib.ptr(trip), B, loss_kind, _lib.ptr(coef),
                                                  _lib.ptr(partials), _lib.ptr(res), _lib.stream_ptr()), "bpr_fwd")
        ctx.graph, ctx.dims, ctx.saved = graph, dims, saved
        ctx.out, ctx.trip, ctx.coef, ctx.nu, ctx.ni = out, trip, coef, n_user, n_item
        return res

    @staticmethod
    def backward(ctx, g):
        out, trip, nu, ni = ctx.out, ctx.trip, ctx.nu, ctx.ni
        dtot = out.shape[1]
        d_out = torch.zeros_like(out)
        U, I = out[:nu], out[nu:nu + ni]
        dU, dI = d_out[:nu], d_out[nu:nu + ni]
        _lib.check(_lib.load().tagrec_bpr_bwd_f32(_lib.ptr(U), _lib.ptr(I), dtot, dtot, _lib.ptr(U), _lib.ptr(I), dtot,
                                                  dtot, _lib.ptr(trip), trip.shape[0], _lib.ptr(ctx.coef),
                                                  _lib.ptr(g.contiguous()), 1.0, _lib.ptr(dU), _lib.ptr(dI),
                                                  _lib.ptr(dU), _lib.ptr(dI), _lib.stream_ptr()), "bpr_bwd")
        d0, dws = propagate_backward(ctx.graph.transpose(), d_out, ctx.saved, ctx.dims)
        ctx.saved = ctx.out = None
        return (None, None, None, None, None, None, None, None, d0, *_mat_grads(dws))


class NGCF(TableModel):
    def __init__(self, data, args=None, config=None, graph=None):
        super().__init__()
        self._config(config if config is not None else _GLOBAL_CFG)
        self._init_table(data, self.use_tag, self.dim_latent, self.device)
        # matrices, in the reference's registration order so a seeded init matches (ngcf.py:45-60)
        self.mat = nn.ParameterDict()
        for k in range(self.num_layer):
            names = [f"W1_{k}", f"b1_{k}"] + ([f"W2_{k}", f"b2_{k}"] if self.agg_type == "bi_agg" else [])
            for name in names:
                shape = (self.dim_layer_list[k] if name[0] == "W" else 1, self.dim_layer_list[k + 1])
                t = torch.empty(*shape)
                nn.init.xavier_uniform_(t)
                self.mat[name] = nn.Parameter(t.to(self.device))
        self.norm_adj = graph if graph is not None else creat_adj(data, self.use_tag, self.norm_type,
                                                                  self.split_adj_k, self.device)

    def _config(self, config):
        self.dim_latent = config["dim_latent"]
        self.num_layer = len(config["dim_layer_list"])
        self.dim_layer_list = [self.dim_latent] + list(config["dim_layer_list"])
        self.agg_type = config["agg_type"]
        self.device = torch.device(config["device"])
        self.message_drop_list = config["message_drop_list"]
        self.norm_type = config["norm_type"]
        self.split_adj_k = config["split_adj_k"]
        self.reg = config["reg"]
        self.loss_func = config["mul_loss_func"]
        self.use_tag = config["use_tag"]
        self.drop_seed = config.get("seed", 2020)

    def _mats(self):
        return [self.mat[f"{n}_{k}"] for k in range(self.num_layer) for n in ("W1", "b1", "W2", "b2")]

    def _fused_ok(self):
        dl = self.dim_layer_list
        dims_ok = all(d in (16, 32, 64, 128) for d in dl)
        return isinstance(self.norm_adj, Graph) and dims_ok

    def _drops(self):
        """(per-layer drop rates, seed of this forward pass) when message dropout is active, else (None, 0): the counter-
        based masks of the library (a function of seed, layer and element), a new seed per training-mode pass."""
        drops = [float(p) for p in self.message_drop_list[:self.num_layer]]
        if not (self.training and any(p > 0 for p in drops)):
            return None, 0
        if torch.cuda.is_current_stream_capturing():
            raise _lib.TagrecError("NGCF: message dropout draws a new seed on the host every step and cannot be captured in a "
                                   "HIP graph")
        self._drop_calls = getattr(self, "_drop_calls", 0) + 1
        return tuple(drops + [0.0] * (self.num_layer - len(drops))), (int(getattr(self, "drop_seed", 2020)) << 24) + self._drop_calls

    def _propagate(self):
        if self.agg_type != "bi_agg":
            raise NotImplementedError                         # ngcf.py:65-68
        if self._fused_ok():
            drops, seed = self._drops()
            return _Propagate.apply(self.norm_adj, tuple(self.dim_layer_list), drops, seed, self.table, *self._mats())
        # operator-by-operator path (row folds, message dropout, odd widths): ngcf.py:73-90 as written
        x = self.table
        outs = [x]
        for k in range(self.num_layer):
            nei = H.split_mm(self.norm_adj, x)
            s = torch.nn.functional.leaky_relu(torch.matmul(nei + x, self.mat[f"W1_{k}"] + self.mat[f"b1_{k}"]), 0.2)
            b = torch.nn.functional.leaky_relu(torch.matmul(nei * x, self.mat[f"W2_{k}"] + self.mat[f"b2_{k}"]), 0.2)
            x = torch.nn.functional.dropout(s + b, p=self.message_drop_list[k], training=self.training)
            outs.append(H.normalize_rows(x))
        return torch.cat(outs, dim=1)

    def forward(self):
        return self._split(self._propagate())

    def loss(self, batch_data):
        batch_data = batch_data.to(self.device, torch.int64).contiguous()
        nu, ni = self.num_list[0], self.num_list[1]
        if self.agg_type == "bi_agg" and self._fused_ok():
            drops, seed = self._drops()
            res = _PropagateBprLoss.apply(self.norm_adj, tuple(self.dim_layer_list), nu, ni, batch_data,
                                          H.loss_kind_id(self.loss_func), drops, seed, self.table, *self._mats())
            return res[0], self.reg * res[1]
        all_users, all_items = self.forward()[:2]
        loss, reg_loss = H.triplet_loss(all_users, all_items, all_users, all_items, batch_data, self.loss_func)
        return loss, self.reg * reg_loss
