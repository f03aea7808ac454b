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
from .train import fused_optimizer


def dense_forward(nei, x, w1p, w2p, xp, inv, z_slot, ldz, row_mask=None):
    """z_slot None: the normalised slot is not written.  row_mask (uint8 per row): rows with a zero byte are neither read
    nor written."""
    n, din = x.shape
    _lib.check(_lib.load().tagrec_ngcf_dense_fwd_rows_f32(_lib.ptr(nei), _lib.ptr(x), _lib.ptr(w1p), _lib.ptr(w2p), n, din,
                                                          w1p.shape[1], _lib.ptr(xp), _lib.ptr(inv), _lib.ptr(z_slot), ldz,
                                                          _lib.ptr(row_mask), _lib.stream_ptr()), "ngcf_dense_fwd")


def dense_backward(dxp, nei, x, w1p, w2p, norm=None, row_mask=None, dz_flags=None):
    """norm = (xp, inv, dz, ldz): the gradient w.r.t. Xp is dxp (may be None) + normalize-backward of the layer's slot
    dz of the concat gradient, formed inside the kernel (dz_flags: rows with a zero byte have dz == 0 and are not read).
    row_mask: rows with a zero byte are skipped altogether -- d_nei / d_xd are left unwritten there and the weight
    gradient counts them as zero."""
    n, din = x.shape
    dout = w1p.shape[1]
    d_nei, d_xd = torch.empty_like(x), torch.empty_like(x)
    dp1, dp2 = torch.empty(n, dout, device=x.device), torch.empty(n, dout, device=x.device)
    lib = _lib.load()
    if norm is not None:
        xp, inv, dz, ldz = norm
        _lib.check(lib.tagrec_ngcf_dense_bwd_rows_f32(_lib.ptr(dxp), _lib.ptr(xp), _lib.ptr(inv), _lib.ptr(dz), ldz,
                                                      _lib.ptr(dz_flags), _lib.ptr(nei), _lib.ptr(x), _lib.ptr(w1p), _lib.ptr(w2p),
                                                      n, din, dout, _lib.ptr(d_nei), _lib.ptr(d_xd), _lib.ptr(dp1), _lib.ptr(dp2),
                                                      _lib.ptr(row_mask), _lib.stream_ptr()), "ngcf_dense_bwd_norm")
    else:
        if row_mask is not None:
            raise _lib.TagrecError("dense_backward: a row mask needs the fused normalize-backward form (norm=...)")
        _lib.check(lib.tagrec_ngcf_dense_bwd_f32(_lib.ptr(dxp), _lib.ptr(nei), _lib.ptr(x), _lib.ptr(w1p), _lib.ptr(w2p), n,
                                                 din, dout, _lib.ptr(d_nei), _lib.ptr(d_xd), _lib.ptr(dp1), _lib.ptr(dp2),
                                                 _lib.stream_ptr()), "ngcf_dense_bwd")
    ws_n = lib.tagrec_ngcf_wgrad_workspace(din, dout)
    ws = torch.empty(ws_n, dtype=torch.float32, device=x.device)
    dw1, dw2 = torch.empty_like(w1p), torch.empty_like(w2p)
    _lib.check(lib.tagrec_ngcf_wgrad_rows_f32(_lib.ptr(nei), _lib.ptr(x), _lib.ptr(dp1), _lib.ptr(dp2), n, din, dout,
                                              _lib.ptr(dw1), _lib.ptr(dw2), _lib.ptr(ws), ws_n, _lib.ptr(row_mask),
                                              _lib.stream_ptr()), "ngcf_wgrad")
    return d_nei, d_xd, dw1, dw2


RESTRICT_FORWARD = True     # loss(): form the top two layers' neighbour sums only on the rows the batch's loss depends on


def _layer_seed(seed, k):
    return (int(seed) * 64 + k) & 0xFFFFFFFFFFFFFFFF


def _drop_renorm(xp, inv, z_slot, ldz, p, seed, rows=None):
    """Message dropout of a layer's output (ngcf.py:85) behind the fused dense kernel: Xp <- mask(seed) Xp / (1 - p) with the
    library's counter-based mask, then the slot of the concatenated output and the inverse norms are recomputed from it
    (two element-wise passes; the MFMA kernels stay in use).  rows: xp holds these rows of the layer output; the mask is
    the full output's, keyed by node id (one mask per node however often a row list names it)."""
    H.message_drop(xp, p, seed, out=xp, rows=rows)
    n, d = xp.shape
    _lib.check(_lib.load().tagrec_rownorm_fwd_f32(_lib.ptr(xp), _lib.ptr(z_slot), ldz, _lib.ptr(inv), n, d, _lib.stream_ptr()),
               "rownorm_fwd")


def _dxp_through_dropout(dx_next, xp, inv, dz, ldz, p, seed, rows=None):
    """d loss / d (pre-dropout Xp) = mask / (1 - p) * (dx_next + normalize-backward(Xp_dropped, inv, dz))."""
    n, d = xp.shape
    g = torch.zeros_like(xp) if dx_next is None else dx_next.contiguous().clone()
    _lib.check(_lib.load().tagrec_rownorm_bwd_f32(_lib.ptr(xp), _lib.ptr(inv), _lib.ptr(dz), ldz, 1.0, _lib.ptr(g), 1, n, d,
                                                  _lib.stream_ptr()), "rownorm_bwd")
    return H.message_drop(g, p, seed, out=g, rows=rows)


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
            if pk > 0:                      # the node's mask (keyed by node id): every slot of a repeated node is identical
                _drop_renorm(xpc, invc, zc, d, pk, _layer_seed(seed, k), rows)
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
                d_nei_c, d_xd_c, dw1, dw2 = dense_backward(_dxp_through_dropout(None, xpc, invc, dzc, d, pk, sk, rows), nc, xc, w1p, w2p)
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


def _scatter_rows(n, rows, compact):
    """[n, D] tensor that holds sum of compact[j] over rows[j] == r at the listed rows and is UNWRITTEN elsewhere."""
    t = torch.empty(n, compact.shape[1], dtype=torch.float32, device=compact.device)
    t.index_fill_(0, rows, 0.0)
    return t.index_add_(0, rows, compact)


def restricted_forward(graph, x0, wps, dims, rows):
    """The forward pass of a training step whose loss reads the concatenated output at `rows` (int64 node ids [T], may
    repeat) only -- the BPR batch rows (ngcf.py:97-101).  Layers below L-1 run on all rows, layer L-1 (neighbour sum AND
    dense block) on the batch rows and their neighbours (row-masked kernels), layer L in compact form on the T batch
    rows alone (`spmm_listed` + the dense block on T rows); no layer writes its normalised slot, the concatenated output
    is formed at the end on the T rows.  Rows a layer did not compute are left UNWRITTEN; every later reader is handed
    the mask.  A node named by several batch slots is computed once per slot: every backward map is linear in the
    slot's upstream gradient, so summing the slots' results (index_add) equals using the summed gradient.
    Returns (out_b [T, sum(dims)], state for `restricted_backward`)."""
    from .lightgcn import spmm_listed
    L, n = len(wps), x0.shape[0]
    mid = graph.mark_rows(rows, torch.zeros(n, dtype=torch.uint8, device=x0.device)) if L >= 2 else None
    saved, x = [], x0
    for k in range(L - 1):
        m = mid if k == L - 2 else None
        w1p, w2p = wps[k]
        nei = torch.empty_like(x)
        if m is None:
            graph.spmm(x, out=nei)
        else:
            graph.spmm_rows(x, nei, m)
        xp = torch.empty(n, dims[k + 1], dtype=torch.float32, device=x0.device)
        inv = torch.empty(n, dtype=torch.float32, device=x0.device)
        dense_forward(nei, x, w1p, w2p, xp, inv, None, 0, m)
        saved.append((x, nei, xp, inv, w1p, w2p, m))
        x = xp
    w1p, w2p = wps[L - 1]
    d = dims[L]
    T = rows.numel()
    nc = spmm_listed(graph, rows, x)
    xc = x.index_select(0, rows)
    xpc = torch.empty(T, d, dtype=torch.float32, device=x0.device)
    invc = torch.empty(T, dtype=torch.float32, device=x0.device)
    out_b = torch.empty(T, sum(dims), dtype=torch.float32, device=x0.device)
    off = dims[0]
    out_b[:, :off] = x0.index_select(0, rows)
    for (_, _, xp, inv, _, _, _) in saved:
        torch.mul(xp.index_select(0, rows), inv.index_select(0, rows)[:, None], out=out_b[:, off:off + xp.shape[1]])
        off += xp.shape[1]
    dense_forward(nc, xc, w1p, w2p, xpc, invc, out_b[:, off:], out_b.shape[1])
    return out_b, (saved, mid, (nc, xc, xpc, invc, w1p, w2p))


def restricted_backward(graph_t, rows, d_b, state, dims, n, fused=None):
    """Gradient of `restricted_forward` given d_b [T, sum(dims)] = d loss / d out_b -> (d_x0 [n, dims[0]], [(dW1', dW2')]).
    The chain starts on the batch rows (compact), lands on their neighbours (row-masked hop) and spreads from there; the
    concat gradient of the lower layers lives on the batch rows (dz_flags), the masked layer's dense backward and weight
    gradient visit the masked rows only, and every product is told which operand rows are valid."""
    saved, mid, (nc, xc, xpc, invc, w1p, w2p) = state
    L = len(saved) + 1
    dtot = d_b.shape[1]
    offs = [0]
    for dd in dims:
        offs.append(offs[-1] + dd)
    dev = d_b.device
    dws = [None] * L
    tflag = torch.zeros(n, dtype=torch.uint8, device=dev)
    tflag.index_fill_(0, rows, 1)
    # top layer, one slot per batch row
    d_nei_c, d_xd_c, dw1, dw2 = dense_backward(None, nc, xc, w1p, w2p, norm=(xpc, invc, d_b[:, offs[L]:], dtot))
    dws[L - 1] = (dw1, dw2)
    g, b = _scatter_rows(n, rows, d_nei_c), _scatter_rows(n, rows, d_xd_c)       # valid on the batch rows only

    def last_product(g_in, in_flags, addend, b_flags):
        """The product that lands on the table: d x0 = A^T g_in + addend (+ the concat gradient's first slot, which lives on
        the batch rows and is folded into the addend first) -- or, with a fused optimizer, Adam applied in its epilogue."""
        addend.index_add_(0, rows, d_b[:, :dims[0]])
        if fused is not None:
            table, opt = fused
            m_, v_, step = opt.fused_state(table)
            graph_t.spmm_axpy_adam(g_in, in_flags, None, addend, 1.0, b_flags, table.data, m_, v_, opt.lr, opt.betas, opt.eps, step,
                                   opt.fused_dev(table))
            opt.fused_commit(table)
            return None
        out = torch.empty(n, dims[0], dtype=torch.float32, device=dev)
        if in_flags is None and b_flags is None:
            graph_t.spmm_axpy(g_in, addend, 1.0, out)
        else:
            graph_t.spmm_axpy_sparse(g_in, in_flags, None, addend, 1.0, out, None, b_flags=b_flags)
        return out

    if L == 1:
        return last_product(g, tflag, b, tflag), dws
    dx = torch.empty(n, dims[L - 1], dtype=torch.float32, device=dev)
    graph_t.spmm_axpy_sparse(g, tflag, None, b, 1.0, dx, mid, b_flags=tflag)     # valid on `mid`
    if L >= 2:
        dzn = _scatter_rows(n, rows, d_b[:, offs[1]:offs[L]])                    # concat gradient of layers 1 .. L-1
        ldz = dzn.shape[1]
    for k in range(L - 2, -1, -1):
        x, nei, xp, inv, w1p, w2p, m = saved[k]
        d_nei, d_xd, dw1, dw2 = dense_backward(dx, nei, x, w1p, w2p, norm=(xp, inv, dzn[:, offs[k + 1] - offs[1]:], ldz),
                                               row_mask=m, dz_flags=tflag)
        dws[k] = (dw1, dw2)
        saved[k] = None
        if k == 0:
            return last_product(d_nei, m, d_xd, m), dws
        dx = torch.empty_like(x)
        if m is not None:
            graph_t.spmm_axpy_sparse(d_nei, m, None, d_xd, 1.0, dx, None, b_flags=m)
        else:
            graph_t.spmm_axpy(d_nei, d_xd, 1.0, dx)
    raise AssertionError("unreachable")


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
    def forward(ctx, graph, dims, n_user, n_item, trip, loss_kind, drops, seed, fused_opt, table, *mats):
        x0 = table.detach()
        ctx.fused = (table, fused_opt) if fused_opt is not None else None
        wps = _wps([m.detach() for m in mats])
        B, n = trip.shape[0], x0.shape[0]
        rows = torch.cat([trip[:, 0], trip[:, 1] + n_user, trip[:, 2] + n_user]) if RESTRICT_FORWARD else None
        coef = torch.empty(B, dtype=torch.float32, device=x0.device)
        partials = torch.empty(2 * ((B + 3) // 4), dtype=torch.float32, device=x0.device)
        res = torch.empty(2, dtype=torch.float32, device=x0.device)
        ctx.graph, ctx.dims, ctx.trip, ctx.coef, ctx.nu, ctx.ni = graph, dims, trip, coef, n_user, n_item
        ctx.compact = bool(RESTRICT_FORWARD and drops is None and len(wps) >= 1 and graph.shape[0] == graph.shape[1]
                           and 3 * B * 16 <= n)                       # a batch that touches most rows gains nothing
        if ctx.compact:
            out_b, ctx.state = restricted_forward(graph, x0, wps, dims, rows)
            dtot = out_b.shape[1]
            ar = torch.arange(B, device=x0.device)
            ctrip = torch.stack([ar, ar, ar + B], dim=1).contiguous()
            U, I = out_b[:B], out_b[B:]
            _lib.check(_lib.load().tagrec_bpr_fwd_f32(_lib.ptr(U), _lib.ptr(I), dtot, dtot, _lib.ptr(U), _lib.ptr(I), dtot, dtot,
                                                      _lib.ptr(ctrip), B, loss_kind, _lib.ptr(coef), _lib.ptr(partials),
                                                      _lib.ptr(res), _lib.stream_ptr()), "bpr_fwd")
            ctx.rows, ctx.out_b, ctx.ctrip, ctx.n = rows, out_b, ctrip, n
            return res
        out, saved = propagate_forward(graph, x0, wps, dims, rows, drops, seed)
        dtot = out.shape[1]
        U, I = out[:n_user], out[n_user:n_user + n_item]
        _lib.check(_lib.load().tagrec_bpr_fwd_f32(_lib.ptr(U), _lib.ptr(I), dtot, dtot, _lib.ptr(U), _lib.ptr(I), dtot,
                                                  dtot, _lib.ptr(trip), B, loss_kind, _lib.ptr(coef),
                                                  _lib.ptr(partials), _lib.ptr(res), _lib.stream_ptr()), "bpr_fwd")
        ctx.saved, ctx.out = saved, out
        return res

    @staticmethod
    def backward(ctx, g):
        if ctx.compact:
            out_b, ctrip = ctx.out_b, ctx.ctrip
            B, dtot = ctrip.shape[0], out_b.shape[1]
            d_b = torch.zeros_like(out_b)
            U, I, dU, dI = out_b[:B], out_b[B:], d_b[:B], d_b[B:]
            _lib.check(_lib.load().tagrec_bpr_bwd_f32(_lib.ptr(U), _lib.ptr(I), dtot, dtot, _lib.ptr(U), _lib.ptr(I), dtot, dtot,
                                                      _lib.ptr(ctrip), B, _lib.ptr(ctx.coef), _lib.ptr(g.contiguous()), 1.0,
                                                      _lib.ptr(dU), _lib.ptr(dI), _lib.ptr(dU), _lib.ptr(dI), _lib.stream_ptr()),
                       "bpr_bwd")
            d0, dws = restricted_backward(ctx.graph.transpose(), ctx.rows, d_b, ctx.state, ctx.dims, ctx.n, ctx.fused)
            ctx.state = ctx.out_b = None
            return (None, None, None, None, None, None, None, None, None, d0, *_mat_grads(dws))
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
        return (None, None, None, None, None, None, None, None, None, d0, *_mat_grads(dws))


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

    fused_capturable = True

    def set_fused_optimizer(self, opt):
        """`Adam.fuse_into(model)`: the compact restricted step applies the TABLE's Adam update in the epilogue of the product
        that lands on it (W / b keep ordinary gradients); every other path hands over a table gradient as usual."""
        self._fused_opt = opt

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
            fused = fused_optimizer(self) if (self.training and torch.is_grad_enabled()) else None
            res = _PropagateBprLoss.apply(self.norm_adj, tuple(self.dim_layer_list), nu, ni, batch_data,
                                          H.loss_kind_id(self.loss_func), drops, seed, fused, self.table, *self._mats())
            return res[0], self.reg * res[1]
        all_users, all_items = self.forward()[:2]
        loss, reg_loss = H.triplet_loss(all_users, all_items, all_users, all_items, batch_data, self.loss_func)
        return loss, self.reg * reg_loss
