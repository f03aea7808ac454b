"""TGCN behind the reference's model surface (/root/reference/model/tgcn.py).

    TGCN(data)                 embed.{user,item,tag,weight}, layer.k.{U,q,p,Wf,bf,atten1.*,conv.*}   (:140-192)
    .forward()                 -> (user, item, tag) each [n, (L+1) D] (concat of normalised layers)     (:204-230)
    .loss(batch[B,3])          BPR (logsigmoid) + reg on propagated rows                                (:235-249)
    .transtag_loss(batch[B,4]) margin ranking on ||u + t - i||_2 over EGO rows + transtag_reg * L2      (:251-261)
    .predict_rating(users)                                                                              (:263-268)

What runs where
  * neighbour-level attention (`Attention1`, :20-37): the k-neighbour gather / softmax / weighted sum and
    its backward are the HIP kernels of csrc/tgcn.hip; the projections they consume (P = ev W1[:D] + b,
    Q = ej W2 computed ONCE per neighbour type and layer instead of once per (node, neighbour), WT = the
    weight-embedding look-up table through W1[D:]) are plain GEMMs;
  * type-level attention, bit-/vector-level convolutions and the fusion layer (:78-106) are evaluated in
    row chunks under activation checkpointing, so the [n, 32 D + 48] convolution output the reference
    materialises for all nodes never exists for more than one chunk (plain GEMMs + elementwise device ops;
    a fused MFMA kernel for this block is the next kernel to write);
  * the static neighbour tables are uploaded once (the reference rebuilds and uploads them for every layer
    of every forward, :194-202; its shuffle there is dead code).
"""
import math

import numpy as np
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from . import _lib, help as H
from . import plan as PL
from . import tgcn_step as TS
from .config import CFG as _GLOBAL_CFG
from .graph import Graph


class InverseTable:
    """A relation's neighbour table inverted once at start-up: for every destination row, the
    (source node, slot) pairs that point at it, as two CSR graphs over the same row pointer --
    S (unit values, columns = flat (node, slot) positions) and G (values = the step's attention weights,
    columns = source nodes).  The attention backward then PULLS dQ = S dh and dEj = G dOut with the SpMM
    kernel instead of scattering with float atomics."""

    def __init__(self, idx, n_dst):
        n, k = idx.shape
        flat = idx.flatten().long()
        pos = torch.nonzero(flat > 0).flatten()
        dest = flat[pos] - 1
        order = torch.argsort(dest, stable=True)
        self.perm = pos[order].contiguous()
        # the same list as int32 + the destination of every entry: what the sort-free per-step inversion compacts
        # (tagrec_inv_filter_i32)
        self.perm32 = self.perm.to(torch.int32)
        self.dest32 = dest[order].to(torch.int32).contiguous()
        rowptr = torch.zeros(n_dst + 1, dtype=torch.int64, device=idx.device)
        torch.cumsum(torch.bincount(dest, minlength=n_dst), 0, out=rowptr[1:])
        ones = torch.ones(self.perm.numel(), dtype=torch.float32, device=idx.device)
        self.S = Graph(rowptr, self.perm.to(torch.int32), ones, (n_dst, n * k))
        self.G = Graph(rowptr, torch.div(self.perm, k, rounding_mode="floor").to(torch.int32), torch.zeros_like(ones),
                       (n_dst, n))


class _TallMM(torch.autograd.Function):
    """X [n, d] @ W [d, a] for n in the millions (the Q / P projections over a node table).  The weight gradient
    X^T dY is a product with n on the contraction axis and a tiny output, for which the library picks one wave-starved
    kernel (2.4 ms for 2 M x 128 x 32, 0.5 TB/s); cut n into slabs, one small product per slab, and add the slabs."""
    SLABS = 256

    @staticmethod
    def forward(ctx, X, W):
        ctx.save_for_backward(X, W)
        return X @ W

    @staticmethod
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        dX = dY @ W.t() if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1]:
            X, dY = X.contiguous(), dY.contiguous()
            n, S = X.shape[0], _TallMM.SLABS
            m = n // S * S
            dW = torch.bmm(X[:m].view(S, m // S, -1).transpose(1, 2), dY[:m].view(S, m // S, -1)).sum(0)
            if m < n:
                dW = dW + X[m:].t() @ dY[m:]
        return dX, dW


def _tall_mm(X, W):
    return _TallMM.apply(X, W) if X.shape[0] >= 65536 else X @ W


def _pull_compact(idxc, attnc, doc, dh, n_dst):
    """dQ = S dh and dEj = G dOut for a COMPACT set of source rows (`InverseTable` built on the spot: one radix sort of
    the rows' n k neighbour ids; pad slots sort behind the last destination row and are never read)."""
    nc, k = idxc.shape
    flat = idxc.flatten()
    key = torch.where(flat > 0, flat - 1, n_dst)
    skey, order = torch.sort(key, stable=True)
    rowptr = torch.searchsorted(skey, torch.arange(n_dst + 1, dtype=skey.dtype, device=skey.device))
    # short-lived handles: their device metadata lives in torch-allocated workspaces (no hipMalloc / hipFree per call)
    G = Graph(rowptr, torch.div(order, k, rounding_mode="floor").to(torch.int32), attnc.flatten().index_select(0, order),
              (n_dst, nc), workspace=True)
    S = G.like(order.to(torch.int32), torch.ones_like(G.val), nc * k)
    return S.spmm(dh), G.spmm(doc)


def _active_rows(d_out):
    """Indices of the rows of d_out [n, D] that hold a non-zero (one pass: tagrec_row_flags_f32 + nonzero)."""
    n, D = d_out.shape
    flags = torch.empty(n, dtype=torch.uint8, device=d_out.device)
    count = torch.zeros(1, dtype=torch.int32, device=d_out.device)
    _lib.check(_lib.load().tagrec_row_flags_f32(_lib.ptr(d_out), n, D, _lib.ptr(flags), _lib.ptr(count), _lib.stream_ptr()),
               "row_flags")
    return torch.nonzero(flags).flatten()


def sum_n(tensors, out=None):
    """tensors[0] + tensors[1] + ... (same shape, contiguous, at most 8 per launch) in one pass over memory."""
    tensors = [t.contiguous() for t in tensors]
    if out is None:
        out = torch.empty_like(tensors[0])
    lib = _lib.load()
    first = True
    while tensors:
        part, tensors = tensors[:8 if first else 7], tensors[8 if first else 7:]
        if not first:
            part = [out] + part
        arr = (_lib.c_void_p * len(part))(*[t.data_ptr() for t in part])
        _lib.check(lib.tagrec_sum_n_f32(_lib.ptr(out), arr, len(part), out.numel(), _lib.stream_ptr()), "sum_n")
        first = False
    return out


class _Fan(torch.autograd.Function):
    """A table with several readers: n_dense aliases of x (the Q projection, the neighbour attentions that gather from it,
    the nodes' own slot) and x[rows_i] for every index list.  Backward = ONE n-way sum of the dense gradients plus an
    index_add of the row-subset gradients, in place of autograd's pairwise accumulation (three passes over the table per
    extra dense gradient; a zero fill and three passes per row subset)."""

    @staticmethod
    def forward(ctx, x, n_dense, *rows):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*rows)
        ctx.n_dense, ctx.shape = n_dense, x.shape
        return tuple([x.view_as(x) for _ in range(n_dense)] + [x.index_select(0, r) for r in rows])

    @staticmethod
    def backward(ctx, *gs):
        rows = ctx.saved_tensors
        dense = [g for g in gs[:ctx.n_dense] if g is not None]
        sub = [(r, g) for r, g in zip(rows, gs[ctx.n_dense:]) if g is not None and g.shape[0] > 0]
        if not dense and not sub:
            return (None, None) + (None,) * len(rows)
        if len(dense) == 1 and not sub:
            out = dense[0]
        elif dense:
            out = sum_n(dense)
        else:
            out = torch.zeros(ctx.shape, dtype=sub[0][1].dtype, device=sub[0][1].device)
        for r, g in sub:
            out.index_add_(0, r, g)
        return (out, None) + (None,) * len(rows)


def fan(x, n_dense, *rows):
    """(aliases, row subsets) of x; plain indexing when nothing needs a gradient."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return [x] * n_dense, [x.index_select(0, r) for r in rows]
    outs = _Fan.apply(x, n_dense, *rows)
    return list(outs[:n_dense]), list(outs[n_dense:])


class _NbrAttention(torch.autograd.Function):
    """(P, Q, WT, v, Ej) + static index tables -> attended neighbour embedding [n, D]."""

    @staticmethod
    def forward(ctx, P, Q, WT, v, Ej, idx, widx, inv=None):
        ctx.inv = inv
        P, Q, WT, v, Ej = (t.contiguous() for t in (P, Q, WT, v, Ej))
        n, A = P.shape
        k, D = idx.shape[1], Ej.shape[1]
        attn = torch.empty(n, k, dtype=torch.float32, device=P.device)
        out = torch.empty(n, D, dtype=torch.float32, device=P.device)
        _lib.check(_timed("attn_fwd", _lib.load().tagrec_tgcn_attn_fwd_f32, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(WT),
                          _lib.ptr(v), _lib.ptr(Ej), _lib.ptr(idx), _lib.ptr(widx), n, k, D, A, _lib.ptr(attn),
                          _lib.ptr(out), _lib.stream_ptr()), "tgcn_attn_fwd")
        ctx.save_for_backward(P, Q, WT, v, Ej, idx, widx, attn)
        return out

    @staticmethod
    def backward(ctx, d_out):
        P, Q, WT, v, Ej, idx, widx, attn = ctx.saved_tensors
        n, A = P.shape
        k, D, n_wt = idx.shape[1], Ej.shape[1], WT.shape[0]
        lib = _lib.load()
        dP = torch.empty_like(P)
        dWT, dv = torch.empty_like(WT), torch.empty_like(v)
        ws_n = lib.tagrec_tgcn_attn_workspace(n_wt, A)
        ws = torch.empty(ws_n, dtype=torch.float32, device=P.device)
        d_out = d_out.contiguous()
        inv = ctx.inv
        # Row-local in d_out like the dense block: nodes whose output gradient is zero (all but the batch rows / their
        # sampled neighbours) are dropped, and the few remaining (node, neighbour) pairs use the scatter form.
        if n >= _SPARSE_MIN_ROWS and not torch.cuda.is_current_stream_capturing():      # (needs a host read)
            active = _active_rows(d_out)
            if active.numel() * 2 < n:
                nc = active.numel()
                dPc = torch.empty(nc, A, dtype=torch.float32, device=P.device)
                pull = nc >= _PULL_MIN_ROWS
                if pull:                 # dh for the active rows; dQ / dEj pulled over a table inverted on the spot
                    dQ = dEj = None
                    dh = torch.empty(nc * k, A, dtype=torch.float32, device=P.device)
                else:                    # a few thousand (node, neighbour) pairs: float atomics into zeroed buffers
                    dQ, dEj, dh = torch.zeros_like(Q), torch.zeros_like(Ej), None
                if nc > 0:
                    sel = lambda x: x.index_select(0, active)
                    Pc, idxc, widxc, attnc, doc = sel(P), sel(idx), sel(widx), sel(attn), sel(d_out)
                    _lib.check(_timed("attn_bwd", lib.tagrec_tgcn_attn_bwd_f32, _lib.ptr(Pc), _lib.ptr(Q), _lib.ptr(WT),
                                      _lib.ptr(v), _lib.ptr(Ej), _lib.ptr(idxc), _lib.ptr(widxc), _lib.ptr(attnc), _lib.ptr(doc),
                                      nc, k, D, A, n_wt, _lib.ptr(dPc), _lib.ptr(dQ), _lib.ptr(dEj), _lib.ptr(dh),
                                      _lib.ptr(dWT), _lib.ptr(dv), _lib.ptr(ws), ws_n, _lib.stream_ptr()), "tgcn_attn_bwd")
                    if pull:
                        dQ, dEj = _pull_compact(idxc, attnc, doc, dh, Ej.shape[0])
                else:
                    dWT.zero_(); dv.zero_()
                dP = torch.zeros_like(P).index_copy_(0, active, dPc)
                return dP, dQ, dWT, dv, dEj, None, None, None
        # a row subset of the pruned forward arrives without a prebuilt table: invert its rows on the spot
        spot = inv is None and n >= _PULL_MIN_ROWS and not torch.cuda.is_current_stream_capturing()
        if inv is None and not spot:     # scatter form: float atomics into zeroed buffers
            dQ, dEj, dh = torch.zeros_like(Q), torch.zeros_like(Ej), None
        else:                            # pull form: write dh, finish with two SpMMs over the inverted table
            dQ = dEj = None
            dh = torch.empty(n * k, A, dtype=torch.float32, device=P.device)
        _lib.check(_timed("attn_bwd", lib.tagrec_tgcn_attn_bwd_f32, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(WT), _lib.ptr(v),
                          _lib.ptr(Ej), _lib.ptr(idx), _lib.ptr(widx), _lib.ptr(attn), _lib.ptr(d_out),
                          n, k, D, A, n_wt, _lib.ptr(dP), _lib.ptr(dQ), _lib.ptr(dEj), _lib.ptr(dh), _lib.ptr(dWT),
                          _lib.ptr(dv), _lib.ptr(ws), ws_n, _lib.stream_ptr()), "tgcn_attn_bwd")
        if inv is not None:
            torch.index_select(attn.reshape(-1), 0, inv.perm, out=inv.G.val)
            dEj = inv.G.spmm(d_out)
            dQ = inv.S.spmm(dh)
        elif spot:
            dQ, dEj = _pull_compact(idx, attn, d_out, dh, Ej.shape[0])
        return dP, dQ, dWT, dv, dEj, None, None, None


def neighbour_attention(P, Q, WT, v, Ej, idx, widx, inv=None):
    return _NbrAttention.apply(P, Q, WT, v, Ej, idx, widx, inv)


class _Att(nn.Module):
    """Parameter holder with the reference's names (tgcn.py:11-18)."""

    def __init__(self, in_features, atten_dim, dim_w):
        super().__init__()
        self.W_1 = nn.Parameter(torch.empty(in_features + dim_w, atten_dim))
        self.W_2 = nn.Parameter(torch.empty(in_features, atten_dim))
        self.b = nn.Parameter(torch.empty(1, atten_dim))
        self.v = nn.Parameter(torch.empty(1, atten_dim))


class _ConvW(nn.Module):
    def __init__(self, *shape):
        super().__init__()
        w = torch.empty(*shape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))      # the draw nn.Conv2d makes at construction (RNG parity)
        self.weight = nn.Parameter(w)


def _dense_block(st, U, q, p, wb, w1, w2, w3, Wf, bf):
    """Type-level attention -> bit-/vector-level convolutions -> fusion, for a chunk of rows.
    st [n, 3, D] = (user-side, item-side, tag-side) vectors of each node (tgcn.py:78-106)."""
    n, _, D = st.shape
    q, p, bf = q.reshape(1, -1), p.reshape(1, -1), bf.reshape(1, -1)
    wb = wb.reshape(wb.shape[0], 1, 3, 1)
    w1, w2, w3 = w1.reshape(-1, 1, 1, D), w2.reshape(-1, 1, 2, D), w3.reshape(-1, 1, 3, D)
    s = torch.relu(st @ U + q) @ p.t()                        # [n,3,1]
    e3 = torch.softmax(s, dim=1) * st                         # scaled, NOT summed
    # bit level: Conv2d(1, C, (3,1)) == a 3 -> C mix per feature
    w = wb[:, 0, :, 0]                                         # [C, 3]
    bit = e3[:, None, 0, :] * w[None, :, 0, None]              # three fused multiply-adds per output element,
    bit = torch.addcmul(bit, e3[:, None, 1, :], w[None, :, 1, None])      # not a batch of 3-deep GEMMs
    bit = torch.addcmul(bit, e3[:, None, 2, :], w[None, :, 2, None])
    bit = torch.relu_(bit).reshape(n, -1)                      # [n, C*D], channel-major
    # vector level: Conv2d(1, V, (j, D)), j = 1..3 == dot products over j stacked rows
    flat = e3.reshape(n, 3 * D)
    v1 = torch.relu(torch.einsum("nhd,cd->nch", e3, w1[:, 0, 0, :])).reshape(n, -1)
    v2 = torch.relu(torch.stack([flat[:, :2 * D] @ w2.reshape(-1, 2 * D).t(),
                                 flat[:, D:] @ w2.reshape(-1, 2 * D).t()], dim=2)).reshape(n, -1)
    v3 = torch.relu(flat @ w3.reshape(-1, 3 * D).t())
    y = torch.cat([bit, v1, v2, v3], dim=1)
    return torch.relu(y @ Wf + bf)


def fused_dense_supported(D, Dout, A, C, V):
    return A == 32 and C == 32 and V == 8 and D in (16, 32, 64, 128) and Dout in (16, 32, 64, 128)


class _FusedDense(torch.autograd.Function):
    """Type-level attention + convolutions + fusion layer as ONE HIP kernel (csrc/tgcn_fuse.hip); the
    [n, 32 D + 48] convolution output is never written to memory."""

    @staticmethod
    def forward(ctx, t0, t1, t2, U, q, p, wb, w1, w2, w3, Wf, bf, chunk_rows):
        t0, t1, t2 = t0.contiguous(), t1.contiguous(), t2.contiguous()
        n, D = t0.shape
        Dout = Wf.shape[1]
        out = torch.empty(n, Dout, dtype=torch.float32, device=t0.device)
        bw = torch.empty(n, 3, dtype=torch.float32, device=t0.device)
        args = [x.contiguous() for x in (U, q, p, wb, w1, w2, w3, Wf, bf)]
        _lib.check(_timed("fuse_fwd", _lib.load().tagrec_tgcn_fuse_fwd_f32, _lib.ptr(t0), _lib.ptr(t1), _lib.ptr(t2), n, D,
                          Dout, U.shape[1], wb.shape[0], w1.shape[0], *[_lib.ptr(a) for a in args], _lib.ptr(bw),
                          _lib.ptr(out), _lib.stream_ptr()), "tgcn_fuse_fwd")
        ctx.save_for_backward(t0, t1, t2, U, q, p, wb, w1, w2, w3, Wf, bf, out, bw)
        ctx.chunk_rows = chunk_rows
        return out

    @staticmethod
    def backward(ctx, d_out):
        t0, t1, t2, U, q, p, wb, w1, w2, w3, Wf, bf, out, bw = ctx.saved_tensors
        n = t0.shape[0]
        d_out = d_out.contiguous()
        # The block is row-local: row r of every input gradient, and row r's share of every weight gradient, is a
        # multiple of d_out[r].  The gradient of a BPR batch reaches only the batch rows in the last layer and their
        # sampled neighbours one layer down, so the rows with d_out[r] == 0 are dropped before the kernels run (exact:
        # they contribute 0) and the results scattered back.  One host read of the row count per call.
        active = None
        if n >= _SPARSE_MIN_ROWS and not torch.cuda.is_current_stream_capturing():      # (needs a host read)
            nz = _active_rows(d_out)
            if nz.numel() * 2 < n:
                active = nz
        if active is None:
            return _FusedDense._backward_rows(t0, t1, t2, U, q, p, wb, w1, w2, w3, Wf, out, bw, d_out)
        if active.numel() == 0:
            z = torch.zeros_like
            return (z(t0), z(t1), z(t2), z(U), z(q), z(p), z(wb), z(w1), z(w2), z(w3), z(Wf), torch.zeros_like(bf), None)
        sel = lambda x: x.index_select(0, active)
        res = _FusedDense._backward_rows(sel(t0), sel(t1), sel(t2), U, q, p, wb, w1, w2, w3, Wf, sel(out), sel(bw), sel(d_out))
        dts = [torch.zeros_like(t0).index_copy_(0, active, d) for d in res[:3]]
        return (*dts, *res[3:])

    @staticmethod
    def _backward_rows(t0, t1, t2, U, q, p, wb, w1, w2, w3, Wf, out, bw, d_out):
        n, D = t0.shape
        Dout, A, C, V = Wf.shape[1], U.shape[1], wb.shape[0], w1.shape[0]
        dev = t0.device
        lib = _lib.load()
        dts = [torch.empty_like(t0) for _ in range(3)]
        yvec = torch.empty(n, 6 * V, dtype=torch.float32, device=dev)
        dfeat = torch.empty(n, 6 * V, dtype=torch.float32, device=dev)
        dS = torch.empty(n, 3 * A, dtype=torch.float32, device=dev)
        small = torch.empty(3 * C + 2 * A, dtype=torch.float32, device=dev)
        ws_n = lib.tagrec_tgcn_fuse_bwd_workspace(Dout)
        ws = torch.empty(ws_n, dtype=torch.float32, device=dev)
        _lib.check(_timed("fuse_bwd", lib.tagrec_tgcn_fuse_bwd_f32, _lib.ptr(t0), _lib.ptr(t1), _lib.ptr(t2), n, D, Dout, A, C,
                          V, _lib.ptr(U), _lib.ptr(q), _lib.ptr(p), _lib.ptr(wb), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(w3),
                          _lib.ptr(Wf), _lib.ptr(out), _lib.ptr(d_out), _lib.ptr(dts[0]), _lib.ptr(dts[1]), _lib.ptr(dts[2]),
                          _lib.ptr(yvec), _lib.ptr(dfeat), _lib.ptr(dS), _lib.ptr(small), _lib.ptr(ws), ws_n,
                          _lib.stream_ptr()), "tgcn_fuse_bwd")
        dwb, dq, dp = small[:3 * C].reshape(C, 3), small[3 * C:3 * C + A], small[3 * C + A:3 * C + 2 * A]
        # every gradient that is a product over the node axis: dWf, G (-> dw1, dw2, dw3) and dU, one launch
        from . import proj as PJ
        dbf = PJ.masked_colsum(d_out, out) if 256 % (Dout // 4) == 0 else (d_out * (out > 0)).sum(0)
        res_n = lib.tagrec_tgcn_fuse_wf_result(D, Dout)
        res = torch.empty(res_n, dtype=torch.float32, device=dev)
        wf_n = lib.tagrec_tgcn_fuse_wf_workspace(D, Dout)
        wf_ws = torch.empty(wf_n, dtype=torch.float32, device=dev)
        _lib.check(_timed("fuse_wf", lib.tagrec_tgcn_fuse_wf_f32, _lib.ptr(t0), _lib.ptr(t1), _lib.ptr(t2), _lib.ptr(bw),
                          _lib.ptr(yvec), _lib.ptr(wb), _lib.ptr(out), _lib.ptr(d_out), _lib.ptr(dfeat), _lib.ptr(dS), n, D,
                          Dout, C, V, _lib.ptr(res), _lib.ptr(wf_ws), wf_n, _lib.stream_ptr()), "tgcn_fuse_wf")
        k_wf = Wf.numel()
        dWf = res[:k_wf].reshape(Wf.shape)
        G = res[k_wf:k_wf + 6 * V * 3 * D].reshape(6 * V, 3, D)            # G[f][j][d] = sum_nodes dfeat[f] e_j[d]
        dU = res[k_wf + 6 * V * 3 * D:].reshape(D, A)
        g1, g2, g3 = G[:3 * V].reshape(V, 3, 3, D), G[3 * V:5 * V].reshape(V, 2, 3, D), G[5 * V:]
        dw1 = g1[:, 0, 0] + g1[:, 1, 1] + g1[:, 2, 2]                                   # feature (c,h) touches e_h
        dw2 = torch.stack([g2[:, 0, 0] + g2[:, 1, 1], g2[:, 0, 1] + g2[:, 1, 2]], dim=1).reshape(V, -1)   # tap a: e_{h+a}
        dw3 = g3.reshape(V, -1)                                                          # tap a: e_a
        return (*dts, dU, dq, dp, dwb, dw1, dw2, dw3, dWf, dbf, None)


class _Layer(nn.Module):
    """`BasicLayer` (tgcn.py:40-137): same parameter names and registration order."""

    def __init__(self, in_features, out_features, atten_dim, weight_dim, num_bit_conv, num_vector_conv):
        super().__init__()
        self.atten1 = nn.ModuleDict()
        for name in ("user", "item", "tag"):
            self.atten1.update({name: _Att(in_features, atten_dim, weight_dim)})
        self.U = nn.Parameter(torch.empty(in_features, atten_dim))
        self.q = nn.Parameter(torch.empty(1, atten_dim))
        self.p = nn.Parameter(torch.empty(1, atten_dim))
        vec = nn.ModuleDict()
        for j in range(1, 4):
            vec.update({f"conv_{j}": _ConvW(num_vector_conv, 1, j, in_features)})
        self.conv = nn.ModuleDict({"bit_level": _ConvW(num_bit_conv, 1, 3, 1), "vec_level": vec})
        in_k = num_bit_conv * in_features + num_vector_conv * (3 + 2 + 1)
        self.Wf = nn.Parameter(torch.empty(in_k, out_features))
        self.bf = nn.Parameter(torch.empty(1, out_features))
        self.in_features = in_features

    def dense(self, trip, chunk_rows, use_checkpoint, fused=True):
        args = (self.U, self.q, self.p, self.conv["bit_level"].weight, self.conv["vec_level"]["conv_1"].weight,
                self.conv["vec_level"]["conv_2"].weight, self.conv["vec_level"]["conv_3"].weight, self.Wf, self.bf)
        if fused and fused_dense_supported(self.in_features, self.Wf.shape[1], self.U.shape[1], args[3].shape[0],
                                           args[4].shape[0]):
            U, q, p, wb, w1, w2, w3, Wf, bf = args
            return _FusedDense.apply(trip[0], trip[1], trip[2], U, q.reshape(-1), p.reshape(-1), wb.reshape(wb.shape[0], 3),
                                     w1.reshape(w1.shape[0], -1), w2.reshape(w2.shape[0], -1), w3.reshape(w3.shape[0], -1),
                                     Wf, bf.reshape(-1), chunk_rows)
        st = torch.stack(trip, dim=1)
        outs = []
        for lo in range(0, st.shape[0], chunk_rows):
            part = st[lo:lo + chunk_rows]
            if use_checkpoint and torch.is_grad_enabled():
                outs.append(checkpoint(_dense_block, part, *args, use_reentrant=False))
            else:
                outs.append(_dense_block(part, *args))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def forward(self, eu, ei, et, ewp, nbr, chunk_rows, use_checkpoint, fused=True, inv=None):
        D = self.in_features
        inv = inv if inv is not None else [None] * 6
        emb = {"user": eu, "item": ei, "tag": et}
        # one GEMM per node type: [Q_t | P_(t<-nb1) | P_(t<-nb2)] = e_t [W2_t | W1_nb1[:D] | W1_nb2[:D]]
        # (Q_t = e_t W2 is shared by the two relations whose NEIGHBOUR type is t; P is per (source, neighbour type))
        others = {"user": ("item", "tag"), "item": ("user", "tag"), "tag": ("user", "item")}
        A = self.U.shape[1]
        Q, P = {}, {}
        for t, (n1, n2) in others.items():
            wcat = torch.cat([self.atten1[t].W_2, self.atten1[n1].W_1[:D], self.atten1[n2].W_1[:D]], dim=1)
            y = _tall_mm(emb[t], wcat)
            Q[t] = y[:, :A]
            P[(t, n1)] = y[:, A:2 * A] + self.atten1[n1].b
            P[(t, n2)] = y[:, 2 * A:] + self.atten1[n2].b
        WT = {t: ewp @ self.atten1[t].W_1[D:] for t in emb}     # weight look-up tables, (n_weight + 1) x A

        def att(src, nb, r):
            a = self.atten1[nb]
            return neighbour_attention(P[(src, nb)], Q[nb], WT[nb], a.v.reshape(-1), emb[nb], nbr[r][0], nbr[r][1], inv[r])

        eu_i, eu_t = att("user", "item", 0), att("user", "tag", 1)
        ei_u, ei_t = att("item", "user", 2), att("item", "tag", 3)
        et_u, et_i = att("tag", "user", 4), att("tag", "item", 5)
        outs = []
        for trip in ((eu, eu_i, eu_t), (ei_u, ei, ei_t), (et_u, et_i, et)):
            outs.append(self.dense(trip, chunk_rows, use_checkpoint, fused))
        return outs


    def forward_rows(self, emb, ewp, nbr, rows_out, rows_in, pos_in, chunk_rows, use_checkpoint, fused=True, inv=None,
                     extra_rows=None):
        """`forward` restricted to the rows a mini-batch's loss depends on, on COMPACT tables.  emb[t]: the input table
        of type t -- all rows when rows_in[t] is None, otherwise only the rows rows_in[t] (in that order), with
        pos_in[t][id + 1] = 1 + position of node id in it (int32, 0 for the pad id).  rows_out[t]: rows whose outputs
        are wanted (None = every row; otherwise a subset of rows_in[t] that also holds every neighbour of those rows).
        extra_rows[t] (optional): positions in emb[t] the caller wants as well (the batch rows the loss reads).
        Returns ({type: outputs of those rows, in that order}, {type: emb[t][extra_rows[t]]}).
        Every table is read through `fan`, so its gradient is formed by one n-way sum."""
        D = self.in_features
        inv = inv if inv is not None else [None] * 6
        others = {"user": ("item", "tag"), "item": ("user", "tag"), "tag": ("user", "item")}
        A = self.U.shape[1]
        pick = lambda x, rows: x if rows is None else x.index_select(0, rows)
        Q, P, selfv, nb_alias, extras = {}, {}, {}, {}, {}
        for t, (n1, n2) in others.items():
            if rows_out[t] is None:
                self_idx = None
            elif rows_in[t] is None:
                self_idx = rows_out[t]
            else:
                self_idx = pos_in[t].index_select(0, rows_out[t] + 1).long() - 1
            sels = [i for i in (self_idx, extra_rows.get(t) if extra_rows else None) if i is not None]
            # dense readers: the Q projection, the two relations whose NEIGHBOUR type is t, (every row's own slot: two readers)
            al, sub = fan(emb[t], 3 + (2 if self_idx is None else 0), *sels)
            Q[t] = _tall_mm(al[0], self.atten1[t].W_2)       # read wherever t is the NEIGHBOUR type: rows_in[t]
            nb_alias[t] = [al[1], al[2]]
            if self_idx is None:
                s_p, s_d = al[3], al[4]
            else:
                (s_p, s_d), _ = fan(sub[0], 2)
                sub = sub[1:]
            if sub:
                extras[t] = sub[0]
            selfv[t] = s_d
            y = _tall_mm(s_p, torch.cat([self.atten1[n1].W_1[:D], self.atten1[n2].W_1[:D]], dim=1))
            P[(t, n1)] = y[:, :A] + self.atten1[n1].b
            P[(t, n2)] = y[:, A:] + self.atten1[n2].b
        WT = {t: ewp @ self.atten1[t].W_1[D:] for t in emb}

        def att(src, nb, r):
            a = self.atten1[nb]
            rows = rows_out[src]
            if rows is not None and rows.numel() == 0:
                return emb[nb].new_zeros(0, D)
            idx, widx = pick(nbr[r][0], rows), pick(nbr[r][1], rows)
            if rows_in[nb] is not None:                      # neighbour ids -> positions in the compact table
                idx = pos_in[nb].index_select(0, idx.flatten().long()).reshape(idx.shape)
            return neighbour_attention(P[(src, nb)].contiguous(), Q[nb], WT[nb], a.v.reshape(-1), nb_alias[nb].pop(), idx, widx,
                                       inv[r] if rows is None and rows_in[nb] is None else None)

        eu_i, eu_t = att("user", "item", 0), att("user", "tag", 1)
        ei_u, ei_t = att("item", "user", 2), att("item", "tag", 3)
        et_u, et_i = att("tag", "user", 4), att("tag", "item", 5)
        trips = {"user": (selfv["user"], eu_i, eu_t), "item": (ei_u, selfv["item"], ei_t), "tag": (et_u, et_i, selfv["tag"])}
        return {t: (self.dense(trip, chunk_rows, use_checkpoint, fused) if trip[0].shape[0] > 0
                    else trip[0].new_zeros(0, self.Wf.shape[1])) for t, trip in trips.items()}, extras


def neighbor_tables(data, neighbor_k, seed=0):
    """First-`neighbor_k`-column semantics of the reference's tables (data/tgcn_load.py:41-53 +
    data/utils.py:87-106 + model/tgcn.py:199) without materialising (n, max_deg): per row of each of the
    six relations (ui, ut, iu, it, tu, ti), k neighbour ids (+1, 0 = no neighbours) drawn WITH replacement
    when deg < max_deg and without when deg == max_deg, and the matching integer edge weights."""
    rng = np.random.RandomState(seed)

    def coo(c):
        return np.asarray(c.row, np.int64), np.asarray(c.col, np.int64), np.asarray(c.data, np.float32), c.shape

    def csr(c, transpose=False):
        r, cc, v, shape = coo(c)
        if transpose:
            r, cc, shape = cc, r, (shape[1], shape[0])
        key = r * shape[1] + cc
        ukey, inv = np.unique(key, return_inverse=True)
        w = np.bincount(inv, weights=v).astype(np.int64)
        rows, cols = ukey // shape[1], ukey % shape[1]
        ptr = np.zeros(shape[0] + 1, np.int64)
        np.cumsum(np.bincount(rows, minlength=shape[0]), out=ptr[1:])
        return ptr, cols, w

    out = []
    for blk, tr in ((data.ui_adj, False), (data.ut_adj, False), (data.ui_adj, True),
                    (data.it_adj, False), (data.ut_adj, True), (data.it_adj, True)):
        ptr, cols, w = csr(blk, tr)
        deg = np.diff(ptr)
        n, max_deg = len(deg), int(deg.max()) if len(deg) else 0
        ids = np.zeros((n, neighbor_k), np.int32)
        wts = np.zeros((n, neighbor_k), np.int32)
        for r in np.flatnonzero(deg > 0):
            lo, d = ptr[r], deg[r]
            if d == max_deg and d >= neighbor_k:
                pick = lo + rng.permutation(d)[:neighbor_k]
            else:
                pick = lo + rng.randint(0, d, size=neighbor_k)
            ids[r], wts[r] = cols[pick] + 1, w[pick]
        out.append((ids, wts))
    return out


def neighbor_tables_device(rel, neighbor_k, seed=0):
    """Same first-k semantics as `neighbor_tables`, on the GPU, from device CSR relations
    (`synth.make_tripartite_device(...).rel`): k draws with replacement per row; rows whose degree equals
    the relation's maximum get k distinct neighbours (a prefix of a random permutation)."""
    out = []
    for name in ("ui", "ut", "iu", "it", "tu", "ti"):
        ptr, col, w = rel[name]
        dev = ptr.device
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed + len(out))
        deg = ptr[1:] - ptr[:-1]
        n = deg.numel()
        r = torch.rand(n, neighbor_k, device=dev, generator=gen)
        pos = ptr[:-1, None] + (r * deg[:, None]).long().clamp_(max=int(deg.max()) - 1).clamp_(min=0)
        pos = torch.minimum(pos, (ptr[1:, None] - 1).clamp_(min=0))
        ids = (col[pos.clamp_(max=max(col.numel() - 1, 0))].long() + 1)
        wts = w[pos].long()
        has = (deg > 0)[:, None]
        ids, wts = ids * has, wts * has
        full = torch.nonzero((deg == deg.max()) & (deg >= neighbor_k)).flatten()
        for row in full.tolist():                     # the max-degree rows: sampling without replacement
            lo, d = int(ptr[row]), int(deg[row])
            pick = lo + torch.randperm(d, device=dev, generator=gen)[:neighbor_k]
            ids[row], wts[row] = col[pick].long() + 1, w[pick].long()
        out.append((ids.to(torch.int32).contiguous(), wts.to(torch.int32).contiguous()))
    return out


timing = None     # set to {} to record (start, end) events around the attention kernels (bench.py)
_PULL_MIN_ROWS = 32768         # compact attention backward: pull form from this many active rows on (measured at C4: the
                               # scatter form of a 20 k-row relation is one launch, the pull form ~35 small ones)
_SPARSE_MIN_ROWS = 16384      # below this a backward call is too small for dropping its zero-gradient rows to pay


def _timed(name, fn, *args):
    if timing is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    timing.setdefault(name, []).append((e0, e1))
    return rc


class TGCN(nn.Module):
    def __init__(self, data, args=None, config=None, neighbors=None):
        super().__init__()
        self._config(config if config is not None else _GLOBAL_CFG)
        if self.device.type != "cuda":
            raise _lib.TagrecError("TGCN: tagrec_amd needs a GPU device (no CPU path)")
        _lib.load()
        self.num_user, self.num_item, self.num_tag = data.num["user"], data.num["item"], data.num["tag"]
        self.prune_forward = self._prune_cfg[0] and (self.num_user + self.num_item + self.num_tag) >= self._prune_cfg[1]
        self.num_weight = data.num["weight"]
        # parameters: same construction and registration order as tgcn.py:173-192 (RNG parity)
        self.embed = nn.ParameterDict({
            "user": nn.Parameter(torch.empty(self.num_user, self.dim_latent)),
            "item": nn.Parameter(torch.empty(self.num_item, self.dim_latent)),
            "tag": nn.Parameter(torch.empty(self.num_tag, self.dim_latent)),
            "weight": nn.Parameter(torch.empty(self.num_weight, self.dim_weight)),
        })
        self.layer = nn.ModuleDict()
        for k in range(self.num_layer):
            self.layer.update({f"{k}": _Layer(self.dim_layer_list[k], self.dim_layer_list[k + 1], self.dim_atten,
                                              self.dim_weight, self.num_bit_conv, self.num_vec_conv)})
        for prm in self.parameters():
            nn.init.xavier_uniform_(prm)
        self.to(self.device)
        # static neighbour tables, uploaded once: the first neighbor_k columns of the reference's tables
        if neighbors is None:
            if hasattr(data, "get_all_neighbor"):
                neighbors = data.get_all_neighbor()
            elif getattr(data, "rel", None) is not None:
                neighbors = neighbor_tables_device(data.rel, self.neighbor_k, self.seed)
            else:
                neighbors = neighbor_tables(data, self.neighbor_k, self.seed)
        def up(t):
            if isinstance(t, torch.Tensor):
                return t[:, :self.neighbor_k].to(self.device, torch.int32).contiguous()
            return torch.as_tensor(np.asarray(t)[:, :self.neighbor_k].astype(np.int32)).contiguous().to(self.device)
        self.nbr = [tuple(up(t) for t in pair) for pair in neighbors]
        # most frequent weight index per relation (a hint for the attention backward: that row of dWT is summed in registers)
        self.w_major = [int(torch.bincount(w.flatten()[:2_000_000].long()).argmax()) if w.numel() else -1 for _, w in self.nbr]
        n_dst = [self.num_item, self.num_tag, self.num_user, self.num_tag, self.num_user, self.num_item]
        self.inv = [InverseTable(self.nbr[r][0], n_dst[r]) for r in range(6)] if self.pull_backward else None
        self._eval_cache = None
        self._row_plan, self._need_pos = None, None
        self._fused_opt = None

    def _config(self, config):
        self.dim_latent = config["dim_latent"]
        self.dim_weight = config["dim_weight"]
        self.num_layer = len(config["dim_layer_list"])
        self.dim_layer_list = [self.dim_latent] + list(config["dim_layer_list"])
        self.dim_atten = config["dim_atten"]
        self.num_bit_conv = config["num_bit_conv"]
        self.num_vec_conv = config["num_vec_conv"]
        self.message_drop_list = config["message_drop_list"]
        self.device = torch.device(config["device"])
        self.neighbor_k = config["neighbor_k"]
        self.reg = config["reg"]
        self.transtag_reg = config["transtag_reg"]
        self.loss_func = config["mul_loss_func"]
        self.margin = config["margin"]
        self.seed = config.get("seed", 2020)
        self.chunk_rows = config.get("tgcn_chunk_rows", 65536)
        # loss(): run every layer only on the rows the batch's loss depends on (see _forward_rows); off below
        # config["tgcn_prune_min_nodes"] nodes, where the bookkeeping costs more than it saves
        self._prune_cfg = (bool(config.get("tgcn_prune_forward", True)), config.get("tgcn_prune_min_nodes", 200_000))
        self.use_checkpoint = config.get("tgcn_checkpoint", True)
        self.fused_dense = config.get("tgcn_fused_dense", True)
        self.pull_backward = config.get("tgcn_pull_backward", True)
        self.step_node = config.get("tgcn_step_node", True)     # loss(): the restricted step as one hand-derived node

    def _step_node_ok(self):
        d = self.dim_layer_list
        return self.fused_dense and all(fused_dense_supported(d[i], d[i + 1], self.dim_atten, self.num_bit_conv, self.num_vec_conv)
                                        for i in range(self.num_layer))

    def train(self, mode=True):
        self._eval_cache = None
        return super().train(mode)

    def fused_tables(self):
        """The parameters whose Adam update the restricted step can apply itself (`Adam.fuse_into`): the three node tables."""
        return [self.embed["user"], self.embed["item"], self.embed["tag"]]

    def set_fused_optimizer(self, opt):
        """`Adam.fuse_into(model)`: the step node (tgcn_step.py) applies the node tables' Adam update in the epilogue of the
        product that forms the last term of their gradient (dQ W_2^T of the bottom layer); every other path -- the TransTag
        phase, the autograd-composed passes -- hands the optimizer gradients as usual.  None switches it off."""
        self._fused_opt = opt

    def _drops(self):
        """(per-layer drop rates, seed of this pass) when message dropout is active (tgcn.py:217-219), else (None, 0): the
        library's counter-based masks, a function of (seed, layer, node type, node id, column); a new seed per
        training-mode pass.  Parity with the reference in training mode is statistical (it draws torch's generator)."""
        drops = [float(p) for p in self.message_drop_list[:self.num_layer]]
        if not (self.training and any(p > 0 for p in drops)):
            return None, 0
        if torch.cuda.is_current_stream_capturing():
            raise _lib.TagrecError("TGCN: message dropout draws a new seed on the host every step and cannot be captured")
        self._drop_calls = getattr(self, "_drop_calls", 0) + 1
        return tuple(drops + [0.0] * (self.num_layer - len(drops))), (int(self.seed) << 24) + self._drop_calls

    def forward(self):
        eu, ei, et, ew = self.embed["user"], self.embed["item"], self.embed["tag"], self.embed["weight"]
        ewp = torch.cat([ew.new_zeros(1, ew.shape[1]), ew])          # index 0 = pad (tgcn.py:21-24)
        cu, ci, ct = [eu], [ei], [et]
        drops, seed = self._drops()
        for i, layer in enumerate(self.layer.values()):
            eu, ei, et = layer(eu, ei, et, ewp, self.nbr, self.chunk_rows, self.use_checkpoint, self.fused_dense, self.inv)
            if drops and drops[i] > 0:
                eu, ei, et = (H.message_dropout(x.contiguous(), drops[i], TS._drop_seed(seed, i, t))
                              for x, t in ((eu, "user"), (ei, "item"), (et, "tag")))
            cu.append(H.normalize_rows(eu))
            ci.append(H.normalize_rows(ei))
            ct.append(H.normalize_rows(et))
        return torch.cat(cu, dim=1), torch.cat(ci, dim=1), torch.cat(ct, dim=1)

    def get_ego_embed(self):
        return self.embed["user"], self.embed["item"], self.embed["tag"]

    # relation r = (source type, neighbour type), in the order of the neighbour tables (ui, ut, iu, it, tu, ti)
    _RELATIONS = (("user", "item"), ("user", "tag"), ("item", "user"), ("item", "tag"), ("tag", "user"), ("tag", "item"))

    def _needed_rows(self, batch):
        """need[l][t]: the rows of layer l's output (l = 1..L; type t) that the loss of `batch` depends on -- the batch
        rows at the top, plus, one layer down, their k sampled neighbours under every relation -- or None once a type
        needs more than half of its rows (then all of them are computed).  Built by csrc/plan.hip (two launches and one
        host read per level); `self._need_pos[l][t]` is the matching position map (int32 [n + 1]: 1 + position of row v at
        slot v + 1, 0 elsewhere) the compact tables of the step are numbered by."""
        sizes = {"user": self.num_user, "item": self.num_item, "tag": self.num_tag}
        if self._row_plan is None:
            self._row_plan = PL.RowPlan(sizes, self.device)
        plan = self._row_plan
        L = len(self.layer)
        B, p0 = batch.shape[0], batch.data_ptr()
        need, pos = [None] * (L + 1), [None] * (L + 1)
        need[L], pos[L] = plan.level([(None, p0, 3, B, 0, "user"), (None, p0 + 8, 3, B, 0, "item"), (None, p0 + 16, 3, B, 0, "item")])
        for l in range(L, 1, -1):
            cur = need[l]
            descs = [(None, cur[t], 1, cur[t].numel(), 0, t) for t in sizes if cur[t] is not None and cur[t].numel()]
            for r, (src, nb) in enumerate(self._RELATIONS):
                if cur[src] is None:
                    descs.append((self.nbr[r][0], None, 1, sizes[src], sizes[src], nb))
                elif cur[src].numel():
                    descs.append((self.nbr[r][0], cur[src], 1, cur[src].numel(), sizes[src], nb))
            rows, ps = plan.level(descs, all_types=[t for t in sizes if cur[t] is None])
            for t, n in sizes.items():
                if rows[t].numel() * 2 > n:
                    rows[t], ps[t] = None, None
            need[l - 1], pos[l - 1] = rows, ps
        self._need_pos = pos
        return need

    def _forward_rows(self, batch):
        """The forward pass restricted, layer by layer, to `_needed_rows`, on compact tables (a layer's output holds
        only the rows the next layer reads; neighbour ids are renumbered into it).  Returns (users, items, triplets):
        the concatenated layer outputs of the batch's distinct users / items and the batch renumbered into those
        tables -- the same rows `forward()` would give the loss.  Bounded fan-in (k sampled neighbours) is what makes
        this pay for TGCN: the top layer runs on the <= 3 B batch rows, the one below on <= 2 k of those per row."""
        need = self._needed_rows(batch)
        sizes = {"user": self.num_user, "item": self.num_item, "tag": self.num_tag}
        ew = self.embed["weight"]
        ewp = torch.cat([ew.new_zeros(1, ew.shape[1]), ew])
        emb = {"user": self.embed["user"], "item": self.embed["item"], "tag": self.embed["tag"]}
        top = need[len(self.layer)]
        cat = {"user": [], "item": []}
        rows_in = {t: None for t in emb}
        pos_in = {t: None for t in emb}
        extra = {t: top[t] for t in cat}                      # positions in the current tables of the rows the loss reads
        for i, layer in enumerate(self.layer.values()):
            rows_out = need[i + 1]
            outs, picked = layer.forward_rows(emb, ewp, self.nbr, rows_out, rows_in, pos_in, self.chunk_rows, self.use_checkpoint,
                                              self.fused_dense, self.inv, extra)
            for t in cat:                                       # the loss reads the normalised rows of the batch only
                pt = picked[t]
                cat[t].append(pt if i == 0 or not pt.shape[0] else H.normalize_rows(pt))
            p = self.message_drop_list[i]
            nxt, pos_out = {}, {}
            for t, o in outs.items():
                if self.training and p > 0:
                    o = torch.nn.functional.dropout(o, p=p, training=True)
                nxt[t] = o
                pos_out[t] = self._need_pos[i + 1][t] if rows_out[t] is not None else None
            extra = {t: (top[t] if rows_out[t] is None else PL.lookup(pos_out[t], top[t])) for t in cat}
            emb, rows_in, pos_in = nxt, rows_out, pos_out
        for t in cat:
            ot = emb[t].index_select(0, extra[t])
            cat[t].append(H.normalize_rows(ot) if ot.shape[0] else ot)
        return torch.cat(cat["user"], dim=1), torch.cat(cat["item"], dim=1), self._batch_positions(batch)

    def _batch_positions(self, batch):
        """The batch renumbered into the tables of its distinct users / items (the top level of the last `_needed_rows`)."""
        B, p0 = batch.shape[0], batch.data_ptr()
        trip = torch.empty(B, 3, dtype=torch.int64, device=batch.device)
        top_pos = self._need_pos[len(self.layer)]
        for c, t in enumerate(("user", "item", "item")):
            PL.lookup(top_pos[t], p0 + 8 * c, stride=3, n=B, out=trip.data_ptr() + 8 * c, out_stride=3)
        return trip

    def loss(self, batch_data):
        batch_data = batch_data.to(self.device, torch.int64).contiguous()
        restricted = self.prune_forward and self.training and not torch.cuda.is_current_stream_capturing()
        if restricted and self.step_node and torch.is_grad_enabled() and self._step_node_ok():
            # the whole step as one hand-derived autograd node (tgcn_step.py)
            drops, seed = self._drops()
            flat = [p for k in range(self.num_layer) for p in TS.layer_params(self.layer[str(k)])]
            res = TS.TgcnBprLoss.apply(self, batch_data, drops, seed, self.embed["user"], self.embed["item"], self.embed["tag"],
                                       self.embed["weight"], *flat)
            return res[0], self.reg * res[1]
        if restricted:
            all_users, all_items, batch_data = self._forward_rows(batch_data)
        else:
            all_users, all_items = self.forward()[:2]
        loss, reg_loss = H.triplet_loss(all_users, all_items, all_users, all_items, batch_data, self.loss_func)
        return loss, self.reg * reg_loss

    def transtag_loss(self, batch_data):
        batch_data = batch_data.to(self.device, torch.int64).contiguous()
        eu, ei, et = self.get_ego_embed()
        loss, reg_loss = H.transtag_batch_loss(eu, ei, et, batch_data, self.margin)
        return loss, self.transtag_reg * reg_loss

    def predict_rating(self, users):
        if self.training or self._eval_cache is None:
            with torch.no_grad():
                all_users, all_items = self.forward()[:2]
            if not self.training:
                self._eval_cache = (all_users, all_items)
        else:
            all_users, all_items = self._eval_cache
        return torch.sigmoid(torch.matmul(all_users[users.to(self.device)], all_items.t()))
