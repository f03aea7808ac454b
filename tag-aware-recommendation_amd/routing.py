"""Host side of the routed-propagation kernels (csrc/routing.hip): propagation whose edge weights are
recomputed inside the forward pass, per factor (DGCF /root/reference/model/dgcf.py:70-110, DisenGCN
model/disengcn.py:23-46).

Per-entry data is `[nnz, K]` (K factors interleaved) in the CSR entry order of the `Graph`; per-node data `[N, K]`;
embeddings `[N, D]` with factor k in columns `[k D/K, (k+1) D/K)`.  The reference detaches the routing weights
before they become edge values (dgcf.py:93, disengcn.py:37), so gradients only ever flow through the embedding
operand; the backward of a routed product is the routed product with the transposed weights (`permute`)."""
import torch

from . import _lib
from .graph import Graph


class RoutingGraph:
    """Structure of an adjacency (values unused) + what the backward products need: the transposed structure and the
    permutation that reorders per-entry data into its entry order.  For a symmetric structure (DGCF / DisenGCN: the
    "plain" tag-aware adjacency) the transposed matrix has the same rowptr / col and only the permutation is kept."""

    def __init__(self, graph):
        if not isinstance(graph, Graph):
            raise _lib.TagrecError("RoutingGraph: needs a single Graph (row folds are not supported)")
        self.graph = graph
        n_rows, n_cols = graph.shape
        deg = graph.rowptr[1:] - graph.rowptr[:-1]
        rows = torch.repeat_interleave(torch.arange(n_rows, device=graph.device), deg)
        cols = graph.col.long()
        perm = torch.argsort(cols * n_rows + rows, stable=True)
        self.symmetric = n_rows == n_cols and torch.equal(rows[perm], cols) and torch.equal(cols[perm], rows)
        if self.symmetric:
            self.graph_t = graph
        else:
            rowptr_t = torch.zeros(n_cols + 1, dtype=torch.int64, device=graph.device)
            torch.cumsum(torch.bincount(cols, minlength=n_cols), 0, out=rowptr_t[1:])
            self.graph_t = Graph(rowptr_t, rows[perm].to(torch.int32).contiguous(), graph.val[perm].contiguous(), (n_cols, n_rows))
        self.rows, self.cols = rows, cols
        self.perm = perm.to(torch.int32).contiguous()
        self.nnz, self.n, self.device = int(cols.numel()), n_rows, graph.device

    @classmethod
    def from_edges(cls, rows, cols, n, device):
        """CSR over the given (row, col) entries, duplicates kept as separate entries.  Returns (RoutingGraph, order):
        CSR entry j is input edge order[j]."""
        rows = torch.as_tensor(rows, dtype=torch.int64, device=device)
        cols = torch.as_tensor(cols, dtype=torch.int64, device=device)
        order = torch.argsort(rows * n + cols, stable=True)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
        torch.cumsum(torch.bincount(rows, minlength=n), 0, out=rowptr[1:])
        g = Graph(rowptr, cols[order].to(torch.int32).contiguous(), torch.ones(rows.numel(), dtype=torch.float32, device=device), (n, n))
        return cls(g), order

    # ---- kernels ------------------------------------------------------------------------------------------
    def softmax(self, logits):
        w = torch.empty_like(logits)
        _lib.check(_lib.load().tagrec_route_softmax_f32(_lib.ptr(logits), _lib.ptr(w), logits.shape[0], logits.shape[1],
                                                        _lib.stream_ptr()), "route_softmax")
        return w

    def rowsum_rsqrt(self, w):
        d = torch.empty(self.n, w.shape[1], dtype=torch.float32, device=self.device)
        self.graph._call("route_rowsum_rsqrt", _lib.load().tagrec_route_rowsum_rsqrt_f32, self.graph.handle, _lib.ptr(w),
                         w.shape[1], _lib.ptr(d), _lib.stream_ptr())
        return d

    def permute(self, w):
        wt = torch.empty_like(w)
        _lib.check(_lib.load().tagrec_route_permute_f32(_lib.ptr(w), _lib.ptr(self.perm), _lib.ptr(wt), w.shape[0], w.shape[1],
                                                        _lib.stream_ptr()), "route_permute")
        return wt

    def spmm(self, w, x, post=None, self_add=None, b=None, b_scale=0.0, raw=True, normed=False, transposed=False,
             row_mask=None, sparse_x=False):
        """(Y, Yn, inv): y = post * (A(w) x) + self_add + b_scale * b;  Yn / inv = per-slice L2 normalisation of y.
        transposed=True: A(w)^T x (w in the ORIGINAL entry order; it is permuted here).
        row_mask (uint8 [n]): only the rows with a non-zero byte are computed; the others come back as zeros.
        sparse_x: x is expected to have many all-zero rows (a gradient a few hops from the batch rows): they are
        flagged first and not gathered (same result)."""
        _lib.require_gpu_tensor(x, torch.float32, "route_spmm x")
        g = self.graph
        if transposed:
            g, w = self.graph_t, self.permute(w)
        K, D = w.shape[1], x.shape[1]
        n_out = g.shape[0]
        new = torch.zeros if row_mask is not None else torch.empty
        y = new(n_out, D, dtype=torch.float32, device=self.device) if raw else None
        yn = new(n_out, D, dtype=torch.float32, device=self.device) if normed else None
        inv = new(n_out, K, dtype=torch.float32, device=self.device) if normed else None
        flags = count = None
        if sparse_x:
            flags = torch.empty(x.shape[0], dtype=torch.uint8, device=self.device)
            count = torch.zeros(1, dtype=torch.int32, device=self.device)
            _lib.check(_lib.load().tagrec_row_flags_f32(_lib.ptr(x), x.shape[0], D, _lib.ptr(flags), _lib.ptr(count),
                                                        _lib.stream_ptr()), "row_flags")
        name = "route_spmm" if (row_mask is None and not sparse_x) else "route_spmm_restricted"     # timing key
        self.graph._call(name, _lib.load().tagrec_route_spmm_ex_f32, g.handle, _lib.ptr(w), K, _lib.ptr(x),
                         _lib.ptr(post), _lib.ptr(self_add), _lib.ptr(b), float(b_scale), _lib.ptr(y), _lib.ptr(yn),
                         _lib.ptr(inv), _lib.ptr(row_mask), _lib.ptr(flags), _lib.ptr(count), D, _lib.stream_ptr())
        return y, yn, inv

    def score(self, h, t, logits, accumulate, row_mask=None):
        """row_mask: only the entries of rows with a non-zero byte are scored (the others keep their logits)."""
        self.graph._call("route_score" if row_mask is None else "route_score_rows", _lib.load().tagrec_route_score_rows_f32,
                         self.graph.handle, _lib.ptr(h), _lib.ptr(t),
                         _lib.ptr(logits), logits.shape[1], int(bool(accumulate)), _lib.ptr(row_mask), h.shape[1],
                         _lib.stream_ptr())

    def loss_row_mask(self, loss_rows):
        """uint8 [n] mask of the rows a loss reads, or None when they are too many for a restriction to pay."""
        if loss_rows is None or loss_rows.numel() * 16 > self.n:
            return None
        return torch.zeros(self.n, dtype=torch.uint8, device=self.device).index_fill_(0, loss_rows, 1)


    def row_softmax(self, logits):
        a = torch.empty_like(logits)
        _lib.check(_lib.load().tagrec_row_softmax_fwd_f32(self.graph.handle, _lib.ptr(logits), _lib.ptr(a), _lib.stream_ptr()),
                   "row_softmax_fwd")
        return a

    def row_softmax_bwd(self, a, da):
        dl = torch.empty_like(a)
        _lib.check(_lib.load().tagrec_row_softmax_bwd_f32(self.graph.handle, _lib.ptr(a), _lib.ptr(da), _lib.ptr(dl),
                                                          _lib.stream_ptr()), "row_softmax_bwd")
        return dl


# ---- differentiable edge-value operators (KGAT: attention values carry gradient, model/kgat.py:88-104) --------------
class _EdgeScore(torch.autograd.Function):
    """s[j] = < H[row_j], T[col_j] > for every stored entry j."""

    @staticmethod
    def forward(ctx, H, T, rg):
        H, T = H.contiguous(), T.contiguous()
        s = torch.empty(rg.nnz, 1, dtype=torch.float32, device=H.device)
        rg.score(H, T, s, accumulate=False)
        ctx.rg = rg
        ctx.save_for_backward(H, T)
        return s.view(-1)

    @staticmethod
    def backward(ctx, ds):
        H, T = ctx.saved_tensors
        ds = ds.contiguous().view(-1, 1)
        dH, _, _ = ctx.rg.spmm(ds, T)                       # dH[r] = sum_j ds_j T[col_j]
        dT, _, _ = ctx.rg.spmm(ds, H, transposed=True)      # dT[c] = sum_j ds_j H[row_j]
        return dH, dT, None


class _RowSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, rg):
        a = rg.row_softmax(logits.contiguous())
        ctx.rg = rg
        ctx.save_for_backward(a)
        return a

    @staticmethod
    def backward(ctx, da):
        (a,) = ctx.saved_tensors
        return ctx.rg.row_softmax_bwd(a, da.contiguous()), None


class _ValuedSpMM(torch.autograd.Function):
    """Y = A(a) X with gradient to the entry values a AND to X (torch.sparse.mm on a sparse tensor with grad)."""

    @staticmethod
    def forward(ctx, a, X, rg):
        X = X.contiguous()
        y, _, _ = rg.spmm(a.contiguous().view(-1, 1), X)
        ctx.rg = rg
        ctx.save_for_backward(a, X)
        return y

    @staticmethod
    def backward(ctx, dY):
        a, X = ctx.saved_tensors
        dY = dY.contiguous()
        dX, _, _ = ctx.rg.spmm(a.view(-1, 1), dY, transposed=True)
        da = torch.empty(ctx.rg.nnz, 1, dtype=torch.float32, device=dY.device)
        ctx.rg.score(dY, X, da, accumulate=False)          # da_j = < dY[row_j], X[col_j] >
        return da.view(-1), dX, None


def edge_score(H, T, rg):
    return _EdgeScore.apply(H, T, rg)


def row_softmax(logits, rg):
    return _RowSoftmax.apply(logits, rg)


def valued_spmm(a, X, rg):
    return _ValuedSpMM.apply(a, X, rg)


def slice_scale(x, scale):
    y = torch.empty_like(x)
    _lib.check(_lib.load().tagrec_slice_scale_f32(_lib.ptr(x), _lib.ptr(scale), _lib.ptr(y), x.shape[0], x.shape[1],
                                                  scale.shape[1], _lib.stream_ptr()), "slice_scale")
    return y


def slice_norm_fwd(x, K, tanh=False, want_inv=True):
    y = torch.empty_like(x)
    inv = torch.empty(x.shape[0], K, dtype=torch.float32, device=x.device) if want_inv else None
    _lib.check(_lib.load().tagrec_slice_norm_fwd_f32(_lib.ptr(x), _lib.ptr(y), _lib.ptr(inv), x.shape[0], x.shape[1], K,
                                                     int(bool(tanh)), _lib.stream_ptr()), "slice_norm_fwd")
    return y, inv


def slice_norm_bwd(x_raw, inv, dz):
    dx = torch.empty_like(x_raw)
    _lib.check(_lib.load().tagrec_slice_norm_bwd_f32(_lib.ptr(x_raw), _lib.ptr(inv), _lib.ptr(dz), _lib.ptr(dx), x_raw.shape[0],
                                                     x_raw.shape[1], inv.shape[1], _lib.stream_ptr()), "slice_norm_bwd")
    return dx


class _SliceNormalize(torch.autograd.Function):
    """F.normalize(x.view(n, K, D/K), dim=2) on the [n, D] layout."""

    @staticmethod
    def forward(ctx, x, K):
        x = x.contiguous()
        y, inv = slice_norm_fwd(x, K)
        ctx.save_for_backward(x, inv)
        return y

    @staticmethod
    def backward(ctx, dz):
        x, inv = ctx.saved_tensors
        return slice_norm_bwd(x, inv, dz.contiguous()), None


def slice_normalize(x, K):
    return _SliceNormalize.apply(x, K)
