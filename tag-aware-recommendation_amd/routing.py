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
    """Structure of a symmetric adjacency + the permutation that reorders per-entry data into the entry order of
    the transposed matrix (for a symmetric structure that matrix has the same rowptr / col)."""

    def __init__(self, graph):
        if not isinstance(graph, Graph):
            raise _lib.TagrecError("RoutingGraph: needs a single Graph (row folds are not supported)")
        self.graph = graph
        n = graph.shape[0]
        deg = graph.rowptr[1:] - graph.rowptr[:-1]
        rows = torch.repeat_interleave(torch.arange(n, device=graph.device), deg)
        cols = graph.col.long()
        perm = torch.argsort(cols * n + rows)
        if graph.shape[0] != graph.shape[1] or not (torch.equal(rows[perm], cols) and torch.equal(cols[perm], rows)):
            raise _lib.TagrecError("RoutingGraph: the adjacency structure must be symmetric")
        self.rows, self.cols = rows, cols
        self.perm = perm.to(torch.int32).contiguous()
        self.nnz, self.n, self.device = int(cols.numel()), n, graph.device

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

    def spmm(self, w, x, post=None, self_add=None, b=None, b_scale=0.0, raw=True, normed=False):
        """(Y, Yn, inv): y = post * (A(w) x) + self_add + b_scale * b;  Yn / inv = per-slice L2 normalisation of y."""
        _lib.require_gpu_tensor(x, torch.float32, "route_spmm x")
        K, D = w.shape[1], x.shape[1]
        y = torch.empty_like(x) if raw else None
        yn = torch.empty_like(x) if normed else None
        inv = torch.empty(self.n, K, dtype=torch.float32, device=self.device) if normed else None
        self.graph._call("route_spmm", _lib.load().tagrec_route_spmm_f32, self.graph.handle, _lib.ptr(w), K, _lib.ptr(x),
                         _lib.ptr(post), _lib.ptr(self_add), _lib.ptr(b), float(b_scale), _lib.ptr(y), _lib.ptr(yn),
                         _lib.ptr(inv), D, _lib.stream_ptr())
        return y, yn, inv

    def score(self, h, t, logits, accumulate):
        self.graph._call("route_score", _lib.load().tagrec_route_score_f32, self.graph.handle, _lib.ptr(h), _lib.ptr(t),
                         _lib.ptr(logits), logits.shape[1], int(bool(accumulate)), h.shape[1], _lib.stream_ptr())


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
