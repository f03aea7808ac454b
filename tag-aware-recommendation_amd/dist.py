"""LightGCN over the GPUs of one node: two shardings of the node table.

FEATURE sharding (`FeatureShardedLightGCN`, the default of bench.py --gpus N): rank g holds columns
[g D/G, (g+1) D/G) of EVERY row (parameters, Adam state, activations) and the whole CSR.  The sparse product is
independent per column, so propagation needs no exchange of embeddings at all; only what reduces over a row's
columns crosses GPUs: the row norms (one all-reduce of L x N floats per step -- the chain A^k x runs on the raw
products, so no layer waits for it), the normalise-backward row dot products (another L x N floats) and the B triplet
scores.  About 50 MB per step at C2 instead of
the 3 GB of embeddings a row partition moves.

ROW sharding (`ShardedLightGCN`): the reference's `split_adj_k` folds on different GPUs, described next.

The reference is single-device; its `split_adj_k` row folds (model/help/adj.py:114-140,
158-164) compute `cat_g(A[rows_g, :] @ X)` -- exactly the 1-D row partition used here, with
the folds living on different GPUs:

  * rank g owns rows [g*R, (g+1)*R) of the [user | item | tag] table (R = ceil(N/G); the tail of
    the last shard is zero padding), the matching Adam state and CSR rows A[rows_g, :];
  * every layer, forward and backward, all-gathers the shard outputs (RCCL over xGMI, through
    torch.distributed) and runs the fused SpMM on the local rows.  The bi_norm adjacency is
    symmetric, so backward is the same pull product (A^T G)[rows_g] = A[rows_g, :] G -- no
    reduce-scatter;
  * the triplet batch is replicated: each rank contributes the batch rows it owns to a
    [6B, D] matrix, one small all-reduce completes it, the loss is computed redundantly and the
    gradient rows are scattered back to their owners.
One process per GPU; nothing here is specific to the number of ranks.

`ops` abstracts the local kernels so the host logic can be exercised on CPU ranks (gloo) in
tests/ with a test double; the product default `HipOps` is the HIP library and nothing else.
"""
import torch
import torch.distributed as dist

from . import _lib, help as H
from .graph import Graph
from .lightgcn import xavier_tables


class HipOps:
    """Local kernels = the C-ABI HIP library."""

    def make_graph(self, rowptr, col, val, shape):
        return Graph(rowptr.contiguous(), col.contiguous(), val.contiguous(), shape)

    def spmm_norm_acc(self, g, x, y, inv, acc, s):
        g.spmm_norm_acc(x, y, inv, acc, s)

    def spmm_normbwd(self, g, g_in, x_raw, inv, dz, s, out):
        g.spmm_normbwd(g_in, x_raw, inv, dz, s, out)

    def spmm_axpy(self, g, g_in, b, s, out):
        g.spmm_axpy(g_in, b, s, out)

    def rownorm_bwd(self, x_raw, inv, dz, s, out):
        n, D = x_raw.shape
        _lib.check(_lib.load().tagrec_rownorm_bwd_f32(_lib.ptr(x_raw), _lib.ptr(inv), _lib.ptr(dz), D, s,
                                                      _lib.ptr(out), 0, n, D, _lib.stream_ptr()), "rownorm_bwd")

    def bpr_fwd(self, U, I, Ur, Ir, trip, kind):
        B, D = trip.shape[0], U.shape[1]
        coef = torch.empty(B, dtype=torch.float32, device=U.device)
        partials = torch.empty(2 * ((B + 3) // 4), dtype=torch.float32, device=U.device)
        res = torch.empty(2, dtype=torch.float32, device=U.device)
        _lib.check(_lib.load().tagrec_bpr_fwd_f32(_lib.ptr(U), _lib.ptr(I), D, D, _lib.ptr(Ur), _lib.ptr(Ir), D, D,
                                                  _lib.ptr(trip), B, kind, _lib.ptr(coef), _lib.ptr(partials),
                                                  _lib.ptr(res), _lib.stream_ptr()), "bpr_fwd")
        return res, coef

    def bpr_bwd(self, U, I, Ur, Ir, trip, coef, g, dU, dI, dUr, dIr):
        B, D = trip.shape[0], U.shape[1]
        _lib.check(_lib.load().tagrec_bpr_bwd_f32(_lib.ptr(U), _lib.ptr(I), D, D, _lib.ptr(Ur), _lib.ptr(Ir), D, D,
                                                  _lib.ptr(trip), B, _lib.ptr(coef), _lib.ptr(g), 1.0, _lib.ptr(dU),
                                                  _lib.ptr(dI), _lib.ptr(dUr), _lib.ptr(dIr), _lib.stream_ptr()),
                   "bpr_bwd")


    # -- column-sharded tables -------------------------------------------------------------------
    def spmm_ss(self, g, x, y, ss):
        g._call("spmm_ss", _lib.load().tagrec_spmm_ss_f32, g.handle, _lib.ptr(x), _lib.ptr(y), _lib.ptr(ss), x.shape[1],
                _lib.stream_ptr())

    def spmm_normbwd_dot(self, g, g_in, x_raw, inv, dz, dot, s, out):
        g._call("spmm_normbwd_dot", _lib.load().tagrec_spmm_normbwd_dot_f32, g.handle, _lib.ptr(g_in), _lib.ptr(x_raw),
                _lib.ptr(inv), _lib.ptr(dz), _lib.ptr(dot), float(s), _lib.ptr(out), g_in.shape[1], _lib.stream_ptr())

    def row_scale_acc(self, y, inv, s, acc):
        _lib.check(_lib.load().tagrec_row_scale_acc_f32(_lib.ptr(y), _lib.ptr(inv), float(s), _lib.ptr(acc), y.shape[0],
                                                        y.shape[1], _lib.stream_ptr()), "row_scale_acc")

    def row_dot(self, x, inv, dz, s, out):
        _lib.check(_lib.load().tagrec_row_dot_f32(_lib.ptr(x), _lib.ptr(inv), _lib.ptr(dz), float(s), _lib.ptr(out),
                                                  x.shape[0], x.shape[1], _lib.stream_ptr()), "row_dot")

    def rownorm_bwd_dot(self, x, inv, dz, dot, s, out):
        _lib.check(_lib.load().tagrec_rownorm_bwd_dot_f32(_lib.ptr(x), _lib.ptr(inv), _lib.ptr(dz), _lib.ptr(dot), float(s),
                                                          _lib.ptr(out), x.shape[0], x.shape[1], _lib.stream_ptr()),
                   "rownorm_bwd_dot")

    # -- forward restricted to the rows the batch's loss depends on (see lightgcn.propagate_forward)
    restrict_forward = True

    def mark_rows(self, g, rows, flags):
        return g.mark_rows(rows, flags)

    def spmm_ss_rows(self, g, x, y, ss, mask):
        g._call("spmm_ss_rows", _lib.load().tagrec_spmm_ss_rows_f32, g.handle, _lib.ptr(x), _lib.ptr(y), _lib.ptr(ss), _lib.ptr(mask),
                x.shape[1], _lib.stream_ptr())

    # -- row-sparse gradients (the batch gradient reaches a few rows per hop; see EpiArgs::in_flags in csrc/spmm.hip)
    sparse_backward = True

    def row_flags(self, x):
        flags = torch.empty(x.shape[0], dtype=torch.uint8, device=x.device)
        count = torch.zeros(1, dtype=torch.int32, device=x.device)
        _lib.check(_lib.load().tagrec_row_flags_f32(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(flags), _lib.ptr(count),
                                                    _lib.stream_ptr()), "row_flags")
        return flags, count

    def spmm_normbwd_dot_sparse(self, g, g_in, fl, x_raw, inv, dz, dot, s, out, row_mask=None):
        """row_mask: only these rows can be non-zero (see lightgcn.propagate_backward); the others are zero-filled here
        and not visited by the kernel."""
        if row_mask is None:
            out_flags = torch.empty(out.shape[0], dtype=torch.uint8, device=out.device)
        else:
            out_flags = torch.zeros(out.shape[0], dtype=torch.uint8, device=out.device)
            out.zero_()
        out_count = torch.zeros(1, dtype=torch.int32, device=out.device)
        g._call("spmm_normbwd_dot", _lib.load().tagrec_spmm_normbwd_dot_sparse_f32, g.handle, _lib.ptr(g_in), _lib.ptr(fl[0]),
                _lib.ptr(fl[1]), _lib.ptr(x_raw), _lib.ptr(inv), _lib.ptr(dz), _lib.ptr(dot), float(s), _lib.ptr(out),
                _lib.ptr(out_flags), _lib.ptr(out_count), _lib.ptr(row_mask), g_in.shape[1], _lib.stream_ptr())
        return out_flags, out_count

    def spmm_axpy_sparse(self, g, g_in, fl, b, s, out):
        g.spmm_axpy_sparse(g_in, fl[0], fl[1], b, s, out)

    def bpr_dots(self, U, I, Ur, Ir, trip):
        B, D = trip.shape[0], U.shape[1]
        dots = torch.empty(B, 3, dtype=torch.float32, device=U.device)
        _lib.check(_lib.load().tagrec_bpr_dots_f32(_lib.ptr(U), _lib.ptr(I), D, D, _lib.ptr(Ur), _lib.ptr(Ir), D, D,
                                                   _lib.ptr(trip), B, _lib.ptr(dots), _lib.stream_ptr()), "bpr_dots")
        return dots


def shard_rows(n, world):
    """Rows per rank (ceil) and the padded total."""
    per = (n + world - 1) // world
    return per, per * world


def local_csr(rowptr, col, val, lo, hi, per):
    """CSR of rows [lo, hi) padded with empty rows up to `per` rows (hi may exceed the real row count)."""
    n = rowptr.numel() - 1
    lo_c, hi_c = min(lo, n), min(hi, n)
    a, b = int(rowptr[lo_c]), int(rowptr[hi_c])
    rp = rowptr[lo_c:hi_c + 1] - a
    if rp.numel() < per + 1:
        rp = torch.cat([rp, rp[-1:].expand(per + 1 - rp.numel())])
    return rp.contiguous(), col[a:b].contiguous(), val[a:b].contiguous()


class _ShardedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, model, trip):
        m = model
        x0 = table.detach()
        L, s = m.num_layer, 1.0 / (m.num_layer + 1)
        out = x0 * s
        raws, invs = [], []
        x = x0
        for _ in range(L):
            full = m.all_gather(x)
            y = torch.empty_like(x0)
            inv = torch.empty(x0.shape[0], dtype=torch.float32, device=x0.device)
            m.ops.spmm_norm_acc(m.graph, full, y, inv, out, s)
            raws.append(y)
            invs.append(inv)
            x = y
        # batch rows: [u | p | n] from the propagated table, then the same from the ego table
        B = trip.shape[0]
        rows = torch.cat([trip[:, 0], m.n_user + trip[:, 1], m.n_user + trip[:, 2]])
        mine = (rows >= m.lo) & (rows < m.hi)
        loc = rows[mine] - m.lo
        compact = torch.zeros(6 * B, x0.shape[1], dtype=torch.float32, device=x0.device)
        idx = torch.nonzero(mine).flatten()
        compact[idx] = out[loc]
        compact[idx + 3 * B] = x0[loc]
        m.all_reduce(compact)
        ar = torch.arange(B, device=x0.device)
        ctrip = torch.stack([ar, ar, ar + B], dim=1).contiguous()
        res, coef = m.ops.bpr_fwd(compact[:B], compact[B:3 * B], compact[3 * B:4 * B], compact[4 * B:], ctrip,
                                  H.loss_kind_id(m.loss_func))
        ctx.m, ctx.raws, ctx.invs = m, raws, invs
        ctx.compact, ctx.ctrip, ctx.coef, ctx.idx, ctx.loc = compact, ctrip, coef, idx, loc
        return res

    @staticmethod
    def backward(ctx, g):
        m, raws, invs = ctx.m, ctx.raws, ctx.invs
        compact, B = ctx.compact, ctx.ctrip.shape[0]
        L, s = m.num_layer, 1.0 / (m.num_layer + 1)
        dcomp = torch.zeros_like(compact)
        m.ops.bpr_bwd(compact[:B], compact[B:3 * B], compact[3 * B:4 * B], compact[4 * B:], ctx.ctrip, ctx.coef,
                      g.contiguous(), dcomp[:B], dcomp[B:3 * B], dcomp[3 * B:4 * B], dcomp[4 * B:])
        d_out = torch.zeros_like(raws[0]) if L else torch.zeros(m.per, compact.shape[1], device=compact.device)
        d_out.index_add_(0, ctx.loc, dcomp[ctx.idx])
        if L == 0:
            g0 = d_out
        else:
            gl = torch.empty_like(d_out)
            m.ops.rownorm_bwd(raws[L - 1], invs[L - 1], d_out, s, gl)
            for k in range(L - 2, -1, -1):
                full = m.all_gather(gl)
                gn = torch.empty_like(d_out)
                m.ops.spmm_normbwd(m.graph, full, raws[k], invs[k], d_out, s, gn)
                gl = gn
            full = m.all_gather(gl)
            g0 = torch.empty_like(d_out)
            m.ops.spmm_axpy(m.graph, full, d_out, s, g0)
        g0.index_add_(0, ctx.loc, dcomp[ctx.idx + 3 * B])        # L2 term on the ego rows
        ctx.raws = ctx.invs = ctx.compact = None
        return g0, None, None


class ShardedLightGCN(torch.nn.Module):
    """LightGCN with the node table row-sharded over the ranks of the default process group.
    Same `loss(batch)` / `forward()` / `parameters()` surface as `LightGCN`; every rank must call
    them with the same batch."""

    def __init__(self, data, config, rowptr, col, val, n_nodes, ops=None, group=None):
        super().__init__()
        self.ops = ops if ops is not None else HipOps()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(config["device"])
        self.num_layer = len(config["dim_layer_list"])
        self.dim_latent = config["dim_latent"]
        self.reg = config["reg"]
        self.loss_func = config["mul_loss_func"]
        self.n_user, self.n_item = data.num["user"], data.num["item"]
        self.n_nodes = int(n_nodes)
        self.per, self.n_pad = shard_rows(self.n_nodes, self.world)
        self.lo, self.hi = self.rank * self.per, (self.rank + 1) * self.per
        rp, c, v = local_csr(rowptr, col, val, self.lo, self.hi, self.per)
        self.graph = self.ops.make_graph(rp, c, v, (self.per, self.n_pad))
        num_list = [self.n_user, self.n_item] + ([data.num["tag"]] if config["use_tag"] else [])
        assert sum(num_list) == self.n_nodes
        full = xavier_tables(num_list, self.dim_latent, "cpu")         # same seed on every rank -> same table
        local = torch.zeros(self.per, self.dim_latent)
        real_hi = min(self.hi, self.n_nodes)
        if real_hi > self.lo:
            local[:real_hi - self.lo] = full[self.lo:real_hi]
        del full
        self.table = torch.nn.Parameter(local.to(self.device))

    # -- collectives (torch.distributed: RCCL on GPUs, gloo in the CPU tests) ---------------------
    def all_gather(self, x):
        if self.world == 1:
            return x.contiguous()
        full = torch.empty(self.n_pad, x.shape[1], dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(full, x.contiguous(), group=self.group)
        return full

    def all_reduce(self, x):
        if self.world > 1:
            dist.all_reduce(x, group=self.group)
        return x

    def loss(self, batch_data):
        batch_data = batch_data.to(self.device, torch.int64).contiguous()
        res = _ShardedLoss.apply(self.table, self, batch_data)
        return res[0], self.reg * res[1]

    @torch.no_grad()
    def forward(self):
        """Full propagated tables, gathered on every rank (evaluation path)."""
        L, s = self.num_layer, 1.0 / (self.num_layer + 1)
        x0 = self.table.detach()
        out = x0 * s
        x = x0
        for _ in range(L):
            y = torch.empty_like(x0)
            inv = torch.empty(x0.shape[0], dtype=torch.float32, device=x0.device)
            self.ops.spmm_norm_acc(self.graph, self.all_gather(x), y, inv, out, s)
            x = y
        full = self.all_gather(out)[:self.n_nodes]
        return full[:self.n_user], full[self.n_user:self.n_user + self.n_item]

    def gathered_table(self):
        return self.all_gather(self.table.detach())[:self.n_nodes]


# ====================================================================================== feature sharding
class _FeatureShardedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, model, trip):
        m = model
        x0 = table.detach()
        L, s, n = m.num_layer, 1.0 / (m.num_layer + 1), x0.shape[0]
        out = x0 * s
        raws = []
        x = x0
        ss = torch.zeros(max(L, 1), n, dtype=torch.float32, device=x0.device)
        # the loss reads `out` at the batch rows only: the top layer is formed on them, the one below on their neighbours
        masks = {}
        nu = m.n_user
        if (getattr(m.ops, "restrict_forward", False) and L >= 1 and trip.shape[0] * 48 <= n
                and x0.shape[1] in (8, 16, 32, 64, 128, 256)):
            rows = torch.cat([trip[:, 0], trip[:, 1] + nu, trip[:, 2] + nu])
            top = torch.zeros(n, dtype=torch.uint8, device=x0.device)
            top.index_fill_(0, rows, 1)
            masks[L - 1] = top
            if L >= 2:
                masks[L - 2] = m.ops.mark_rows(m.graph, rows, torch.zeros_like(top))
        for k in range(L):                                          # the chain A^k x0 runs on the raw products, so
            if k in masks:                                          # no layer waits for a collective
                y = torch.zeros_like(x0)
                m.ops.spmm_ss_rows(m.graph, x, y, ss[k], masks[k])
            else:
                y = torch.empty_like(x0)
                m.ops.spmm_ss(m.graph, x, y, ss[k])
            raws.append(y)
            x = y
        if L:
            m.all_reduce(ss)                                        # row norms span every rank's columns: one collective
        inv_all = 1.0 / torch.sqrt(ss).clamp_min_(1e-12)
        invs = [inv_all[k] for k in range(L)]
        for k in range(L):
            m.ops.row_scale_acc(raws[k], invs[k], s, out)
        nu, ni, B = m.n_user, m.n_item, trip.shape[0]
        U, I, Ue, Ie = out[:nu], out[nu:nu + ni], x0[:nu], x0[nu:nu + ni]
        dots = m.ops.bpr_dots(U, I, Ue, Ie, trip)
        m.all_reduce(dots)                                          # full-width scores and L2 term
        xd = dots[:, 1] - dots[:, 0]                                # neg - pos
        if m.loss_func == "logsigmoid":
            loss = -torch.nn.functional.logsigmoid(-xd).mean()
            coef = torch.sigmoid(xd)
        else:
            loss = torch.nn.functional.softplus(xd).mean()
            coef = torch.where(xd > 20.0, torch.ones_like(xd), torch.sigmoid(xd))
        res = torch.stack([loss, dots[:, 2].sum() / B])
        ctx.m, ctx.raws, ctx.invs, ctx.masks = m, raws, invs, masks
        ctx.out, ctx.x0, ctx.trip, ctx.coef = out, x0, trip, coef.contiguous()
        return res

    @staticmethod
    def backward(ctx, g):
        m, raws, invs, out, x0, trip = ctx.m, ctx.raws, ctx.invs, ctx.out, ctx.x0, ctx.trip
        L, s, n = m.num_layer, 1.0 / (m.num_layer + 1), x0.shape[0]
        nu, ni = m.n_user, m.n_item
        g = g.contiguous()
        d_out = torch.zeros_like(out)
        U, I, Ue, Ie = out[:nu], out[nu:nu + ni], x0[:nu], x0[nu:nu + ni]
        m.ops.bpr_bwd(U, I, None, None, trip, ctx.coef, g, d_out[:nu], d_out[nu:nu + ni], None, None)
        if L == 0:
            g0 = d_out
        else:
            dots = torch.empty(L, n, dtype=torch.float32, device=x0.device)
            for k in range(L):
                m.ops.row_dot(raws[k], invs[k], d_out, s, dots[k])
            m.all_reduce(dots)                                      # every layer's z . (s dZ), one collective
            gl = torch.empty_like(d_out)
            m.ops.rownorm_bwd_dot(raws[L - 1], invs[L - 1], d_out, dots[L - 1], s, gl)
            # the gradient spreads from the batch rows by one hop per product: rows still zero are not gathered
            sparse = getattr(m.ops, "sparse_backward", False) and x0.shape[1] in (8, 16, 32, 64, 128, 256)
            fl = m.ops.row_flags(gl) if sparse else None
            for k in range(L - 2, -1, -1):
                gn = torch.empty_like(d_out)
                if sparse:
                    # below the top layer of a restricted forward the result is exactly zero outside that layer's row mask
                    mask = ctx.masks.get(k) if (k + 1) in ctx.masks else None
                    fl = m.ops.spmm_normbwd_dot_sparse(m.graph, gl, fl, raws[k], invs[k], d_out, dots[k], s, gn, mask)
                else:
                    m.ops.spmm_normbwd_dot(m.graph, gl, raws[k], invs[k], d_out, dots[k], s, gn)
                gl = gn
            g0 = torch.empty_like(d_out)
            if sparse:
                m.ops.spmm_axpy_sparse(m.graph, gl, fl, d_out, s, g0)
            else:
                m.ops.spmm_axpy(m.graph, gl, d_out, s, g0)
        if m.reg != 0:
            m.ops.bpr_bwd(U, I, Ue, Ie, trip, ctx.coef, g, None, None, g0[:nu], g0[nu:nu + ni])
        ctx.raws = ctx.invs = ctx.out = None
        return g0, None, None


class FeatureShardedLightGCN(torch.nn.Module):
    """LightGCN with the embedding COLUMNS sharded over the ranks of the process group (see the module docstring).
    Same `loss(batch)` / `parameters()` surface as `LightGCN`; every rank must call `loss` with the same batch.
    Needs a symmetric adjacency (bi_norm) and dim_latent divisible by the number of ranks."""

    def __init__(self, data, config, rowptr, col, val, n_nodes, ops=None, group=None):
        super().__init__()
        self.ops = ops if ops is not None else HipOps()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(config["device"])
        self.num_layer = len(config["dim_layer_list"])
        self.dim_latent = config["dim_latent"]
        if self.dim_latent % self.world:
            raise ValueError(f"feature sharding needs dim_latent ({self.dim_latent}) divisible by the world size ({self.world})")
        self.dim_local = self.dim_latent // self.world
        self.reg = config["reg"]
        self.loss_func = config["mul_loss_func"]
        self.n_user, self.n_item = data.num["user"], data.num["item"]
        self.n_nodes = int(n_nodes)
        self.graph = self.ops.make_graph(rowptr, col, val, (self.n_nodes, self.n_nodes))
        if hasattr(self.graph, "symmetric"):
            self.graph.symmetric = True
        num_list = [self.n_user, self.n_item] + ([data.num["tag"]] if config["use_tag"] else [])
        assert sum(num_list) == self.n_nodes
        full = xavier_tables(num_list, self.dim_latent, "cpu")          # same seed on every rank -> same table
        lo = self.rank * self.dim_local
        self.table = torch.nn.Parameter(full[:, lo:lo + self.dim_local].contiguous().to(self.device))

    def all_reduce(self, x):
        if self.world > 1:
            dist.all_reduce(x, group=self.group)
        return x

    def loss(self, batch_data):
        batch_data = batch_data.to(self.device, torch.int64).contiguous()
        res = _FeatureShardedLoss.apply(self.table, self, batch_data)
        return res[0], self.reg * res[1]

    def gathered_table(self):
        """Full [N, D] table on every rank (checkpointing / evaluation)."""
        if self.world == 1:
            return self.table.detach().clone()
        parts = [torch.empty_like(self.table.data) for _ in range(self.world)]
        dist.all_gather(parts, self.table.data.contiguous(), group=self.group)
        return torch.cat(parts, dim=1)
