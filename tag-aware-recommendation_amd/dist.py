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
from .train import fused_optimizer


class HipOps:
    """Local kernels = the C-ABI HIP library."""

    def make_graph(self, rowptr, col, val, shape):
        return Graph(rowptr.contiguous(), col.contiguous(), val.contiguous(), shape)

    # -- row-sharded tables ------------------------------------------------------------------------
    row_sparse_backward = True       # gradient tables travel with one flag byte per row; zero rows are not gathered

    def row_block(self, rowptr, col, val, lo, hi, n_cols):
        """Handle on rows [lo, hi) of a CSR matrix: the row pointer is a VIEW (its entries stay absolute offsets into the
        shared col / val arrays), so the row blocks of a shard cost no copy."""
        return Graph(rowptr[lo:hi + 1], col, val, (hi - lo, n_cols))

    def mark_cols(self, g, rows, flags):
        rows = rows.contiguous()
        _lib.check(_lib.load().tagrec_graph_mark_cols_u8(g.handle, _lib.ptr(rows), rows.numel(), _lib.ptr(flags),
                                                         _lib.stream_ptr()), "graph_mark_cols")

    def spmm_listed(self, g, rows, x, out):
        from .lightgcn import spmm_listed
        spmm_listed(g, rows, x, out)

    def layer_fwd(self, g, x_full, y, inv, acc, s, row_mask=None):
        """acc None: the layer mean is not accumulated.  row_mask: only these rows are computed / written."""
        if row_mask is None and acc is not None:
            g.spmm_norm_acc(x_full, y, inv, acc, s)
        else:
            g.spmm_norm_acc_rows(x_full, y, inv, acc, s, row_mask)

    def layer_bwd(self, g, g_in, in_flags, in_count, x_raw, inv, dz, s, out, out_flags, row_mask=None, dz_flags=None):
        """in_flags with in_count None: rows flagged zero may hold anything and are never read.  dz_flags: rows of dz whose
        byte is 0 are zero (and may be unwritten)."""
        if in_flags is None and out_flags is None and row_mask is None and dz_flags is None:
            g.spmm_normbwd(g_in, x_raw, inv, dz, s, out)
            return
        cnt = torch.empty(1, dtype=torch.int32, device=out.device) if out_flags is not None else None
        g.spmm_normbwd_sparse(g_in, in_flags, in_count, x_raw, inv, dz, s, out, out_flags, cnt, row_mask=row_mask,
                              dz_flags=dz_flags)

    def last_hop(self, g, g_in, in_flags, in_count, b, s, out, b_flags=None, row_mask=None):
        if in_flags is None and b_flags is None and row_mask is None:
            g.spmm_axpy(g_in, b, s, out)
        else:
            g.spmm_axpy_sparse(g_in, in_flags, in_count, b, s, out, row_mask=row_mask, b_flags=b_flags)

    def last_hop_adam(self, g, g_in, in_flags, in_count, b, s, b_flags, p, m, v, lr, betas, eps, step):
        """`last_hop` with Adam folded in: the rows of A g_in + s b are the gradient rows of p and update p / m / v in the
        epilogue (torch.optim.Adam's arithmetic at step `step`); no gradient is written."""
        g.spmm_axpy_adam(g_in, in_flags, in_count, b, s, b_flags, p, m, v, lr, betas, eps, step)

    def rownorm_fwd(self, x):
        n, D = x.shape
        z = torch.empty_like(x)
        inv = torch.empty(n, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().tagrec_rownorm_fwd_f32(_lib.ptr(x), _lib.ptr(z), D, _lib.ptr(inv), n, D, _lib.stream_ptr()),
                   "rownorm_fwd")
        return z, inv

    def rownorm_bwd(self, x_raw, inv, dz, s, out):
        n, D = x_raw.shape
        _lib.check(_lib.load().tagrec_rownorm_bwd_f32(_lib.ptr(x_raw), _lib.ptr(inv), _lib.ptr(dz), D, s,
                                                      _lib.ptr(out), 0, n, D, _lib.stream_ptr()), "rownorm_bwd")

    def rownorm_bwd_flags(self, x_raw, inv, dz, s, out):
        n, D = x_raw.shape
        flags = torch.empty(n, dtype=torch.uint8, device=out.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=out.device)
        _lib.check(_lib.load().tagrec_rownorm_bwd_flags_f32(_lib.ptr(x_raw), _lib.ptr(inv), _lib.ptr(dz), D, s, _lib.ptr(out), 0,
                                                            n, D, _lib.ptr(flags), _lib.ptr(cnt), _lib.stream_ptr()),
                   "rownorm_bwd_flags")
        return flags

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


    # -- NGCF dense block (csrc/ngcf.hip)
    def ngcf_dense_fwd(self, nei, x, w1p, w2p, xp, inv, z_slot, ldz, row_mask=None):
        """z_slot None: the normalised slot is not written.  row_mask: rows with a zero byte are neither read nor written."""
        from .ngcf import dense_forward
        dense_forward(nei, x, w1p, w2p, xp, inv, z_slot, ldz, row_mask)

    def ngcf_dense_bwd(self, dxp, nei, x, w1p, w2p, norm, row_mask=None, dz_flags=None):
        """row_mask: rows with a zero byte are skipped (outputs unwritten there, no share in the weight gradients);
        dz_flags: rows of the concat gradient with a zero byte are zero and are not read."""
        from .ngcf import dense_backward
        return dense_backward(dxp, nei, x, w1p, w2p, norm=norm, row_mask=row_mask, dz_flags=dz_flags)

    # -- column-sharded tables -------------------------------------------------------------------
    def spmm_axpy(self, g, g_in, b, s, out):
        g.spmm_axpy(g_in, b, s, out)

    def spmm_plain(self, g, x, y, row_mask=None):
        """y = A x (rows of the mask only when given; the others are left unwritten)."""
        if row_mask is None:
            g.spmm(x, out=y)
        else:
            g.spmm_rows(x, y, row_mask)

    def spmm_flags(self, g, g_in, in_flags, in_count, out, out_flags, row_mask=None):
        g.spmm_flags(g_in, in_flags, in_count, out, out_flags, None, row_mask)

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


# Tests set this to run every collective through torch.distributed even in a group of ONE rank (where the models would
# otherwise copy): the RCCL calls -- dtypes, contiguity, async handles, stream ordering -- then run on a one-GPU box.
ALWAYS_COLLECTIVE = False
# How a block of a row shard reaches the other ranks (`_Gather.put`):
#   "collective": dist.all_gather_into_tensor -- whatever schedule RCCL picks for the communicator (a ring moves every shard
#                 over every link in turn: G - 1 hops per exchange);
#   "direct":     one send to and one receive from EVERY peer, posted together (dist.batch_isend_irecv = one grouped
#                 ncclSend / ncclRecv launch) -- the schedule SURVEY.md 8e prescribes for the xGMI full mesh: each GPU pushes
#                 its block straight over the link it shares with each peer, all G - 1 links at once, one hop.
# Same bytes, same destination layout, bit-identical result (a copy); selectable per model (config["all_gather"]) and from
# bench.py (--all-gather), default "collective".
ALL_GATHER_MODES = ("collective", "direct")


def shard_rows(n, world, n_chunks=1):
    """Rows per rank (a multiple of n_chunks, rounded up) and the padded total."""
    rc = (n + world * n_chunks - 1) // (world * n_chunks)
    return rc * n_chunks, rc * n_chunks * world


def local_csr(rowptr, col, val, lo, hi, per):
    """CSR of rows [lo, hi) padded with empty rows up to `per` rows (hi may exceed the real row count)."""
    n = rowptr.numel() - 1
    lo_c, hi_c = min(lo, n), min(hi, n)
    a, b = int(rowptr[lo_c]), int(rowptr[hi_c])
    rp = rowptr[lo_c:hi_c + 1] - a
    if rp.numel() < per + 1:
        rp = torch.cat([rp, rp[-1:].expand(per + 1 - rp.numel())])
    return rp.contiguous(), col[a:b].contiguous(), val[a:b].contiguous()


class RowPartition:
    """The 1-D fold partition of adj.py:114-140 over `world` ranks, plus the layout of a gathered table.

    Rank h owns the original rows [h R, (h+1) R) (R = `per`; the tail is zero padding).  A shard is exchanged in
    `n_chunks` row blocks of Rc = R / n_chunks rows so that the all-gather of block c can run while block c+1 is still
    being computed; `all_gather_into_tensor` wants a contiguous destination, so a gathered table is laid out
    [chunk][rank][row in chunk]: original row j = h R + c Rc + i sits at position p = c (G Rc) + h Rc + i.  The column
    indices of the local matrices are rewritten to p once, at construction; with one chunk p == j."""

    def __init__(self, n_nodes, world, n_chunks=1):
        self.n, self.world, self.n_chunks = int(n_nodes), int(world), int(n_chunks)
        self.per, self.n_pad = shard_rows(self.n, self.world, self.n_chunks)
        self.rc = self.per // self.n_chunks

    def owner(self, ids):
        return torch.div(ids, self.per, rounding_mode="floor")

    def local(self, ids):
        return ids - self.owner(ids) * self.per

    def gathered(self, ids):
        h = self.owner(ids)
        loc = ids - h * self.per
        c = torch.div(loc, self.rc, rounding_mode="floor")
        return c * (self.world * self.rc) + h * self.rc + (loc - c * self.rc)

    def chunk_rows(self, c):
        """Local rows of block c."""
        return slice(c * self.rc, (c + 1) * self.rc)

    def chunk_gathered(self, c):
        """Positions of block c (all ranks) in a gathered table."""
        return slice(c * self.world * self.rc, (c + 1) * self.world * self.rc)


def transpose_csr(rowptr, col, val, n_cols):
    """(rowptr, col, val) of the transposed matrix; torch ops only (runs on the GPU and in the CPU tests)."""
    n_rows = rowptr.numel() - 1
    deg = rowptr[1:] - rowptr[:-1]
    rows = torch.repeat_interleave(torch.arange(n_rows, device=rowptr.device), deg)
    key = col.to(torch.int64) * n_rows + rows
    key, order = torch.sort(key)
    trow = torch.div(key, n_rows, rounding_mode="floor")
    rp = torch.zeros(n_cols + 1, dtype=torch.int64, device=rowptr.device)
    torch.cumsum(torch.bincount(trow, minlength=n_cols), 0, out=rp[1:])
    return rp, (key - trow * n_rows).to(torch.int32), val[order].contiguous()


def _global_rank(group, r):
    return r if group is None else dist.get_global_rank(group, r)


class _Gather:
    """One table being all-gathered block by block.  `put(c, x_block)` starts the all-gather of block c (asynchronous:
    with RCCL it runs on the process group's stream behind an event of the producing stream, so the next block's
    kernels overlap it); `table()` waits for every block and returns the [n_pad, D] gathered table."""

    def __init__(self, model, width, dtype=torch.float32, key="table"):
        m = model
        self.m, self.key = m, key
        shape = (m.part.n_pad,) if width is None else (m.part.n_pad, width)
        self.full = m._scratch((key, width, dtype), shape, dtype)
        self.works = []
        self.keep = []

    def put(self, c, block):
        m = self.m
        dst = self.full[m.part.chunk_gathered(c)]
        if m.world == 1 and not ALWAYS_COLLECTIVE:
            dst.copy_(block)
            return
        m.comm_bytes += block.numel() * block.element_size() * (m.world - 1)
        block = block.contiguous()
        if getattr(m, "all_gather_mode", "collective") == "direct":
            rows = block.shape[0]
            if block.is_cuda and dist.get_backend(m.group) != "nccl":
                # rehearsal on a shared GPU over gloo: its point-to-point path reads the buffer from the host side without
                # looking at HIP streams (RCCL orders the transfer behind the producing kernel by itself)
                torch.cuda.current_stream(block.device).synchronize()
            dst[m.rank * rows:(m.rank + 1) * rows].copy_(block)                     # own slot: a local copy
            ops = []
            for d in range(1, m.world):                                                # peer order staggered by rank
                to, frm = (m.rank + d) % m.world, (m.rank - d) % m.world
                ops.append(dist.P2POp(dist.isend, block, _global_rank(m.group, to), group=m.group))
                ops.append(dist.P2POp(dist.irecv, dst[frm * rows:(frm + 1) * rows], _global_rank(m.group, frm), group=m.group))
            if ops:
                self.works.extend(dist.batch_isend_irecv(ops))
                self.keep.append(block)            # the send buffer must outlive the transfer
            return
        self.works.append(dist.all_gather_into_tensor(dst, block, group=m.group, async_op=True))

    def put_all(self, x):
        for c in range(self.m.part.n_chunks):
            self.put(c, x[self.m.part.chunk_rows(c)])
        return self

    def table(self):
        ev = self.m._wait_begin()
        for w in self.works:
            w.wait()
        self.works, self.keep = [], []
        self.m._wait_end(ev, "all_gather_wait")
        return self.full


class _ShardedLoss(torch.autograd.Function):
    """table shard -> [mul_loss, l2reg_loss(ego rows)] for a replicated batch; see ShardedLightGCN."""

    @staticmethod
    def forward(ctx, table, model, trip):
        m, ops, part = model, model.ops, model.part
        x0 = table.detach()
        L, s, D = m.num_layer, 1.0 / (m.num_layer + 1), x0.shape[1]
        B = trip.shape[0]
        T = 3 * B
        dev = x0.device
        rows = torch.cat([trip[:, 0], m.n_user + trip[:, 1], m.n_user + trip[:, 2]])      # original node ids, [T]
        rows_p = part.gathered(rows)
        slot = torch.nonzero(part.owner(rows) == m.rank).flatten()                       # batch slots this rank owns
        loc = rows[slot] - m.lo
        vec = D in (8, 16, 32, 64, 128, 256)
        restricted = bool(m.restrict_forward and getattr(ops, "restrict_forward", False) and L >= 1 and vec
                          and T * m.restrict_min_ratio <= part.n)
        raws, invs = [], []
        mid_mask = None
        if restricted and L >= 2:
            # rows of layer L-1 the batch depends on = the batch rows and their neighbours; the column slice lists, for
            # every node, the neighbours THIS rank owns, so the local part of the mask needs no exchange
            mid_mask = torch.zeros(part.per, dtype=torch.uint8, device=dev)
            ops.mark_cols(m.graph_cols, rows_p, mid_mask)
            mid_mask.index_fill_(0, loc, 1)
        n_pull = L - 1 if restricted else L               # layers computed as pull products on (a subset of) the local rows
        x = x0
        gat = None
        if n_pull > 0:
            gat, m._x0_prefetched = m._x0_prefetched, None       # started by the previous step's fused last hop, if any
            if gat is None:
                gat = _Gather(m, D, key="fwd").put_all(x0)
        elif m._x0_prefetched is not None:
            m.invalidate_prefetch()
        for k in range(n_pull):
            masked = restricted and k == L - 2
            xf = gat.table()
            feeds_pull = k + 1 < n_pull                    # its output is gathered for the next pull layer
            gat = _Gather(m, D, key=("fwd", k & 1)) if feeds_pull else None
            # rows outside the mask are left unwritten: every later reader is confined to the mask
            y = torch.empty_like(x0)
            inv = torch.empty(part.per, dtype=torch.float32, device=dev)
            for c in range(part.n_chunks):
                r = part.chunk_rows(c)
                ops.layer_fwd(m.graph_chunks[c], xf, y[r], inv[r], None, s, mid_mask[r] if masked else None)
                if feeds_pull:
                    gat.put(c, y[r])
            raws.append(y)
            invs.append(inv)
            x = y
        # batch rows: every rank contributes what it owns to a [*, T, D] buffer, one all-reduce completes it.  The layer
        # mean is formed here, on the batch rows this rank owns (no layer accumulates it over the whole shard).
        buf = torch.zeros(3 if restricted else 2, T, D, dtype=torch.float32, device=dev)
        ego = x0.index_select(0, loc)
        mean = ego * s
        for y, inv in zip(raws, invs):
            mean.addcmul_(y.index_select(0, loc), inv.index_select(0, loc)[:, None], value=s)
        buf[0].index_copy_(0, slot, mean)
        buf[1].index_copy_(0, slot, ego)
        if restricted:
            # top layer in push form: this rank's share of (A x)[batch rows] from the rows of x it owns
            ops.spmm_listed(m.graph_cols, rows_p, x, buf[2])
        m.all_reduce(buf, "batch_rows")
        out_b, ego_b = buf[0], buf[1]
        y_top = inv_top = None
        if restricted:
            y_top = buf[2]
            z_top, inv_top = ops.rownorm_fwd(y_top)
            out_b = out_b + s * z_top
        ar = torch.arange(B, device=dev)
        ctrip = torch.stack([ar, ar, ar + B], dim=1).contiguous()
        res, coef = ops.bpr_fwd(out_b[:B], out_b[B:], ego_b[:B], ego_b[B:], ctrip, H.loss_kind_id(m.loss_func))
        ctx.m, ctx.raws, ctx.invs, ctx.restricted, ctx.mid_mask = m, raws, invs, restricted, mid_mask
        ctx.out_b, ctx.ego_b, ctx.ctrip, ctx.coef = out_b.contiguous(), ego_b, ctrip, coef
        ctx.rows_p, ctx.slot, ctx.loc, ctx.y_top, ctx.inv_top = rows_p, slot, loc, y_top, inv_top
        ctx.shape = x0.shape
        return res

    @staticmethod
    def backward(ctx, g):
        m, raws, invs = ctx.m, ctx.raws, ctx.invs
        ops, part = m.ops, m.part
        out_b, ego_b, ctrip = ctx.out_b, ctx.ego_b, ctx.ctrip
        B = ctrip.shape[0]
        T, D = 3 * B, ctx.shape[1]
        L, s = m.num_layer, 1.0 / (m.num_layer + 1)
        dev = out_b.device
        restricted = ctx.restricted
        vec = D in (8, 16, 32, 64, 128, 256)
        sparse = vec and getattr(ops, "row_sparse_backward", False)
        d_b = torch.zeros(2, T, D, dtype=torch.float32, device=dev)        # d loss / d out_b, d loss / d ego_b (replicated)
        ops.bpr_bwd(out_b[:B], out_b[B:], ego_b[:B], ego_b[B:], ctrip, ctx.coef, g.contiguous(),
                    d_b[0][:B], d_b[0][B:], d_b[1][:B], d_b[1][B:])
        # gradient w.r.t. the layer mean on the local rows: non-zero on the batch rows this rank owns.  The restricted step
        # tells every epilogue so (dz_flags) and leaves the other rows unwritten; the all-rows step zero-fills.
        dzf = None
        if restricted:
            d_out = torch.empty(ctx.shape, dtype=torch.float32, device=dev)
            d_out.index_fill_(0, ctx.loc, 0.0)
            dzf = torch.zeros(part.per, dtype=torch.uint8, device=dev)
            dzf.index_fill_(0, ctx.loc, 1)
        else:
            d_out = torch.zeros(ctx.shape, dtype=torch.float32, device=dev)
        d_out.index_add_(0, ctx.loc, d_b[0][ctx.slot])
        n_pull = len(raws)
        if L == 0:
            g0 = d_out
        else:
            # ---- head of the chain: G^L, either on the batch rows (restricted) or on the local rows
            if restricted:
                if not sparse:
                    raise _lib.TagrecError("restricted sharded step needs the row-flag kernels")
                g_top = torch.empty(T, D, dtype=torch.float32, device=dev)
                ops.rownorm_bwd(ctx.y_top, ctx.inv_top, d_b[0], s, g_top)
                # the operand of the next product: valid at the batch rows only (slots naming one node are summed); its
                # flags are always consulted (count None), so the other rows are never read
                gfull = m._scratch(("bwd", D, 0), (part.n_pad, D), torch.float32)
                gfull.index_fill_(0, ctx.rows_p, 0.0)
                gfull.index_add_(0, ctx.rows_p, g_top)
                flags = torch.zeros(part.n_pad, dtype=torch.uint8, device=dev)
                flags.index_fill_(0, ctx.rows_p, 1)
                operand = (gfull, flags, None)
            else:
                gl = torch.empty_like(d_out)
                fl = ops.rownorm_bwd_flags(raws[L - 1], invs[L - 1], d_out, s, gl) if sparse else None
                if not sparse:
                    ops.rownorm_bwd(raws[L - 1], invs[L - 1], d_out, s, gl)
                operand = m._gather_grad(gl, fl, (L - 1) & 1)
            # ---- hops: G^k = A G^(k+1) + nb(X^k) on the local rows, gathered for the next hop
            first = n_pull - 1 if restricted else n_pull - 2          # index into raws of the layer the first hop lands on
            for k in range(first, -1, -1):
                masked = restricted and k == n_pull - 1
                gn = torch.empty_like(d_out)                           # a masked hop writes the rows of the mask only
                fo = None
                if sparse:
                    fo = (torch.zeros if masked else torch.empty)(part.per, dtype=torch.uint8, device=dev)
                gat = m._grad_gather(D, k & 1, sparse, exact_flags=masked)
                for c in range(part.n_chunks):
                    r = part.chunk_rows(c)
                    ops.layer_bwd(m.graph_chunks[c], operand[0], operand[1], operand[2], raws[k][r], invs[k][r], d_out[r], s,
                                  gn[r], fo[r] if sparse else None, ctx.mid_mask[r] if masked else None,
                                  dzf[r] if dzf is not None else None)
                    gat.put(c, gn[r], fo[r] if sparse else None)
                operand = gat.result()
            fused = fused_optimizer(m) if (restricted and m.reg == 0 and dzf is not None) else None
            if fused is not None:          # Adam in the epilogue of the last hop (Adam.fuse_into): no gradient tensor
                am, av, step = fused.fused_state(m.table)
                # the updated rows of block c are the first thing the NEXT step exchanges (the all-gather of X^0 in front of
                # its first layer): start that exchange now, behind block c's update, so it runs under the remaining blocks
                # of this hop and everything up to the next forward pass (one of the step's four table exchanges hidden)
                pre = _Gather(m, m.table.shape[1], key="fwd") if m.prefetch_x0 and L >= 2 else None
                for c in range(part.n_chunks):
                    r = part.chunk_rows(c)
                    ops.last_hop_adam(m.graph_chunks[c], operand[0], operand[1], operand[2], d_out[r], s, dzf[r],
                                      m.table.data[r], am[r], av[r], fused.lr, fused.betas, fused.eps, step)
                    if pre is not None:
                        pre.put(c, m.table.data[r])
                m._x0_prefetched = pre
                fused.fused_commit(m.table)
                ctx.raws = ctx.invs = ctx.y_top = None
                return None, None, None
            g0 = torch.empty_like(d_out)
            for c in range(part.n_chunks):
                r = part.chunk_rows(c)
                ops.last_hop(m.graph_chunks[c], operand[0], operand[1], operand[2], d_out[r], s, g0[r],
                             dzf[r] if dzf is not None else None)
        g0.index_add_(0, ctx.loc, d_b[1][ctx.slot])                      # L2 term on the ego rows this rank owns
        ctx.raws = ctx.invs = ctx.y_top = None
        return g0, None, None


class _GradGather:
    """`_Gather` of a gradient block together with its row flags; result() = (table, flags, count) for the next product."""

    def __init__(self, model, width, parity, sparse, exact_flags=False):
        self.m = model
        self.g = _Gather(model, width, key=("bwd", parity))
        self.f = _Gather(model, None, torch.uint8, key=("bwdf", parity)) if sparse else None
        self.exact = exact_flags          # the blocks were written on a row subset: rows flagged zero hold nothing valid

    def put(self, c, block, flags):
        self.g.put(c, block)
        if self.f is not None:
            self.f.put(c, flags)

    def result(self):
        full = self.g.table()
        if self.f is None:
            return full, None, None
        fl = self.f.table()
        return full, fl, (None if self.exact else fl.sum(dtype=torch.int32).reshape(1))


class ShardedLightGCN(torch.nn.Module):
    """LightGCN with the node table ROW-sharded over the ranks of the process group (module docstring).
    Same `loss(batch)` / `forward()` / `parameters()` surface as `LightGCN`; every rank must call them with the same
    batch.  Needs a symmetric adjacency (bi_norm / plain): the backward product (A^T G)[rows_g] is then the same pull
    over A[rows_g, :]; other normalisations raise.

    n_chunks: row blocks per shard for the pipelined all-gathers (default 4 with more than one rank).
    timing: set to {} to collect, per step, the milliseconds the compute stream waited for collectives."""

    restrict_min_ratio = 16          # the restricted step is used when 3 B * this <= number of nodes

    def __init__(self, data, config, rowptr, col, val, n_nodes, ops=None, group=None, n_chunks=None, symmetric=None):
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
        self.restrict_forward = bool(config.get("restrict_forward", True))
        self.all_gather_mode = config.get("all_gather", "collective")
        if self.all_gather_mode not in ALL_GATHER_MODES:
            raise _lib.TagrecError(f"config['all_gather'] must be one of {ALL_GATHER_MODES}, got {self.all_gather_mode!r}")
        if symmetric is None:
            symmetric = config.get("norm_type", "bi_norm") in ("bi_norm", "plain")
        if not symmetric:
            raise _lib.TagrecError(
                f"ShardedLightGCN: norm_type {config.get('norm_type')!r} is not symmetric; the row-sharded backward multiplies by "
                "A[rows_g, :] in place of (A^T G)[rows_g] and would be wrong (use bi_norm, or the single-GPU model)")
        self.n_user, self.n_item = data.num["user"], data.num["item"]
        self.n_nodes = int(n_nodes)
        if n_chunks is None:
            n_chunks = 4 if self.world > 1 else 1
        self.part = RowPartition(self.n_nodes, self.world, n_chunks)
        self.per, self.n_pad = self.part.per, self.part.n_pad
        self.lo, self.hi = self.rank * self.per, (self.rank + 1) * self.per
        rp, c, v = local_csr(rowptr, col, val, self.lo, self.hi, self.per)
        c = self.part.gathered(c.to(torch.int64)).to(torch.int32).contiguous()     # entry order within a row is kept
        # row slice A[rows_g, :] (pull products), as a whole and per row block; column slice A[:, rows_g] = its transpose
        self.graph = self.ops.row_block(rp, c, v, 0, self.per, self.n_pad)
        self.graph_chunks = ([self.graph] if n_chunks == 1 else
                             [self.ops.row_block(rp, c, v, k * self.part.rc, (k + 1) * self.part.rc, self.n_pad)
                              for k in range(n_chunks)])
        trp, tc, tv = transpose_csr(rp, c, v, self.n_pad)
        self.graph_cols = self.ops.row_block(trp, tc, tv, 0, self.n_pad, self.per)
        num_list = [self.n_user, self.n_item] + ([data.num["tag"]] if config["use_tag"] else [])
        assert sum(num_list) == self.n_nodes
        full = xavier_tables(num_list, self.dim_latent, "cpu")         # same seed on every rank -> same table
        local = torch.zeros(self.per, self.dim_latent)
        real_hi = min(self.hi, self.n_nodes)
        if real_hi > self.lo:
            local[:real_hi - self.lo] = full[self.lo:real_hi]
        del full
        self.table = torch.nn.Parameter(local.to(self.device))
        self._buffers_cache = {}
        self.timing = None
        self.comm_bytes = 0
        # the all-gather of X^0 for the next step, started by a fused last hop (see _ShardedLoss.backward).  It is consumed
        # by the next loss(); anything else that may change or re-gather the table drops it (train / eval switches,
        # forward(), load_state_dict, invalidate_prefetch()).  config["prefetch_x0"] = False switches it off.
        self.prefetch_x0 = bool(config.get("prefetch_x0", True))
        self._x0_prefetched = None
        self._register_load_state_dict_pre_hook(lambda *a, **k: self.invalidate_prefetch())

    def invalidate_prefetch(self):
        """Call after changing `table` by hand between two steps: the next step gathers X^0 afresh."""
        if self._x0_prefetched is not None:
            self._x0_prefetched.table()              # drain the exchange in flight before its buffer is reused
            self._x0_prefetched = None

    def train(self, mode=True):
        self.invalidate_prefetch()
        return super().train(mode)

    # -- scratch: gathered tables are reused from step to step (no 512 MB allocations inside the step) -------------
    def _scratch(self, key, shape, dtype, zero=False):
        t = self._buffers_cache.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._buffers_cache[key] = t
        return t

    # -- collectives (torch.distributed: RCCL on GPUs, gloo in the CPU tests) ---------------------
    def _wait_begin(self):
        if self.timing is None or self.device.type != "cuda":
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _wait_end(self, e0, name):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.timing.setdefault(name, []).append((e0, e1))

    def timing_ms(self):
        torch.cuda.synchronize(self.device)
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in (self.timing or {}).items()}

    def all_gather(self, x):
        """[per, D] shard -> the gathered [n_pad, D] table in ORIGINAL row order (setup / evaluation helper)."""
        if self.world == 1 and not ALWAYS_COLLECTIVE:
            return x.contiguous()
        full = torch.empty((self.n_pad,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(full, x.contiguous(), group=self.group)
        return full

    def all_reduce(self, x, name="all_reduce"):
        if self.world > 1 or ALWAYS_COLLECTIVE:
            e = self._wait_begin()
            self.comm_bytes += x.numel() * x.element_size()
            dist.all_reduce(x, group=self.group)
            self._wait_end(e, name)
        return x

    def _grad_gather(self, width, parity, sparse, exact_flags=False):
        return _GradGather(self, width, parity, sparse, exact_flags)

    def _gather_grad(self, g_local, flags_local, parity):
        gat = self._grad_gather(g_local.shape[1], parity, flags_local is not None)
        for c in range(self.part.n_chunks):
            r = self.part.chunk_rows(c)
            gat.put(c, g_local[r], flags_local[r] if flags_local is not None else None)
        return gat.result()

    def loss(self, batch_data):
        batch_data = batch_data.to(self.device, torch.int64).contiguous()
        res = _ShardedLoss.apply(self.table, self, batch_data)
        return res[0], self.reg * res[1]

    def set_fused_optimizer(self, opt):
        """`Adam.fuse_into(model)`: the restricted step with reg == 0 applies the shard's Adam update in the epilogue of its
        last hop (per row block); otherwise the optimizer gets a gradient as usual."""
        self._fused_opt = opt

    @torch.no_grad()
    def forward(self):
        """Full propagated tables, gathered on every rank (evaluation path): every layer on all rows."""
        L, s = self.num_layer, 1.0 / (self.num_layer + 1)
        self.invalidate_prefetch()
        x0 = self.table.detach()
        out = x0 * s
        x = x0
        for _ in range(L):
            xf = _Gather(self, x0.shape[1], key="fwd").put_all(x).table()
            y = torch.empty_like(x0)
            inv = torch.empty(x0.shape[0], dtype=torch.float32, device=x0.device)
            for c in range(self.part.n_chunks):
                r = self.part.chunk_rows(c)
                self.ops.layer_fwd(self.graph_chunks[c], xf, y[r], inv[r], out[r], s, None)
            x = y
        full = self.all_gather(out)[:self.n_nodes]
        return full[:self.n_user], full[self.n_user:self.n_user + self.n_item]

    def gathered_table(self):
        return self.all_gather(self.table.detach())[:self.n_nodes]


# ====================================================================================== NGCF, row-sharded
class _ShardedNgcfRestrictedLoss(torch.autograd.Function):
    """The compact restricted NGCF step (ngcf.restricted_forward / restricted_backward) on a ROW shard.

    Layers below L-1 run on all local rows, layer L-1 (neighbour sum, dense block, weight gradient) on the local part of
    the batch rows' neighbourhood -- marked through the column slice A[:, rows_g], no exchange --, layer L in push form:
    every rank sums the part of the batch rows' neighbourhoods it owns and one all-reduce of [T, .] completes the
    neighbour sums, the batch rows' own vectors and the lower layers' slots of the concatenated output; the top dense
    block and the loss are then computed redundantly on the T batch rows.  Exchanges per step: X^0 .. X^(L-1) forward,
    dN^(L-1) (flagged: valid on the masked rows) .. dN^0 backward = 2 L - 1 table exchanges (2 L + 1 in the all-rows
    step), one small all-reduce each way."""

    @staticmethod
    def forward(ctx, model, trip, table, *mats):
        from .ngcf import _wps
        m, ops, part = model, model.ops, model.part
        x0 = table.detach()
        dims = m.dims
        L, dtot = len(dims) - 1, sum(dims)
        dev = x0.device
        wps = _wps([t.detach() for t in mats])
        B = trip.shape[0]
        T = 3 * B
        rows = torch.cat([trip[:, 0], m.n_user + trip[:, 1], m.n_user + trip[:, 2]])      # original node ids, [T]
        rows_p = part.gathered(rows)
        slot = torch.nonzero(part.owner(rows) == m.rank).flatten()                       # batch slots this rank owns
        loc = rows[slot] - m.lo
        mid = None
        if L >= 2:
            mid = torch.zeros(part.per, dtype=torch.uint8, device=dev)
            ops.mark_cols(m.graph_cols, rows_p, mid)
            mid.index_fill_(0, loc, 1)
        saved, x = [], x0
        for k in range(L - 1):
            w1p, w2p = wps[k]
            mk = mid if k == L - 2 else None
            xf = _Gather(m, x.shape[1], key=("ngcf_fwd", k & 1)).put_all(x).table()
            nei = torch.empty_like(x)
            for c in range(part.n_chunks):
                r = part.chunk_rows(c)
                ops.spmm_plain(m.graph_chunks[c], xf, nei[r], mk[r] if mk is not None else None)
            xp = torch.empty(part.per, dims[k + 1], dtype=torch.float32, device=dev)
            inv = torch.empty(part.per, dtype=torch.float32, device=dev)
            ops.ngcf_dense_fwd(nei, x, w1p, w2p, xp, inv, None, 0, mk)
            saved.append((x, nei, xp, inv, w1p, w2p, mk))
            x = xp
        # batch rows: [lower slots of the concat output | the rows themselves | partial neighbour sums of the top layer]
        dlow, dl = sum(dims[:L]), dims[L - 1]
        buf = torch.zeros(T, dlow + 2 * dl, dtype=torch.float32, device=dev)
        own = torch.empty(slot.numel(), dlow + dl, dtype=torch.float32, device=dev)
        own[:, :dims[0]] = x0.index_select(0, loc)
        off = dims[0]
        for (_, _, xp, inv, _, _, _) in saved:
            torch.mul(xp.index_select(0, loc), inv.index_select(0, loc)[:, None], out=own[:, off:off + xp.shape[1]])
            off += xp.shape[1]
        own[:, dlow:] = x.index_select(0, loc)
        buf[:, :dlow + dl].index_copy_(0, slot, own)
        part_nei = torch.empty(T, dl, dtype=torch.float32, device=dev)
        ops.spmm_listed(m.graph_cols, rows_p, x, part_nei)
        buf[:, dlow + dl:] = part_nei
        m.all_reduce(buf, "batch_rows")
        xc, nc = buf[:, dlow:dlow + dl].contiguous(), buf[:, dlow + dl:].contiguous()
        w1p, w2p = wps[L - 1]
        xpc = torch.empty(T, dims[L], dtype=torch.float32, device=dev)
        invc = torch.empty(T, dtype=torch.float32, device=dev)
        out_b = torch.empty(T, dtot, dtype=torch.float32, device=dev)
        out_b[:, :dlow] = buf[:, :dlow]
        ops.ngcf_dense_fwd(nc, xc, w1p, w2p, xpc, invc, out_b[:, dlow:], dtot)
        ar = torch.arange(B, device=dev)
        ctrip = torch.stack([ar, ar, ar + B], dim=1).contiguous()
        res, coef = ops.bpr_fwd(out_b[:B], out_b[B:], out_b[:B], out_b[B:], ctrip, H.loss_kind_id(m.loss_func))
        ctx.m, ctx.saved, ctx.mid, ctx.top = m, saved, mid, (nc, xc, xpc, invc, w1p, w2p)
        ctx.out_b, ctx.ctrip, ctx.coef, ctx.rows_p, ctx.slot, ctx.loc = out_b, ctrip, coef, rows_p, slot, loc
        return res

    @staticmethod
    def backward(ctx, g):
        from .ngcf import _mat_grads
        m, saved, mid, out_b, ctrip = ctx.m, ctx.saved, ctx.mid, ctx.out_b, ctx.ctrip
        nc, xc, xpc, invc, w1p, w2p = ctx.top
        rows_p, slot, loc = ctx.rows_p, ctx.slot, ctx.loc
        ops, part, dims = m.ops, m.part, m.dims
        L = len(dims) - 1
        B = ctrip.shape[0]
        T, dtot = 3 * B, sum(dims)
        dev = out_b.device
        offs = [0]
        for d in dims:
            offs.append(offs[-1] + d)
        d_b = torch.zeros_like(out_b)                      # replicated: every rank holds the whole batch
        ops.bpr_bwd(out_b[:B], out_b[B:], out_b[:B], out_b[B:], ctrip, ctx.coef, g.contiguous(), d_b[:B], d_b[B:], d_b[:B], d_b[B:])
        dws = [None] * L
        # top layer on the T batch rows, identical on every rank: its weight gradients enter the all-reduce from rank 0 only
        d_nei_c, d_xd_c, dw1, dw2 = ops.ngcf_dense_bwd(None, nc, xc, w1p, w2p, (xpc, invc, d_b[:, offs[L]:], dtot))
        dws[L - 1] = (dw1, dw2) if m.rank == 0 else (torch.zeros_like(dw1), torch.zeros_like(dw2))
        dl = dims[L - 1]
        # the hop onto the local rows: operand = the gathered-layout table, valid on the batch rows only (always flagged)
        gfull = m._scratch(("ngcf_top", dl), (part.n_pad, dl), torch.float32)
        gfull.index_fill_(0, rows_p, 0.0)
        gfull.index_add_(0, rows_p, d_nei_c)
        fl_full = torch.zeros(part.n_pad, dtype=torch.uint8, device=dev)
        fl_full.index_fill_(0, rows_p, 1)
        bfl = torch.zeros(part.per, dtype=torch.uint8, device=dev)       # local batch rows
        bfl.index_fill_(0, loc, 1)
        b_loc = torch.empty(part.per, dl, dtype=torch.float32, device=dev)
        b_loc.index_fill_(0, loc, 0.0)
        b_loc.index_add_(0, loc, d_xd_c.index_select(0, slot))
        dx = torch.empty(part.per, dl, dtype=torch.float32, device=dev)  # valid on `mid` (every local row if L == 1)
        for c in range(part.n_chunks):
            r = part.chunk_rows(c)
            ops.last_hop(m.graph_t_chunks[c], gfull, fl_full, None, b_loc[r], 1.0, dx[r], b_flags=bfl[r],
                         row_mask=mid[r] if mid is not None else None)
        if L >= 2:
            ldz = offs[L] - offs[1]
            dzn = torch.empty(part.per, ldz, dtype=torch.float32, device=dev)          # valid on the local batch rows
            dzn.index_fill_(0, loc, 0.0)
            dzn.index_add_(0, loc, d_b[:, offs[1]:offs[L]].index_select(0, slot))
        for k in range(L - 2, -1, -1):
            x, nei, xp, inv, w1p, w2p, mk = saved[k]
            d_nei, d_xd, dw1, dw2 = ops.ngcf_dense_bwd(dx, nei, x, w1p, w2p, (xp, inv, dzn[:, offs[k + 1] - offs[1]:], ldz),
                                                       row_mask=mk, dz_flags=bfl)
            dws[k] = (dw1, dw2)
            saved[k] = None
            # dX = dX_direct + (A^T dN)[rows_g]; dN of the masked layer travels with its mask as flags (always consulted)
            if mk is not None:
                gat = _GradGather(m, d_nei.shape[1], k & 1, True, exact_flags=True)
                for c in range(part.n_chunks):
                    r = part.chunk_rows(c)
                    gat.put(c, d_nei[r], mk[r])
                gf, ff, _ = gat.result()
            else:
                gf, ff = _Gather(m, d_nei.shape[1], key=("ngcf_bwd", k & 1)).put_all(d_nei).table(), None
            dx = torch.empty_like(x)
            for c in range(part.n_chunks):
                r = part.chunk_rows(c)
                if ff is not None:
                    ops.last_hop(m.graph_t_chunks[c], gf, ff, None, d_xd[r], 1.0, dx[r], b_flags=mk[r])
                else:
                    ops.spmm_axpy(m.graph_t_chunks[c], gf, d_xd[r], 1.0, dx[r])
        dx.index_add_(0, loc, d_b[:, :dims[0]].index_select(0, slot))
        gm = _mat_grads(dws)
        flat = torch.cat([t.reshape(-1) for t in gm])
        m.all_reduce(flat, "weight_grads")
        o = 0
        for i, t in enumerate(gm):
            gm[i] = flat[o:o + t.numel()].reshape(t.shape)
            o += t.numel()
        ctx.saved = ctx.top = None
        return (None, None, dx, *gm)


class _ShardedNgcfLoss(torch.autograd.Function):
    """(table shard, W / b) -> [mul_loss, l2reg_loss(propagated rows)] for a replicated batch; see ShardedNGCF."""

    @staticmethod
    def forward(ctx, model, trip, table, *mats):
        from .ngcf import _wps
        m, ops, part = model, model.ops, model.part
        x0 = table.detach()
        dims = m.dims
        L, dtot = len(dims) - 1, sum(dims)
        dev = x0.device
        wps = _wps([t.detach() for t in mats])
        out = torch.empty(part.per, dtot, dtype=torch.float32, device=dev)
        out[:, :dims[0]] = x0
        saved, x, off = [], x0, dims[0]
        for k, (w1p, w2p) in enumerate(wps):
            xf = _Gather(m, x.shape[1], key=("ngcf_fwd", k & 1)).put_all(x).table()
            nei = torch.empty_like(x)
            for c in range(part.n_chunks):
                r = part.chunk_rows(c)
                ops.spmm_plain(m.graph_chunks[c], xf, nei[r])
            xp = torch.empty(part.per, dims[k + 1], dtype=torch.float32, device=dev)
            inv = torch.empty(part.per, dtype=torch.float32, device=dev)
            ops.ngcf_dense_fwd(nei, x, w1p, w2p, xp, inv, out[:, off:], dtot)
            saved.append((x, nei, xp, inv, w1p, w2p))
            x, off = xp, off + dims[k + 1]
        B = trip.shape[0]
        T = 3 * B
        rows = torch.cat([trip[:, 0], m.n_user + trip[:, 1], m.n_user + trip[:, 2]])
        slot = torch.nonzero(part.owner(rows) == m.rank).flatten()
        loc = rows[slot] - m.lo
        out_b = torch.zeros(T, dtot, dtype=torch.float32, device=dev)
        out_b.index_copy_(0, slot, out.index_select(0, loc))
        m.all_reduce(out_b, "batch_rows")
        ar = torch.arange(B, device=dev)
        ctrip = torch.stack([ar, ar, ar + B], dim=1).contiguous()
        res, coef = ops.bpr_fwd(out_b[:B], out_b[B:], out_b[:B], out_b[B:], ctrip, H.loss_kind_id(m.loss_func))
        ctx.m, ctx.saved, ctx.out_b, ctx.ctrip, ctx.coef, ctx.slot, ctx.loc = m, saved, out_b, ctrip, coef, slot, loc
        return res

    @staticmethod
    def backward(ctx, g):
        from .ngcf import _mat_grads
        m, saved, out_b, ctrip = ctx.m, ctx.saved, ctx.out_b, ctx.ctrip
        ops, part, dims = m.ops, m.part, m.dims
        B = ctrip.shape[0]
        dtot = sum(dims)
        dev = out_b.device
        d_b = torch.zeros_like(out_b)
        # NGCF regularises the PROPAGATED rows: both loss parts send their gradient into the same compact rows
        ops.bpr_bwd(out_b[:B], out_b[B:], out_b[:B], out_b[B:], ctrip, ctx.coef, g.contiguous(), d_b[:B], d_b[B:], d_b[:B], d_b[B:])
        d_out = torch.zeros(part.per, dtot, dtype=torch.float32, device=dev)
        d_out.index_add_(0, ctx.loc, d_b[ctx.slot])
        offs = [0]
        for d in dims:
            offs.append(offs[-1] + d)
        dws = [None] * len(saved)
        dx_next = None
        for k in range(len(saved) - 1, -1, -1):
            x, nei, xp, inv, w1p, w2p = saved[k]
            d_nei, d_xd, dw1, dw2 = ops.ngcf_dense_bwd(dx_next, nei, x, w1p, w2p, (xp, inv, d_out[:, offs[k + 1]:], dtot))
            dws[k] = (dw1, dw2)
            # dX = dX_direct + (A^T dN)[rows_g]: A = D^-1 A + I is not symmetric, the rank holds rows_g of A^T
            gf = _Gather(m, d_nei.shape[1], key=("ngcf_bwd", k & 1)).put_all(d_nei).table()
            dx = torch.empty_like(x)
            for c in range(part.n_chunks):
                r = part.chunk_rows(c)
                ops.spmm_axpy(m.graph_t_chunks[c], gf, d_xd[r], 1.0, dx[r])
            dx_next = dx
        d0 = dx_next + d_out[:, :dims[0]] if dx_next is not None else d_out[:, :dims[0]].contiguous()
        gm = _mat_grads(dws)
        if gm:                                        # every rank summed its own rows: one small all-reduce for W / b
            flat = torch.cat([t.reshape(-1) for t in gm])
            m.all_reduce(flat, "weight_grads")
            o = 0
            for i, t in enumerate(gm):
                gm[i] = flat[o:o + t.numel()].reshape(t.shape)
                o += t.numel()
        ctx.saved = None
        return (None, None, d0, *gm)


class ShardedNGCF(torch.nn.Module):
    """NGCF (model/ngcf.py) with the node table ROW-sharded over the ranks of the process group, the reference's
    `split_adj_k` folds (adj.py:114-140,158-164) living on different GPUs.  Rank g owns rows_g of the table (parameters,
    Adam state), the row slice A[rows_g, :] and the row slice of A^T (A = D^-1 A + I is not symmetric); W / b are
    replicated and their gradients all-reduced.  Per layer, forward and backward: one pipelined all-gather (of X, of dN),
    the local product, the local MFMA dense block.  `loss()` runs the compact restricted step (`_ShardedNgcfRestrictedLoss`:
    2 L - 1 table exchanges instead of 2 L + 1, the top two layers on the rows the batch depends on) when the batch is small
    against the graph, otherwise every layer on all rows.  Same `loss(batch)` / `parameters()` surface as `NGCF`; every rank
    calls with the same batch."""

    def __init__(self, data, config, rowptr, col, val, n_nodes, ops=None, group=None, n_chunks=None):
        super().__init__()
        self.ops = ops if ops is not None else HipOps()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(config["device"])
        self.dims = tuple([config["dim_latent"]] + list(config["dim_layer_list"]))
        self.reg = config["reg"]
        self.loss_func = config["mul_loss_func"]
        if config.get("agg_type", "bi_agg") != "bi_agg":
            raise NotImplementedError                     # ngcf.py:65-68
        self.n_user, self.n_item = data.num["user"], data.num["item"]
        self.n_nodes = int(n_nodes)
        if n_chunks is None:
            n_chunks = 4 if self.world > 1 else 1
        self.part = RowPartition(self.n_nodes, self.world, n_chunks)
        self.per, self.n_pad = self.part.per, self.part.n_pad
        self.lo, self.hi = self.rank * self.per, (self.rank + 1) * self.per

        def row_blocks(rp_g, c_g, v_g):
            rp, c, v = local_csr(rp_g, c_g, v_g, self.lo, self.hi, self.per)
            c = self.part.gathered(c.to(torch.int64)).to(torch.int32).contiguous()
            return [self.ops.row_block(rp, c, v, k * self.part.rc, (k + 1) * self.part.rc, self.n_pad) for k in range(n_chunks)]

        self.graph_chunks = row_blocks(rowptr, col, val)
        t_global = transpose_csr(rowptr, col, val, self.n_nodes)
        self.graph_t_chunks = row_blocks(*t_global)
        # column slice A[:, rows_g] (a CSR over all nodes in gathered order, columns = local rows) = the transpose of the
        # rank's rows of A^T: marks the local part of the batch rows' neighbourhood and carries the push-form top layer
        trp, tc, tv = local_csr(*t_global, self.lo, self.hi, self.per)
        tc = self.part.gathered(tc.to(torch.int64)).to(torch.int32).contiguous()
        crp, cc, cv = transpose_csr(trp, tc, tv, self.n_pad)
        self.graph_cols = self.ops.row_block(crp, cc, cv, 0, self.n_pad, self.per)
        self.restrict_forward = bool(config.get("restrict_forward", True))
        self.all_gather_mode = config.get("all_gather", "collective")
        if self.all_gather_mode not in ALL_GATHER_MODES:
            raise _lib.TagrecError(f"config['all_gather'] must be one of {ALL_GATHER_MODES}, got {self.all_gather_mode!r}")
        num_list = [self.n_user, self.n_item] + ([data.num["tag"]] if config["use_tag"] else [])
        assert sum(num_list) == self.n_nodes
        full = xavier_tables(num_list, self.dims[0], "cpu")            # same seed on every rank -> same table, same W / b
        local = torch.zeros(self.per, self.dims[0])
        real_hi = min(self.hi, self.n_nodes)
        if real_hi > self.lo:
            local[:real_hi - self.lo] = full[self.lo:real_hi]
        del full
        self.table = torch.nn.Parameter(local.to(self.device))
        self.mat = torch.nn.ParameterDict()
        for k in range(len(self.dims) - 1):                            # the reference's registration order (ngcf.py:45-60)
            for name in (f"W1_{k}", f"b1_{k}", f"W2_{k}", f"b2_{k}"):
                t = torch.empty(self.dims[k] if name[0] == "W" else 1, self.dims[k + 1])
                torch.nn.init.xavier_uniform_(t)
                self.mat[name] = torch.nn.Parameter(t.to(self.device))
        self._buffers_cache = {}
        self.timing = None
        self.comm_bytes = 0

    _scratch = ShardedLightGCN._scratch
    _wait_begin = ShardedLightGCN._wait_begin
    _wait_end = ShardedLightGCN._wait_end
    timing_ms = ShardedLightGCN.timing_ms
    all_gather = ShardedLightGCN.all_gather
    all_reduce = ShardedLightGCN.all_reduce
    gathered_table = ShardedLightGCN.gathered_table

    def _mats(self):
        return [self.mat[f"{n}_{k}"] for k in range(len(self.dims) - 1) for n in ("W1", "b1", "W2", "b2")]

    restrict_min_ratio = 16          # the restricted step is used when 3 B * this <= number of nodes

    def loss(self, batch_data):
        batch_data = batch_data.to(self.device, torch.int64).contiguous()
        vec = all(d in (8, 16, 32, 64, 128, 256) for d in self.dims)
        restricted = (self.restrict_forward and getattr(self.ops, "restrict_forward", False) and vec
                      and 3 * batch_data.shape[0] * self.restrict_min_ratio <= self.part.n)
        fn = _ShardedNgcfRestrictedLoss if restricted else _ShardedNgcfLoss
        res = fn.apply(self, batch_data, self.table, *self._mats())
        return res[0], self.reg * res[1]


# ====================================================================================== feature sharding
class _FeatureRestrictedLoss(torch.autograd.Function):
    """The restricted training step (lightgcn.restricted_forward / _backward) on a COLUMN slice of the table.

    Products are independent per column, so the chain runs without any exchange.  What reduces over a row's columns is
    needed at the <= 3 B batch rows only -- the layer mean is read there, and so are the normalize-backward terms -- so
    three small all-reduces replace the two [L, N] ones of the all-rows step: squared norms [L, 3B], triplet dots
    [B, 3], normalize-backward dots [L, 3B]."""

    @staticmethod
    def forward(ctx, table, model, trip):
        m, ops = model, model.ops
        x0 = table.detach()
        L, s = m.num_layer, 1.0 / (m.num_layer + 1)
        n, Dl = x0.shape
        B = trip.shape[0]
        dev = x0.device
        rows = torch.cat([trip[:, 0], m.n_user + trip[:, 1], m.n_user + trip[:, 2]])
        mid = ops.mark_rows(m.graph, rows, torch.zeros(n, dtype=torch.uint8, device=dev)) if L >= 2 else None
        raws = []
        x = x0
        for k in range(L - 1):
            y = torch.empty_like(x0)                       # rows outside the mask stay unwritten; nothing reads them
            ops.spmm_plain(m.graph, x, y, mid if k == L - 2 else None)
            raws.append(y)
            x = y
        y_top = torch.empty(rows.numel(), Dl, dtype=torch.float32, device=dev)
        ops.spmm_listed(m.graph, rows, x, y_top)
        at_rows = [y.index_select(0, rows) for y in raws] + [y_top]          # [T, Dl] per layer
        ss = torch.stack([(a * a).sum(1) for a in at_rows])                  # local columns' share of the squared norms
        m.all_reduce(ss)
        inv = 1.0 / torch.sqrt(ss).clamp_min(1e-12)                          # [L, T]
        ego_b = x0.index_select(0, rows)
        out_b = ego_b * s
        for a, iv in zip(at_rows, inv):
            out_b.addcmul_(a, iv[:, None], value=s)
        ar = torch.arange(B, device=dev)
        ctrip = torch.stack([ar, ar, ar + B], dim=1).contiguous()
        dots = ops.bpr_dots(out_b[:B], out_b[B:], ego_b[:B], ego_b[B:], ctrip)
        m.all_reduce(dots)                                                   # full-width scores and L2 term
        xd = dots[:, 1] - dots[:, 0]                                         # neg - pos
        if m.loss_func == "logsigmoid":
            loss = -torch.nn.functional.logsigmoid(-xd).mean()
            coef = torch.sigmoid(xd)
        else:
            loss = torch.nn.functional.softplus(xd).mean()
            coef = torch.where(xd > 20.0, torch.ones_like(xd), torch.sigmoid(xd))
        ctx.m, ctx.raws, ctx.mid, ctx.rows, ctx.at_rows, ctx.inv = m, raws, mid, rows, at_rows, inv
        ctx.out_b, ctx.ego_b, ctx.ctrip, ctx.coef, ctx.shape = out_b, ego_b, ctrip, coef.contiguous(), x0.shape
        return torch.stack([loss, dots[:, 2].sum() / B])

    @staticmethod
    def backward(ctx, g):
        m, ops, raws, rows, mid = ctx.m, ctx.m.ops, ctx.raws, ctx.rows, ctx.mid
        L, s = m.num_layer, 1.0 / (m.num_layer + 1)
        n, Dl = ctx.shape
        B = ctx.ctrip.shape[0]
        T = rows.numel()
        dev = ctx.out_b.device
        g = g.contiguous()
        out_b, ego_b = ctx.out_b, ctx.ego_b
        d_b = torch.zeros(2, T, Dl, dtype=torch.float32, device=dev)          # d / d out_b, d / d ego_b (local columns)
        reg = m.reg != 0
        ops.bpr_bwd(out_b[:B], out_b[B:], ego_b[:B] if reg else None, ego_b[B:] if reg else None, ctx.ctrip, ctx.coef, g,
                    d_b[0][:B], d_b[0][B:], d_b[1][:B] if reg else None, d_b[1][B:] if reg else None)
        dz = d_b[0] * s                                                       # gradient w.r.t. each normalised layer, per slot
        # normalize-backward of every layer at the batch slots: nb = inv (dz - z (z . dz)), the dot over ALL columns
        zs = [a * iv[:, None] for a, iv in zip(ctx.at_rows, ctx.inv)]
        dot = torch.stack([(z * dz).sum(1) for z in zs])
        m.all_reduce(dot)
        dot = torch.where(ctx.inv >= 1e12, torch.zeros_like(dot), dot)        # the clamp is constant where ||x|| <= eps
        nb = [iv[:, None] * (dz - z * d[:, None]) for z, iv, d in zip(zs, ctx.inv, dot)]
        tflag = torch.zeros(n, dtype=torch.uint8, device=dev)
        tflag.index_fill_(0, rows, 1)
        gcur = torch.empty(n, Dl, dtype=torch.float32, device=dev)            # G^L: valid on the batch rows only
        gcur.index_fill_(0, rows, 0.0)
        gcur.index_add_(0, rows, nb[L - 1])
        flags, count = tflag, None
        for k in range(L - 2, -1, -1):
            masked = k == L - 2
            gn = torch.empty(n, Dl, dtype=torch.float32, device=dev)
            fo = (torch.zeros if masked else torch.empty)(n, dtype=torch.uint8, device=dev)
            ops.spmm_flags(m.graph, gcur, flags, count, gn, fo, mid if masked else None)
            gn.index_add_(0, rows, nb[k])                  # (batch rows lie inside the mask: the product wrote them)
            fo.index_fill_(0, rows, 1)
            gcur, flags, count = gn, fo, None              # flags always consulted: a masked hop wrote its mask only
        fused = fused_optimizer(m) if not reg else None
        if fused is not None:              # Adam in the epilogue of the last hop (Adam.fuse_into): no gradient tensor
            b = torch.empty(n, Dl, dtype=torch.float32, device=dev)             # the ego layer's share of the mean, batch rows
            b.index_fill_(0, rows, 0.0)
            b.index_add_(0, rows, dz)
            am, av, step = fused.fused_state(m.table)
            ops.last_hop_adam(m.graph, gcur, flags, count, b, 1.0, tflag, m.table.data, am, av, fused.lr, fused.betas, fused.eps, step)
            fused.fused_commit(m.table)
            ctx.raws = ctx.at_rows = None
            return None, None, None
        g0 = torch.empty(n, Dl, dtype=torch.float32, device=dev)
        ops.spmm_flags(m.graph, gcur, flags, count, g0, None, None)
        g0.index_add_(0, rows, dz)                          # the ego layer's share of the mean
        if reg:
            g0.index_add_(0, rows, d_b[1])
        ctx.raws = ctx.at_rows = None
        return g0, None, None


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
        self.restrict_forward = bool(config.get("restrict_forward", True))
        self.all_gather_mode = config.get("all_gather", "collective")
        if self.all_gather_mode not in ALL_GATHER_MODES:
            raise _lib.TagrecError(f"config['all_gather'] must be one of {ALL_GATHER_MODES}, got {self.all_gather_mode!r}")
        if config.get("norm_type", "bi_norm") not in ("bi_norm", "plain"):
            raise _lib.TagrecError(
                f"FeatureShardedLightGCN: norm_type {config.get('norm_type')!r} is not symmetric; the backward products use "
                "the same matrix as the forward ones (use bi_norm, or the single-GPU model)")
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
        self.timing = None
        self.comm_bytes = 0

    # collectives: timed / counted like the row-sharded model's (bench.py prints them)
    _wait_begin = ShardedLightGCN._wait_begin
    _wait_end = ShardedLightGCN._wait_end
    timing_ms = ShardedLightGCN.timing_ms

    def all_reduce(self, x, name="all_reduce"):
        if self.world > 1 or ALWAYS_COLLECTIVE:
            e = self._wait_begin()
            self.comm_bytes += x.numel() * x.element_size()
            dist.all_reduce(x, group=self.group)
            self._wait_end(e, name)
        return x

    restrict_min_ratio = 16          # the restricted step is used when 3 B * this <= number of nodes

    def loss(self, batch_data):
        batch_data = batch_data.to(self.device, torch.int64).contiguous()
        restricted = (self.restrict_forward and getattr(self.ops, "restrict_forward", False) and self.num_layer >= 1
                      and self.dim_local in (8, 16, 32, 64, 128, 256)
                      and 3 * batch_data.shape[0] * self.restrict_min_ratio <= self.n_nodes)
        fn = _FeatureRestrictedLoss if restricted else _FeatureShardedLoss
        res = fn.apply(self.table, self, batch_data)
        return res[0], self.reg * res[1]

    def set_fused_optimizer(self, opt):
        """`Adam.fuse_into(model)`: the restricted step with reg == 0 applies the column slice's Adam update in the epilogue
        of its last hop; otherwise the optimizer gets a gradient as usual."""
        self._fused_opt = opt

    def gathered_table(self):
        """Full [N, D] table on every rank (checkpointing / evaluation)."""
        if self.world == 1:
            return self.table.detach().clone()
        parts = [torch.empty_like(self.table.data) for _ in range(self.world)]
        dist.all_gather(parts, self.table.data.contiguous(), group=self.group)
        return torch.cat(parts, dim=1)
