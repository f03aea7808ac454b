"""Normalised block adjacency -> device CSR -> `Graph` handle of the HIP library.

Host-side counterpart of /root/reference/model/help/adj.py:
  `create_ui_adj` / `create_uit_adj` (:7-35), `get_norm_adj` and the two
  Laplacians (:75-110), `split_sp_mat` / `split_sp2tensor` (:114-140) and
  `creat_adj` (:38-46).  The reference goes scipy LIL -> torch sparse COO; here
  the CSR is built directly (sort + segment sum) either on the host in numpy --
  bit-identical values to the reference's scipy arithmetic -- or on the GPU with
  torch ops for graphs that are generated on the device and never visit the host.
"""
import numpy as np
import torch

from . import _lib

NORM_TYPES = ("bi_norm", "si_norm", "si_norm_self", "ngcf")


# ---------------------------------------------------------------------------------- host build
def _coalesce_host(rows, cols, vals, n_r, n_c):
    key = rows.astype(np.int64) * n_c + cols.astype(np.int64)
    order = np.argsort(key, kind="stable")
    key, vals = key[order], vals[order].astype(np.float64)
    head = np.ones(len(key), dtype=bool)
    head[1:] = key[1:] != key[:-1]
    starts = np.flatnonzero(head)
    ukey = key[starts]
    uval = np.add.reduceat(vals, starts).astype(np.float32) if len(key) else np.zeros(0, np.float32)
    deg = np.bincount(ukey // n_c, minlength=n_r)
    rowptr = np.zeros(n_r + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    return rowptr, (ukey % n_c).astype(np.int32), uval


def _block(coo):
    """Accepts a scipy COO matrix or anything with .row .col .data .shape."""
    if hasattr(coo, "tocoo"):
        coo = coo.tocoo()
    return (np.asarray(coo.row, np.int64), np.asarray(coo.col, np.int64),
            np.asarray(coo.data, np.float32), tuple(int(s) for s in coo.shape))


def block_adjacency_host(ui_adj, ut_adj=None, it_adj=None):
    """Symmetric [user | item | tag] block adjacency as CSR (adj.py:7-35).
    Repeated (row, col) pairs inside a block are summed, which is how the
    reference's tag blocks get integer weights (data/utils.py:50-53)."""
    r, c, v, (n_u, n_i) = _block(ui_adj)
    rows, cols, vals = [r, c + n_u], [c + n_u, r], [v, v]
    n = n_u + n_i
    if ut_adj is not None:
        r2, c2, v2, (_, n_t) = _block(ut_adj)
        r3, c3, v3, _ = _block(it_adj)
        rows += [r2, c2 + n, r3 + n_u, c3 + n]
        cols += [c2 + n, r2, c3 + n, r3 + n_u]
        vals += [v2, v2, v3, v3]
        n += n_t
    rowptr, col, val = _coalesce_host(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), n, n)
    return rowptr, col, val, n


def normalise_host(rowptr, col, val, n, norm_type):
    """`get_norm_adj` (adj.py:75-87) in the reference's fp32 operation order:
    (d[r] * a) * d[c] for bi_norm, d[r] * a for the row-stochastic forms."""
    def add_eye(rowptr, col, val):
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
        eye = np.arange(n, dtype=np.int64)
        return _coalesce_host(np.concatenate([rows, eye]), np.concatenate([col.astype(np.int64), eye]),
                              np.concatenate([val, np.ones(n, np.float32)]), n, n)

    if norm_type == "si_norm_self":
        rowptr, col, val = add_eye(rowptr, col, val)
    if norm_type not in NORM_TYPES:
        return rowptr, col, val
    deg = np.diff(rowptr)
    rs = np.zeros(n, dtype=np.float32)
    nz = np.flatnonzero(deg > 0)
    if nz.size:
        rs[nz] = np.add.reduceat(val.astype(np.float64), rowptr[nz]).astype(np.float32)
    rows = np.repeat(np.arange(n, dtype=np.int64), deg)
    with np.errstate(divide="ignore"):
        if norm_type == "bi_norm":
            d = np.power(rs, np.float32(-0.5)).astype(np.float32)
            d[np.isinf(d)] = 0.0
            val = (d[rows] * val).astype(np.float32) * d[col]
        else:
            d = np.power(rs, np.float32(-1)).astype(np.float32)
            d[np.isinf(d)] = 0.0
            val = (d[rows] * val).astype(np.float32)
    val = val.astype(np.float32)
    if norm_type == "ngcf":
        rowptr, col, val = add_eye(rowptr, col, val)
    return rowptr, col, val


def fold_bounds(n_rows, k):
    """`split_sp_mat` (adj.py:114-130): fold_len = n // k, the last fold takes the rest."""
    if k < 2:
        return [(0, n_rows)]
    fold = n_rows // k
    return [(i * fold, n_rows if i == k - 1 else (i + 1) * fold) for i in range(k)]


# ---------------------------------------------------------------------------------- device build
def coalesce_device(rows, cols, vals, n_r, n_c):
    """Sort-based COO -> CSR on the GPU (int64 keys); duplicates summed."""
    key = rows.to(torch.int64) * n_c + cols.to(torch.int64)
    key, order = torch.sort(key)
    vals = vals[order]
    ukey, inv = torch.unique_consecutive(key, return_inverse=True)
    if ukey.numel() != key.numel():
        uval = torch.zeros(ukey.numel(), dtype=torch.float32, device=key.device).index_add_(0, inv, vals)
    else:
        uval = vals
    urow = torch.div(ukey, n_c, rounding_mode="floor")
    deg = torch.bincount(urow, minlength=n_r)
    rowptr = torch.zeros(n_r + 1, dtype=torch.int64, device=key.device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    return rowptr, (ukey - urow * n_c).to(torch.int32), uval.to(torch.float32)


def _block_device(coo):
    """(row, col, val | None, (n_r, n_c)) of device tensors; val None = unit entries."""
    r, c, v, shape = coo
    r, c = r.to(torch.int64), c.to(torch.int64)
    if v is None:
        v = torch.ones(r.numel(), dtype=torch.float32, device=r.device)
    return r, c, v.to(torch.float32), (int(shape[0]), int(shape[1]))


def block_adjacency_device(ui, ut=None, it=None):
    """`block_adjacency_host` on the GPU (adj.py:7-35): the symmetric [user | item | tag] block adjacency as CSR from COO
    blocks given as device tensors -- (row, col, val or None, shape); `ut` / `it` hold one entry per (user, item, tag)
    assignment and repeated pairs are summed into integer weights, as the reference's scipy round trip does
    (data/tgcn_load.py:21-23).  One sort + segmented sum; nothing visits the host."""
    r, c, v, (n_u, n_i) = _block_device(ui)
    rows, cols, vals = [r, c + n_u], [c + n_u, r], [v, v]
    n = n_u + n_i
    if ut is not None:
        r2, c2, v2, (_, n_t) = _block_device(ut)
        r3, c3, v3, _ = _block_device(it)
        rows += [r2, c2 + n, r3 + n_u, c3 + n]
        cols += [c2 + n, r2, c3 + n, r3 + n_u]
        vals += [v2, v2, v3, v3]
        n += n_t
    rowptr, col, val = coalesce_device(torch.cat(rows), torch.cat(cols), torch.cat(vals), n, n)
    return rowptr, col, val, n


def normalise_device(rowptr, col, val, n, norm_type):
    """`normalise_host` on the GPU (`get_norm_adj`, adj.py:75-110), same fp32 operation order: row sums accumulated in
    fp64 and rounded once (they are sums of integer weights, so the order of the atomic adds cannot change them),
    (d[r] * a) * d[c] for bi_norm, d[r] * a for the row-stochastic forms, identity added before (si_norm_self) or after
    (ngcf) the scaling.  Values agree with the host path to the last bit wherever torch.pow and numpy's power agree
    (tests allow 1 ulp)."""
    dev = rowptr.device

    def rows_of(rowptr):
        return torch.repeat_interleave(torch.arange(n, device=dev), rowptr[1:] - rowptr[:-1])

    def add_eye(rowptr, col, val):
        eye = torch.arange(n, device=dev)
        return coalesce_device(torch.cat([rows_of(rowptr), eye]), torch.cat([col.to(torch.int64), eye]),
                               torch.cat([val, torch.ones(n, dtype=torch.float32, device=dev)]), n, n)

    if norm_type == "si_norm_self":
        rowptr, col, val = add_eye(rowptr, col, val)
    if norm_type not in NORM_TYPES:
        return rowptr, col, val.contiguous()
    r_of = rows_of(rowptr)
    rs = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, r_of, val.to(torch.float64)).to(torch.float32)
    # the power is evaluated in fp64 and rounded once: the correctly rounded fp32 value, which is what numpy's fp32 power
    # (glibc powf) returns; torch's fp32 pow takes an rsqrt shortcut for -0.5 that can be one ulp off
    if norm_type == "bi_norm":
        d = torch.pow(rs.to(torch.float64), -0.5).to(torch.float32)
        d[torch.isinf(d)] = 0.0
        val = (d[r_of] * val) * d[col.long()]
    else:
        d = torch.pow(rs.to(torch.float64), -1.0).to(torch.float32)
        d[torch.isinf(d)] = 0.0
        val = d[r_of] * val
    if norm_type == "ngcf":
        rowptr, col, val = add_eye(rowptr, col, val)
    return rowptr, col, val.contiguous()


def bipartite_norm_device(u, i, n_user, n_item, norm_type="bi_norm"):
    """User-item adjacency straight on the GPU: u, i int64 tensors of (user, item) pairs.
    Returns (rowptr, col, val, n).  = `block_adjacency_device` + `normalise_device`."""
    rowptr, col, val, n = block_adjacency_device((u, i, None, (n_user, n_item)))
    rowptr, col, val = normalise_device(rowptr, col, val, n, norm_type)
    return rowptr, col, val, n


# ---------------------------------------------------------------------------------- handle
class Graph:
    """Device CSR + the library handle built on it.  Owns the three tensors (the
    library only borrows them) and, lazily, the transposed graph for backward."""

    def __init__(self, rowptr, col, val, shape, symmetric=False, workspace=False, deferred=False):
        """workspace=True: the handle's device metadata (long-row work list, partial-sum slab) lives in a torch tensor
        owned by this object instead of hipMalloc'ed memory -- for matrices created and dropped inside a training step.
        deferred=True (with workspace): no host read of the long-row counters at creation (the launch queue is not drained);
        the work list is sized by its upper bounds."""
        self.rowptr = _lib.require_gpu_tensor(rowptr, torch.int64, "rowptr")
        self.col = _lib.require_gpu_tensor(col, torch.int32, "col")
        self.val = _lib.require_gpu_tensor(val, torch.float32, "val")
        self.shape = (int(shape[0]), int(shape[1]))
        if rowptr.numel() != self.shape[0] + 1 or col.numel() != val.numel():
            raise _lib.TagrecError("Graph: inconsistent CSR array sizes")
        self.symmetric = bool(symmetric) and self.shape[0] == self.shape[1]
        self._T = None
        self.timing = None            # set to {} to record (start, end) HIP events per kernel entry point
        self._h = _lib.c_void_p()
        lib = _lib.load()
        self._ws = None
        with torch.cuda.device(self.val.device):
            if workspace:
                nb = lib.tagrec_graph_workspace(col.numel())
                self._ws = torch.empty(nb + 256, dtype=torch.uint8, device=self.val.device)
                off = (-self._ws.data_ptr()) % 256
                self._ws_ptr = self._ws.data_ptr() + off
                create = lib.tagrec_graph_create_ws_deferred if deferred else lib.tagrec_graph_create_ws
                _lib.check(create(_lib.ctypes.byref(self._h), self.shape[0], self.shape[1], col.numel(), _lib.ptr(rowptr), _lib.ptr(col),
                                  _lib.ptr(val), _lib.c_void_p(self._ws_ptr), nb, _lib.stream_ptr()), "graph_create_ws")
            else:
                _lib.check(lib.tagrec_graph_create(_lib.ctypes.byref(self._h), self.shape[0], self.shape[1], col.numel(),
                                                   _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), _lib.stream_ptr()),
                           "graph_create")

    def like(self, col, val, n_cols):
        """A second matrix over this graph's row pointer with its own columns / values; shares the long-row work list
        (no second scan, no synchronisation).  Keeps `self` alive."""
        g = object.__new__(Graph)
        g.rowptr = self.rowptr
        g.col = _lib.require_gpu_tensor(col, torch.int32, "col")
        g.val = _lib.require_gpu_tensor(val, torch.float32, "val")
        if col.numel() != self.col.numel() or val.numel() != col.numel():
            raise _lib.TagrecError("Graph.like: needs as many stored entries as the graph it is made like")
        g.shape = (self.shape[0], int(n_cols))
        g.symmetric, g._T, g.timing, g._parent = False, None, None, self
        g._h = _lib.c_void_p()
        g._ws = None
        lib = _lib.load()
        if self._ws is not None:                       # a workspace-backed graph begets workspace-backed ones
            nb = lib.tagrec_graph_workspace(self.col.numel())
            g._ws = torch.empty(nb + 256, dtype=torch.uint8, device=self.val.device)
            g._ws_ptr = g._ws.data_ptr() + (-g._ws.data_ptr()) % 256
            _lib.check(lib.tagrec_graph_create_like_ws(_lib.ctypes.byref(g._h), self._h, g.shape[1], _lib.ptr(col), _lib.ptr(val),
                                                       _lib.c_void_p(g._ws_ptr), nb), "graph_create_like_ws")
            return g
        _lib.check(lib.tagrec_graph_create_like(_lib.ctypes.byref(g._h), self._h, g.shape[1], _lib.ptr(col),
                                                _lib.ptr(val)), "graph_create_like")
        return g

    @classmethod
    def from_host(cls, rowptr, col, val, shape, device, symmetric=False):
        dev = torch.device(device)
        return cls(torch.from_numpy(np.ascontiguousarray(rowptr, np.int64)).to(dev),
                   torch.from_numpy(np.ascontiguousarray(col, np.int32)).to(dev),
                   torch.from_numpy(np.ascontiguousarray(val, np.float32)).to(dev), shape, symmetric)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.load().tagrec_graph_destroy(h)
            except Exception:
                pass

    @property
    def handle(self):
        return self._h

    @property
    def device(self):
        return self.val.device

    @property
    def nnz(self):
        return int(self.col.numel())

    def info(self):
        out = [_lib.c_int64() for _ in range(5)]
        _lib.check(_lib.load().tagrec_graph_info(self._h, *[_lib.ctypes.byref(o) for o in out]), "graph_info")
        return dict(zip(("n_rows", "n_cols", "nnz", "n_long_rows", "n_chunks"), (o.value for o in out)))

    def transpose(self):
        """A^T as its own Graph (what the autograd backward of `split_mm` multiplies by).
        The bi_norm adjacency is symmetric, so it is its own transpose."""
        if self.symmetric:
            return self
        if self._T is None:
            deg = self.rowptr[1:] - self.rowptr[:-1]
            rows = torch.repeat_interleave(torch.arange(self.shape[0], device=self.device), deg)
            rp, c, v = coalesce_device(self.col.long(), rows, self.val, self.shape[1], self.shape[0])
            self._T = Graph(rp, c, v, (self.shape[1], self.shape[0]))
            self._T._T = self
        return self._T

    # -- raw kernel entry points (no autograd) ---------------------------------------------------
    def _call(self, name, fn, *args):
        """Run one library call; when `timing` is a dict, bracket it with events recorded on the
        stream the kernel is launched on (torch's current stream)."""
        if self.timing is None:
            _lib.check(fn(*args), name)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(fn(*args), name)
        e1.record()
        self.timing.setdefault(name, []).append((e0, e1))

    def timing_ms(self):
        """{entry point: [ms per call]} for the events recorded so far (synchronises)."""
        torch.cuda.synchronize(self.device)
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in (self.timing or {}).items()}

    def _chk_x(self, X, rows, name):
        _lib.require_gpu_tensor(X, torch.float32, name)
        if X.dim() != 2 or X.shape[0] != rows:
            raise _lib.TagrecError(f"{name}: expected [{rows}, D], got {tuple(X.shape)}")
        return int(X.shape[1])

    def spmm(self, X, out=None):
        D = self._chk_x(X, self.shape[1], "spmm X")
        if out is None:
            out = torch.empty(self.shape[0], D, dtype=torch.float32, device=X.device)
        self._chk_x(out, self.shape[0], "spmm out")
        self._call("spmm", _lib.load().tagrec_spmm_f32, self._h, _lib.ptr(X), _lib.ptr(out), D, _lib.stream_ptr())
        return out

    def spmm_norm_acc(self, X, y_raw, inv_norm, acc, acc_scale, drop_p=0.0, seed=0):
        """drop_p > 0: message dropout of the product inside the epilogue (mask = f(seed, element))."""
        D = self._chk_x(X, self.shape[1], "spmm_norm_acc X")
        self._chk_x(y_raw, self.shape[0], "spmm_norm_acc y_raw")
        self._chk_x(acc, self.shape[0], "spmm_norm_acc acc")
        _lib.require_gpu_tensor(inv_norm, torch.float32, "inv_norm")
        if inv_norm.numel() != self.shape[0] or acc.shape[1] != D or y_raw.shape[1] != D:
            raise _lib.TagrecError("spmm_norm_acc: shape mismatch")
        if drop_p > 0:
            self._call("spmm_norm_acc", _lib.load().tagrec_spmm_norm_acc_drop_f32, self._h, _lib.ptr(X), _lib.ptr(y_raw),
                       _lib.ptr(inv_norm), _lib.ptr(acc), float(acc_scale), float(drop_p), int(seed), D, _lib.stream_ptr())
            return
        self._call("spmm_norm_acc", _lib.load().tagrec_spmm_norm_acc_f32, self._h, _lib.ptr(X), _lib.ptr(y_raw),
                   _lib.ptr(inv_norm), _lib.ptr(acc), float(acc_scale), D, _lib.stream_ptr())

    def mark_rows(self, rows, flags):
        """flags[c] = 1 for every column stored in `rows` (int64 node ids) and for the rows themselves."""
        rows = rows.contiguous()
        _lib.check(_lib.load().tagrec_graph_mark_rows_u8(self._h, _lib.ptr(rows), rows.numel(), _lib.ptr(flags),
                                                         _lib.stream_ptr()), "graph_mark_rows")
        return flags

    def spmm_rows(self, X, out, row_mask):
        """`spmm` for the rows with row_mask[r] != 0 (the others of `out` are left as they are)."""
        D = self._chk_x(X, self.shape[1], "spmm X")
        self._call("spmm_rows", _lib.load().tagrec_spmm_rows_f32, self._h, _lib.ptr(X), _lib.ptr(out), _lib.ptr(row_mask), D,
                   _lib.stream_ptr())
        return out

    def spmm_norm_acc_rows(self, X, y_raw, inv_norm, acc, acc_scale, row_mask, drop_p=0.0, seed=0):
        """`spmm_norm_acc` for the rows with row_mask[r] != 0 only (the others are left as they are; None = every row).
        acc None: the layer mean is not accumulated."""
        D = self._chk_x(X, self.shape[1], "spmm_norm_acc X")
        self._call("spmm_norm_acc_rows" if row_mask is not None else "spmm_norm_acc", _lib.load().tagrec_spmm_norm_acc_rows_f32,
                   self._h, _lib.ptr(X), _lib.ptr(y_raw),
                   _lib.ptr(inv_norm), _lib.ptr(acc), float(acc_scale), _lib.ptr(row_mask), float(drop_p), int(seed), D,
                   _lib.stream_ptr())

    def spmm_normbwd(self, g_in, x_raw, inv_norm, dz, d_scale, g_out, drop_p=0.0, seed=0):
        D = self._chk_x(g_in, self.shape[1], "spmm_normbwd g_in")
        for t, nm in ((x_raw, "x_raw"), (dz, "dz"), (g_out, "g_out")):
            if self._chk_x(t, self.shape[0], "spmm_normbwd " + nm) != D:
                raise _lib.TagrecError("spmm_normbwd: width mismatch on " + nm)
        if drop_p > 0:
            self._call("spmm_normbwd", _lib.load().tagrec_spmm_normbwd_drop_f32, self._h, _lib.ptr(g_in), _lib.ptr(x_raw),
                       _lib.ptr(inv_norm), _lib.ptr(dz), float(d_scale), float(drop_p), int(seed), _lib.ptr(g_out), D,
                       _lib.stream_ptr())
            return
        self._call("spmm_normbwd", _lib.load().tagrec_spmm_normbwd_f32, self._h, _lib.ptr(g_in), _lib.ptr(x_raw),
                   _lib.ptr(inv_norm), _lib.ptr(dz), float(d_scale), _lib.ptr(g_out), D, _lib.stream_ptr())

    def spmm_normbwd_sparse(self, g_in, in_flags, in_count, x_raw, inv_norm, dz, d_scale, g_out, out_flags, out_count,
                            drop_p=0.0, seed=0, row_mask=None, dz_flags=None):
        """`spmm_normbwd` on a row-sparse g_in: rows whose in_flags byte is 0 are not gathered (same result); writes the
        flags / count of its own output when out_flags is given.  row_mask: rows whose byte is 0 are not touched at all
        (the caller knows their result is zero and has zeroed g_out / out_flags there)."""
        D = self._chk_x(g_in, self.shape[1], "spmm_normbwd g_in")
        if row_mask is not None:
            _lib.require_gpu_tensor(row_mask, torch.uint8, "row_mask")
            if row_mask.numel() != self.shape[0]:
                raise _lib.TagrecError("spmm_normbwd_sparse: row_mask must have one byte per row")
        self._call("spmm_normbwd_rows" if row_mask is not None else "spmm_normbwd", _lib.load().tagrec_spmm_normbwd_sparse_f32,
                   self._h, _lib.ptr(g_in), _lib.ptr(in_flags),
                   _lib.ptr(in_count), _lib.ptr(x_raw), _lib.ptr(inv_norm), _lib.ptr(dz), float(d_scale), float(drop_p),
                   int(seed), _lib.ptr(g_out), _lib.ptr(out_flags), _lib.ptr(out_count), _lib.ptr(row_mask), _lib.ptr(dz_flags), D,
                   _lib.stream_ptr())

    def spmm_axpy_sparse(self, g_in, in_flags, in_count, b, b_scale, g_out, row_mask=None, b_flags=None):
        """row_mask: rows whose byte is 0 are not touched (the caller knows their result and has written it).
        b_flags: rows of `b` whose byte is 0 are zero and are not read."""
        D = self._chk_x(g_in, self.shape[1], "spmm_axpy g_in")
        if row_mask is not None:
            _lib.require_gpu_tensor(row_mask, torch.uint8, "row_mask")
            if row_mask.numel() != self.shape[0]:
                raise _lib.TagrecError("spmm_axpy_sparse: row_mask must have one byte per row")
        self._call("spmm_axpy_rows" if row_mask is not None else "spmm_axpy", _lib.load().tagrec_spmm_axpy_sparse_f32, self._h,
                   _lib.ptr(g_in), _lib.ptr(in_flags), _lib.ptr(in_count), _lib.ptr(b), float(b_scale), _lib.ptr(g_out),
                   _lib.ptr(row_mask), _lib.ptr(b_flags), D, _lib.stream_ptr())

    def spmm_axpy_adam(self, g_in, in_flags, in_count, b, b_scale, b_flags, p, m, v, lr, betas, eps, step, dev=None):
        """The last hop with Adam folded in: row r of A g_in + b_scale b is the gradient of parameter row r and updates
        p / m / v in the epilogue (no gradient tensor is written).  dev = (step counter, factors) in device memory: the
        capturable form (the counter is advanced on the stream, `step` is ignored)."""
        D = self._chk_x(g_in, self.shape[1], "spmm_axpy_adam g_in")
        for t, nm in ((b, "b"), (p, "p"), (m, "m"), (v, "v")):
            if self._chk_x(t, self.shape[0], "spmm_axpy_adam " + nm) != D:
                raise _lib.TagrecError("spmm_axpy_adam: width mismatch on " + nm)
        if dev is not None:
            self._call("spmm_axpy", _lib.load().tagrec_spmm_axpy_adam_graph_f32, self._h, _lib.ptr(g_in), _lib.ptr(in_flags),
                       _lib.ptr(in_count), _lib.ptr(b), float(b_scale), _lib.ptr(b_flags), _lib.ptr(p), _lib.ptr(m), _lib.ptr(v),
                       float(lr), float(betas[0]), float(betas[1]), float(eps), _lib.ptr(dev[0]), _lib.ptr(dev[1]), D,
                       _lib.stream_ptr())
            return
        self._call("spmm_axpy", _lib.load().tagrec_spmm_axpy_adam_f32, self._h, _lib.ptr(g_in), _lib.ptr(in_flags),
                   _lib.ptr(in_count), _lib.ptr(b), float(b_scale), _lib.ptr(b_flags), _lib.ptr(p), _lib.ptr(m), _lib.ptr(v),
                   float(lr), float(betas[0]), float(betas[1]), float(eps), int(step), D, _lib.stream_ptr())

    def spmm_flags(self, g_in, in_flags, in_count, g_out, out_flags, out_count=None, row_mask=None):
        """g_out = A @ g_in on a row-sparse operand (in_count None: flags always consulted), row flags of the result written
        to out_flags; row_mask: only these rows are computed / written."""
        D = self._chk_x(g_in, self.shape[1], "spmm_flags g_in")
        self._call("spmm_flags_rows" if row_mask is not None else "spmm_flags", _lib.load().tagrec_spmm_flags_f32, self._h,
                   _lib.ptr(g_in), _lib.ptr(in_flags), _lib.ptr(in_count), _lib.ptr(g_out), _lib.ptr(out_flags),
                   _lib.ptr(out_count), _lib.ptr(row_mask), D, _lib.stream_ptr())

    def spmm_axpy(self, g_in, b, b_scale, g_out):
        D = self._chk_x(g_in, self.shape[1], "spmm_axpy g_in")
        for t, nm in ((b, "b"), (g_out, "g_out")):
            if self._chk_x(t, self.shape[0], "spmm_axpy " + nm) != D:
                raise _lib.TagrecError("spmm_axpy: width mismatch on " + nm)
        self._call("spmm_axpy", _lib.load().tagrec_spmm_axpy_f32, self._h, _lib.ptr(g_in), _lib.ptr(b),
                   float(b_scale), _lib.ptr(g_out), D, _lib.stream_ptr())


def creat_adj(data, use_tag, norm_type, split_adj_k, device):
    """Same name, arguments and result convention as the reference's `creat_adj`
    (adj.py:38-46): one `Graph`, or a list of `split_adj_k` row-fold Graphs."""
    if use_tag:
        rowptr, col, val, n = block_adjacency_host(data.ui_adj, data.ut_adj, data.it_adj)
    else:
        rowptr, col, val, n = block_adjacency_host(data.ui_adj)
    rowptr, col, val = normalise_host(rowptr, col, val, n, norm_type)
    sym = norm_type in ("bi_norm", "plain")
    if split_adj_k < 2:
        return Graph.from_host(rowptr, col, val, (n, n), device, symmetric=sym)
    out = []
    for lo, hi in fold_bounds(n, split_adj_k):
        a, b = int(rowptr[lo]), int(rowptr[hi])
        out.append(Graph.from_host(rowptr[lo:hi + 1] - a, col[a:b], val[a:b], (hi - lo, n), device))
    return out
