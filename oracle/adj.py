"""Oracle: block adjacency + normalisation (numpy only).  TEST INFRASTRUCTURE.

Restates `/root/reference/model/help/adj.py`:
  * `create_ui_adj` (:7-16) / `create_uit_adj` (:19-35): symmetric block matrix
    over the node order [users | items | tags];
  * `bi_norm_laplacian` (:90-98), `si_norm_laplacian` (:101-110) and the
    `get_norm_adj` dispatch (:75-87);
  * `split_sp_mat` (:114-130) row folds.
The reference goes through scipy LIL slice assignment; here the CSR is built
directly (sort + segment-sum), which is what the device builder does too.
"""
import numpy as np


class CSR:
    """Plain CSR triple.  rowptr int64 [n_rows+1], col int32 [nnz], val float32 [nnz]."""

    def __init__(self, rowptr, col, val, shape):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float32)
        self.shape = (int(shape[0]), int(shape[1]))

    @property
    def nnz(self):
        return int(self.col.shape[0])

    def rows(self):
        """Expanded row index per stored entry (COO view)."""
        deg = np.diff(self.rowptr)
        return np.repeat(np.arange(self.shape[0], dtype=np.int64), deg)

    def to_dense(self):
        out = np.zeros(self.shape, dtype=np.float32)
        np.add.at(out, (self.rows(), self.col.astype(np.int64)), self.val)
        return out

    def transpose(self):
        return coo_to_csr(self.col.astype(np.int64), self.rows(), self.val,
                          (self.shape[1], self.shape[0]))


def coo_to_csr(rows, cols, vals, shape):
    """Sort by (row, col) and sum duplicates -- what scipy does when the
    reference converts COO -> LIL/CSR (`data/utils.py:50-53` relies on it to turn
    repeated (user, tag) pairs into integer weights)."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float32)
    n_r, n_c = int(shape[0]), int(shape[1])
    if rows.size == 0:
        return CSR(np.zeros(n_r + 1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32), shape)
    key = rows * n_c + cols
    order = np.argsort(key, kind="stable")
    key = key[order]
    vals = vals[order]
    first = np.ones(key.shape[0], dtype=bool)
    first[1:] = key[1:] != key[:-1]
    starts = np.flatnonzero(first)
    ukey = key[starts]
    # duplicates are integer counts in this code base, so fp32 segment sums are exact
    uval = np.add.reduceat(vals.astype(np.float64), starts).astype(np.float32)
    urow = ukey // n_c
    ucol = ukey % n_c
    rowptr = np.zeros(n_r + 1, dtype=np.int64)
    np.add.at(rowptr, urow + 1, 1)
    rowptr = np.cumsum(rowptr)
    return CSR(rowptr, ucol.astype(np.int32), uval, shape)


def block_adjacency(ui, ut=None, it=None):
    """`create_ui_adj` (adj.py:7-16) when ut/it are None, else `create_uit_adj`
    (adj.py:19-35).  Each block is (rows, cols, vals, (n_rows, n_cols)) in COO
    form, duplicates allowed (they are summed, as scipy does)."""
    r, c, v, (n_u, n_i) = ui
    r = np.asarray(r, np.int64)
    c = np.asarray(c, np.int64)
    v = np.asarray(v, np.float32)
    rows = [r, c + n_u]
    cols = [c + n_u, r]
    vals = [v, v]
    n = n_u + n_i
    if ut is not None:
        r2, c2, v2, (n_u2, n_t) = ut
        r3, c3, v3, (n_i3, n_t3) = it
        assert n_u2 == n_u and n_i3 == n_i and n_t3 == n_t
        r2 = np.asarray(r2, np.int64); c2 = np.asarray(c2, np.int64)
        r3 = np.asarray(r3, np.int64); c3 = np.asarray(c3, np.int64)
        v2 = np.asarray(v2, np.float32); v3 = np.asarray(v3, np.float32)
        n_ui = n
        n = n_ui + n_t
        rows += [r2, c2 + n_ui, r3 + n_u, c3 + n_ui]
        cols += [c2 + n_ui, r2, c3 + n_ui, r3 + n_u]
        vals += [v2, v2, v3, v3]
    # NB each block must be coalesced on its own first (LIL assignment of a block
    # overwrites, it does not add) -- blocks never overlap, so one global
    # coalesce is the same thing.
    return coo_to_csr(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), (n, n))


def _row_sums(csr):
    """`np.array(adj.sum(1))` on a float32 CSR: scipy keeps float32 and adds the
    row's entries in storage order (adj.py:92,103)."""
    out = np.zeros(csr.shape[0], dtype=np.float32)
    nz = np.flatnonzero(np.diff(csr.rowptr) > 0)
    if nz.size:
        # entries are small integers (counts) -> exact in fp32 whatever the order
        out[nz] = np.add.reduceat(csr.val.astype(np.float64), csr.rowptr[nz]).astype(np.float32)
    return out


def _add_identity(csr):
    n = csr.shape[0]
    rows = np.concatenate([csr.rows(), np.arange(n, dtype=np.int64)])
    cols = np.concatenate([csr.col.astype(np.int64), np.arange(n, dtype=np.int64)])
    vals = np.concatenate([csr.val, np.ones(n, np.float32)])
    return coo_to_csr(rows, cols, vals, csr.shape)


def normalise(csr, norm_type):
    """`get_norm_adj` (adj.py:75-87).

    bi_norm      : D^-1/2 A D^-1/2          (LightGCN, config.py:8-12)
    si_norm      : D^-1 A
    si_norm_self : D'^-1 (A + I)
    ngcf         : D^-1 A + I              (NGCF default, config.py:1-6)
    other        : A unchanged ("plain")
    All arithmetic in float32, inf -> 0 for isolated nodes (adj.py:94,106).
    """
    if norm_type == "si_norm_self":
        csr = _add_identity(csr)
    if norm_type not in ("bi_norm", "si_norm", "si_norm_self", "ngcf"):
        return csr
    rs = _row_sums(csr)
    rows = csr.rows()
    with np.errstate(divide="ignore"):
        if norm_type == "bi_norm":
            d = np.power(rs, np.float32(-0.5)).astype(np.float32)
            d[np.isinf(d)] = 0.0
            # scipy evaluates diag.dot(adj).dot(diag): (d[r] * a) * d[c], each product rounded to fp32
            val = (d[rows] * csr.val).astype(np.float32) * d[csr.col]
        else:
            d = np.power(rs, np.float32(-1)).astype(np.float32)
            d[np.isinf(d)] = 0.0
            val = (d[rows] * csr.val).astype(np.float32)
    out = CSR(csr.rowptr, csr.col, val.astype(np.float32), csr.shape)
    if norm_type == "ngcf":
        out = _add_identity(out)
    return out


def row_folds(n_rows, k):
    """`split_sp_mat` (adj.py:114-130): fold_len = n // k, last fold takes the remainder."""
    if k < 2:
        return [(0, n_rows)]
    fold = n_rows // k
    return [(i * fold, n_rows if i == k - 1 else (i + 1) * fold) for i in range(k)]


def slice_rows(csr, start, end):
    lo, hi = int(csr.rowptr[start]), int(csr.rowptr[end])
    return CSR(csr.rowptr[start:end + 1] - lo, csr.col[lo:hi], csr.val[lo:hi], (end - start, csr.shape[1]))
