"""Host side of csrc/plan.hip: the row plan of a restricted TGCN training step -- which rows of each layer the loss of one
batch depends on (/root/reference/model/tgcn.py:236-249 reads the top layer at the batch rows; the neighbour tables of
tgcn.py:194-202 name the rows of the layer below those depend on).  A level is one mark launch, one compaction and ONE host
read of three counts; the product is, per node type, the ascending row list and the int32 position map the step's compact
tables are numbered by."""
import ctypes

import torch

from . import _lib

TYPES = ("user", "item", "tag")
_MAX_DESC = 12


class RowPlan:
    """Reusable buffers (flags, scan workspace, counts) of one model's plans; the row lists and position maps of a level are
    fresh tensors (the step keeps them until its backward pass has run)."""

    def __init__(self, sizes, device):
        self.sizes = [int(sizes[t]) for t in TYPES]
        self.device = device
        self._sizes3 = (ctypes.c_int64 * 3)(*self.sizes)
        lib = _lib.load()
        self.total = int(lib.tagrec_plan_flags_workspace(self._sizes3))
        self.off = [int(lib.tagrec_plan_segment_result(self._sizes3, t)) for t in range(3)]
        self.ws_ints = int(lib.tagrec_plan_scan_workspace(self._sizes3))
        self.flags = torch.empty(self.total, dtype=torch.uint8, device=device)
        self.ws = torch.empty(self.ws_ints, dtype=torch.int32, device=device)
        self.counts = torch.empty(4, dtype=torch.int64, device=device)
        self.row_base = [0, self.sizes[0], self.sizes[0] + self.sizes[1]]

    def level(self, descs, all_types=()):
        """descs: (idx table or None, rows pointer (int) or tensor or None, stride, n_rows, n_src, destination type name).
        Returns ({type: rows int64 ascending}, {type: pos int32 [n + 1]}) -- every type, whatever its count."""
        lib = _lib.load()
        n = len(descs)
        if n > _MAX_DESC:
            raise _lib.TagrecError("row plan: more than 12 descriptors in one level")
        keep = []                                    # tensors behind raw pointers stay alive until the launch is queued
        idx_a, rows_a = (ctypes.c_void_p * max(n, 1))(), (ctypes.c_void_p * max(n, 1))()
        stride_a, nrows_a, nsrc_a = (ctypes.c_int64 * max(n, 1))(), (ctypes.c_int64 * max(n, 1))(), (ctypes.c_int64 * max(n, 1))()
        k_a, type_a = (ctypes.c_int * max(n, 1))(), (ctypes.c_int * max(n, 1))()
        for i, (idx, rows, stride, n_rows, n_src, t) in enumerate(descs):
            if idx is not None:
                _lib.require_gpu_tensor(idx, torch.int32, "row plan: neighbour table")
                idx_a[i], k_a[i] = idx.data_ptr(), idx.shape[1]
                keep.append(idx)
            if isinstance(rows, torch.Tensor):
                if rows.dtype != torch.int64 or not rows.is_cuda:
                    raise _lib.TagrecError("row plan: row lists are int64 GPU tensors")
                keep.append(rows)
                rows_a[i] = rows.data_ptr()
            elif rows is not None:
                rows_a[i] = int(rows)
            stride_a[i], nrows_a[i], nsrc_a[i], type_a[i] = int(stride), int(n_rows), int(n_src), TYPES.index(t)
        all3 = (ctypes.c_int * 3)(*[1 if t in all_types else 0 for t in TYPES])
        s = _lib.stream_ptr()
        _lib.check(lib.tagrec_plan_mark_u8(n, idx_a, rows_a, stride_a, nrows_a, nsrc_a, k_a, type_a, all3, self._sizes3,
                                           _lib.ptr(self.flags), _lib.ptr(self.counts), s), "plan_mark")
        rows_out = torch.empty(sum(self.sizes), dtype=torch.int64, device=self.device)
        pos_out = torch.empty(self.total, dtype=torch.int32, device=self.device)
        _lib.check(lib.tagrec_plan_compact_i64(_lib.ptr(self.flags), self._sizes3, _lib.ptr(rows_out), _lib.ptr(pos_out),
                                               _lib.ptr(self.counts), _lib.ptr(self.ws), self.ws_ints, s), "plan_compact")
        c = self.counts.tolist()                     # the level's one host read
        if c[3]:
            raise IndexError(f"row plan: {c[3]} node ids out of range in the batch / the neighbour tables")
        rows = {t: rows_out[self.row_base[i]:self.row_base[i] + c[i]] for i, t in enumerate(TYPES)}
        pos = {t: pos_out[self.off[i]:self.off[i] + self.sizes[i] + 1] for i, t in enumerate(TYPES)}
        return rows, pos


def lookup(pos, rows, *, stride=1, n=None, out=None, out_stride=1):
    """out[i * out_stride] = pos[rows[i * stride] + 1] - 1 (int64): row ids -> positions in the compact table `pos` numbers.
    `rows` / `out` may be raw device pointers (then `n` is required)."""
    if n is None:
        n = rows.numel()
    if out is None:
        out = torch.empty(n, dtype=torch.int64, device=pos.device)
    rp = rows.data_ptr() if isinstance(rows, torch.Tensor) else int(rows)
    op = out.data_ptr() if isinstance(out, torch.Tensor) else int(out)
    _lib.check(_lib.load().tagrec_plan_lookup_i64(_lib.ptr(pos), ctypes.c_void_p(rp), stride, n, ctypes.c_void_p(op), out_stride,
                                                  _lib.stream_ptr()), "plan_lookup")
    return out
