"""Host side of csrc/proj.hip: the tall-skinny dense products of the TGCN step (Q = X W2, P = X[self] W1[:D] + b, their
data and weight gradients) and the look-up-table sized products, as raw kernel calls (no autograd: the step's backward is
written out by hand in tgcn_step.py).  Reference: /root/reference/model/tgcn.py:20-37."""
import torch

from . import _lib

DIMS = (16, 32, 64, 128)


def supported(*dims):
    return all(d in DIMS for d in dims)


def _strides(W, transposed):
    """(pointer tensor, stride over k, stride over c) of W used as [K, NO]: W itself (row-major [K, NO]) or the transpose of a
    row-major [NO, K'] matrix whose first K rows... -- `transposed`: W is stored [NO, ld] and used as W^T."""
    if transposed:
        return W, 1, W.stride(0)
    return W, W.stride(0), 1


def tall_mm(X1, W1, out, *, X2=None, sel=None, W2=None, w_split=0, transposed=False, b1=None, b2=None, out2=None,
            accumulate=False, n=None):
    """out (+)= [X1 | X2][sel] W + bias (see include/tagrec.h, tagrec_tall_mm_f32).  W1 / W2 may be row slices of a larger
    matrix (their row stride is honoured); transposed=True uses W^T for matrices stored [NO, K]."""
    n = (sel.numel() if sel is not None else X1.shape[0]) if n is None else n
    K = X1.shape[1] * (2 if X2 is not None else 1)
    NO = out.shape[1] * (2 if out2 is not None else 1)
    _, sk, sc = _strides(W1, transposed)
    _lib.check(_lib.load().tagrec_tall_mm_f32(_lib.ptr(X1), _lib.ptr(X2), _lib.ptr(sel), n, K, NO, _lib.ptr(W1), _lib.ptr(W2), sk, sc,
                                              w_split, _lib.ptr(b1), _lib.ptr(b2), _lib.ptr(out), _lib.ptr(out2),
                                              1 if accumulate else 0, _lib.stream_ptr()), "tall_mm")
    return out


def tall_mm_adam(X, W, g_in, param, m, v, lr, betas, eps, step, *, transposed=False):
    """The Adam update of the table `param` [n, NO] for the gradient g_in + X W (g_in may be None), applied in the product's
    epilogue: no gradient tensor is written (include/tagrec.h, tagrec_tall_mm_adam_f32)."""
    n, K = X.shape
    NO = param.shape[1]
    _, sk, sc = _strides(W, transposed)
    _lib.check(_lib.load().tagrec_tall_mm_adam_f32(_lib.ptr(X), n, K, NO, _lib.ptr(W), sk, sc, _lib.ptr(g_in), _lib.ptr(param), _lib.ptr(m),
                                                   _lib.ptr(v), lr, betas[0], betas[1], eps, step, _lib.stream_ptr()), "tall_mm_adam")


def tall_wgrad(X, dY1, dY2=None, *, dW=None, db1=None, db2=None, acc_w=False, acc_b=False):
    """dW [KI, NO] (+)= X^T [dY1 | dY2]; db1 / db2 (+)= column sums.  Returns dW (allocated when None and not accumulating)."""
    lib = _lib.load()
    n, KI = X.shape
    NO = dY1.shape[1] * (2 if dY2 is not None else 1)
    if dW is None:
        dW = torch.empty(KI, NO, dtype=torch.float32, device=X.device)
    ws_n = lib.tagrec_tall_wgrad_workspace(KI, NO)
    ws = torch.empty(ws_n, dtype=torch.float32, device=X.device)
    _lib.check(lib.tagrec_tall_wgrad_f32(_lib.ptr(X), _lib.ptr(dY1), _lib.ptr(dY2), n, KI, NO, _lib.ptr(dW), _lib.ptr(db1), _lib.ptr(db2),
                                         1 if acc_w else 0, 1 if acc_b else 0, _lib.ptr(ws), ws_n, _lib.stream_ptr()), "tall_wgrad")
    return dW


def small_mm(A, B, out=None, accumulate=False):
    """out (+)= A @ B for look-up-table sized operands; A / B may be any 2-D strided views (transposes, row slices)."""
    M, K = A.shape
    N = B.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    _lib.check(_lib.load().tagrec_small_mm_f32(_lib.ptr(A), _lib.ptr(B), _lib.ptr(out), M, N, K, A.stride(0), A.stride(1), B.stride(0),
                                               B.stride(1), 1 if accumulate else 0, _lib.stream_ptr()), "small_mm")
    return out


def row_add_at(dst, pos, src):
    """dst[pos[i]] += src[i] for DISTINCT positions (int64)."""
    _lib.check(_lib.load().tagrec_row_add_at_f32(_lib.ptr(dst), _lib.ptr(pos), _lib.ptr(src), src.shape[0], src.shape[1],
                                                 _lib.stream_ptr()), "row_add_at")
    return dst


def masked_colsum(d_out, out):
    """column sums of d_out * (out > 0): the bias gradient of a ReLU layer."""
    lib = _lib.load()
    n, D = d_out.shape
    res = torch.empty(D, dtype=torch.float32, device=d_out.device)
    ws_n = lib.tagrec_masked_colsum_workspace(D)
    ws = torch.empty(ws_n, dtype=torch.float32, device=d_out.device)
    _lib.check(lib.tagrec_masked_colsum_f32(_lib.ptr(d_out), _lib.ptr(out), n, D, _lib.ptr(res), _lib.ptr(ws), ws_n, _lib.stream_ptr()),
               "masked_colsum")
    return res
