"""The TGCN training step as ONE hand-derived autograd node (/root/reference/model/tgcn.py:108-137, 204-249).

`TGCN.loss(batch)` in training mode: (embedding tables, every layer's parameters) -> [mul_loss, l2reg_loss] with the
forward pass and the whole backward pass written out by hand over the library's kernels, as `_PropagateBprLoss` is for
LightGCN / NGCF.  What the autograd-composed form paid for and this one does not: pairwise accumulation of the gradients
of every tensor with several readers (154 element-wise adds per step at C4), `zeros_like().index_copy_()` to lift row-subset
gradients back to table size, the zero-filled full-size gradients of column slices, host round trips to find non-zero rows.

Structure of a step (bounded fan-in is what makes it pay: k sampled neighbours per node and relation):
  * `TGCN._needed_rows`: layer l is computed on the rows the batch's loss depends on -- the <= 3 B batch rows at the top, their
    2 k neighbours per row one layer down, ... -- on COMPACT tables (a layer's output holds only those rows, in ascending
    node order; neighbour ids are renumbered into it through a position map) -- row lists and position maps come from
    csrc/plan.hip: two launches and one host read per level;
  * forward per layer: Q = X W2 on every input row, P = X[self rows] W1[:D] + b, the six neighbour attentions
    (csrc/tgcn.hip), the fused type-attention / convolution / fusion block (csrc/tgcn_fuse.hip) per node type;
  * the loss reads [ego | normalize(layer 1) | ... | normalize(layer L)] at the batch rows (compact BPR kernels);
  * backward per layer, top down: fused block (input gradients + its weight gradients), attention backward per relation
    (dP, dWT, dv written; dQ and dEj PULLED over the relation's inverted table, the second relation of a neighbour type
    adding the first one's result in the product's epilogue), then dX = dEj + dQ W2^T (+ the self rows' dXs = dT_self +
    dP W1[:D]^T at their positions) -- every gradient buffer is written once where it is formed;
  * with `T.Adam(...).fuse_into(model)` the bottom layer's last product dQ W2^T carries the Adam update of the node table
    in its epilogue (tagrec_tall_mm_adam_f32): the tables' gradients are never stored, the optimizer gets None for them.
No host read after the plan: the inverted tables are created without one (`Graph(..., deferred=True)`), so the host has
the whole step queued a few ms in (tools/c4_host_timeline.py).

Same loss and gradients as the all-rows pass (`TGCN.forward()` + triplet loss + autograd): tests/test_gpu_tgcn.py.
"""
import torch

from . import _lib, help as H, plan as PL, proj as PJ

MARKS = None      # tools/c4_host_timeline.py sets a list: (name, host time, HIP event) at a few points of the step


def _mark(name):
    if MARKS is not None:
        import time
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        MARKS.append((name, time.perf_counter(), ev))


TYPES = ("user", "item", "tag")
RELATIONS = (("user", "item"), ("user", "tag"), ("item", "user"), ("item", "tag"), ("tag", "user"), ("tag", "item"))
OTHERS = {"user": ("item", "tag"), "item": ("user", "tag"), "tag": ("user", "item")}
# slot of the neighbour type's vector inside a node's (user-side, item-side, tag-side) triple
SLOT = {"user": 0, "item": 1, "tag": 2}


def layer_params(layer):
    """The parameters of one `_Layer` in the order the step's autograd node takes / returns them."""
    ps = []
    for t in TYPES:
        a = layer.atten1[t]
        ps += [a.W_1, a.W_2, a.b, a.v]
    ps += [layer.U, layer.q, layer.p, layer.conv["bit_level"].weight, layer.conv["vec_level"]["conv_1"].weight,
           layer.conv["vec_level"]["conv_2"].weight, layer.conv["vec_level"]["conv_3"].weight, layer.Wf, layer.bf]
    return ps


N_LAYER_PARAMS = 21
SEGMENTED_DQ = True      # dQ of the pull-form attention backward: segmented sum over the sorted pairs (False: row-per-wave pull)


def _drop_seed(seed, layer, t):
    return (int(seed) * 64 + 3 * layer + SLOT[t]) & 0xFFFFFFFFFFFFFFFF


# ------------------------------------------------------------------------------------------------ raw kernel calls
def attn_fwd(P, Q, WT, v, Ej, idx, widx, out=None):
    from . import tgcn as TG
    n, A = P.shape
    k, D = idx.shape[1], Ej.shape[1]
    attn = torch.empty(n, k, dtype=torch.float32, device=P.device)
    if out is None:
        out = torch.empty(n, D, dtype=torch.float32, device=P.device)
    _lib.check(TG._timed("attn_fwd", _lib.load().tagrec_tgcn_attn_fwd_f32, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(WT), _lib.ptr(v),
                         _lib.ptr(Ej), _lib.ptr(idx), _lib.ptr(widx), n, k, D, A, _lib.ptr(attn), _lib.ptr(out),
                         _lib.stream_ptr()), "tgcn_attn_fwd")
    return out, attn


def attn_bwd(P, Q, WT, v, Ej, idx, widx, attn, d_out, dQ, dEj, dh):
    """dP, dWT, dv written; pull form (dh [n k, A] written) or scatter form (float atomics into dQ / dEj)."""
    from . import tgcn as TG
    lib = _lib.load()
    n, A = P.shape
    k, D, n_wt = idx.shape[1], Ej.shape[1], WT.shape[0]
    dP = torch.empty_like(P)
    dWT, dv = torch.empty_like(WT), torch.empty_like(v)
    ws_n = lib.tagrec_tgcn_attn_workspace(n_wt, A)
    ws = torch.empty(ws_n, dtype=torch.float32, device=P.device)
    _lib.check(TG._timed("attn_bwd", lib.tagrec_tgcn_attn_bwd_f32, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(WT), _lib.ptr(v),
                         _lib.ptr(Ej), _lib.ptr(idx), _lib.ptr(widx), _lib.ptr(attn), _lib.ptr(d_out), n, k, D, A, n_wt,
                         _lib.ptr(dP), _lib.ptr(dQ), _lib.ptr(dEj), _lib.ptr(dh), _lib.ptr(dWT), _lib.ptr(dv), _lib.ptr(ws), ws_n,
                         _lib.stream_ptr()), "tgcn_attn_bwd")
    return dP, dWT, dv


STATIC_INVERSION_MIN_SHARE = 0.05      # use the sort-free inversion when the step's rows hold at least this share of a relation's pairs


def attn_bwd_pulls(P, Q, WT, v, Ej, idx, widx, attn, d_out, addQ, addX, w_major=-1, out=None, static=None):
    """The attention backward of one relation in pull form without the dh round trip (include/tagrec.h): the relation's
    table of this step's rows is inverted on the spot (one radix sort of the n k destination ids; pads sort behind the last
    row), then  dEj (+ addX) and da  <-  attn_pull_da;  dP, dWT, dv, (ds, relu bits)  <-  attn_bwd_ds;  dQ (+ addQ)  <-
    attn_pull_dq.  out: where dEj is written (it may be addX itself).  Returns (dP, dWT, dv, dQ, dEj)."""
    from . import tgcn as TG
    from .graph import Graph
    lib = _lib.load()
    n, A = P.shape
    k, n_wt = idx.shape[1], WT.shape[0]
    n_dst, D = Ej.shape
    dev = P.device
    pair = torch.empty(n * k, dtype=torch.int32, device=dev)
    src = torch.empty(n * k, dtype=torch.int32, device=dev)
    val = torch.empty(n * k, dtype=torch.float32, device=dev)
    if static is not None:
        # the relation's pair list sorted by destination exists since start-up (the tables are static): compact the entries
        # of this step's rows in order -- no sort
        perm32, dest32, pos_src, pos_dst = static
        skey = torch.empty(n * k, dtype=torch.int32, device=dev)
        fw_n = lib.tagrec_inv_filter_workspace(perm32.numel())
        fw = torch.empty(fw_n, dtype=torch.int32, device=dev)
        _lib.check(TG._timed("attn_invert", lib.tagrec_inv_filter_i32, _lib.ptr(perm32), _lib.ptr(dest32), perm32.numel(), k,
                             _lib.ptr(pos_src), _lib.ptr(pos_dst), _lib.ptr(attn), n_dst, n * k, _lib.ptr(skey), _lib.ptr(pair),
                             _lib.ptr(src), _lib.ptr(val), _lib.ptr(fw), fw_n, _lib.stream_ptr()), "inv_filter")
    else:
        key = torch.empty(n * k, dtype=torch.int32, device=dev)
        _lib.check(lib.tagrec_attn_keys_i32(_lib.ptr(idx), n * k, n_dst, _lib.ptr(key), _lib.stream_ptr()), "attn_keys")
        skey, order = torch.sort(key, stable=True)
        _lib.check(lib.tagrec_attn_invert_fill(_lib.ptr(order), _lib.ptr(attn), k, n * k, _lib.ptr(pair), _lib.ptr(src), _lib.ptr(val),
                                               _lib.stream_ptr()), "attn_invert_fill")
    rowptr = torch.searchsorted(skey, torch.arange(n_dst + 1, dtype=torch.int32, device=dev))
    inv = Graph(rowptr, src, val, (n_dst, n), workspace=True, deferred=True)
    da = torch.empty(n * k, dtype=torch.float32, device=dev)
    dEj = out if out is not None else torch.empty(n_dst, D, dtype=torch.float32, device=dev)      # (may alias addX: element-wise)
    _lib.check(TG._timed("attn_pull_da", lib.tagrec_attn_pull_da_f32, inv.handle, _lib.ptr(pair), _lib.ptr(d_out), _lib.ptr(Ej),
                         _lib.ptr(addX), _lib.ptr(dEj), _lib.ptr(da), D, _lib.stream_ptr()), "attn_pull_da")
    dP = torch.empty_like(P)
    dWT, dv = torch.empty_like(WT), torch.empty_like(v)
    comp = torch.empty(n * k, 2, dtype=torch.float32, device=dev)
    ws_n = lib.tagrec_tgcn_attn_workspace(n_wt, A)
    ws = torch.empty(ws_n, dtype=torch.float32, device=dev)
    _lib.check(TG._timed("attn_bwd", lib.tagrec_tgcn_attn_bwd_ds_f32, _lib.ptr(P), _lib.ptr(Q), _lib.ptr(WT), _lib.ptr(v), _lib.ptr(idx),
                         _lib.ptr(widx), _lib.ptr(attn), _lib.ptr(da), n, k, A, n_wt, int(w_major), _lib.ptr(dP), _lib.ptr(comp), _lib.ptr(dWT),
                         _lib.ptr(dv), _lib.ptr(ws), ws_n, _lib.stream_ptr()), "tgcn_attn_bwd_ds")
    if SEGMENTED_DQ:         # balanced segmented sum over the sorted pair list, float atomics at segment ends (adds into addQ)
        dQ = addQ if addQ is not None else torch.zeros(n_dst, A, dtype=torch.float32, device=dev)
        _lib.check(TG._timed("attn_pull_dq", lib.tagrec_attn_seg_dq_f32, _lib.ptr(skey), _lib.ptr(pair), n * k, n_dst, _lib.ptr(comp),
                             _lib.ptr(v), A, _lib.ptr(dQ), _lib.stream_ptr()), "attn_seg_dq")
    else:                    # row-per-wave pull on the inverted table (deterministic order)
        dQ = torch.empty(n_dst, A, dtype=torch.float32, device=dev)
        inv = inv.like(pair, val, n * k)                    # the same rows, colidx = pair ids
        _lib.check(TG._timed("attn_pull_dq", lib.tagrec_attn_pull_dq_f32, inv.handle, _lib.ptr(comp), _lib.ptr(v), A, _lib.ptr(addQ),
                             _lib.ptr(dQ), _lib.stream_ptr()), "attn_pull_dq")
    return dP, dWT, dv, dQ, dEj


def fuse_fwd(t0, t1, t2, dp, drop=None):
    """drop = (p, (seed_user, seed_item, seed_tag), (rows_user, rows_item, rows_tag) node ids or None, lo_item, lo_tag): message
    dropout of the output in the kernel's epilogue (the mask keyed by node id); the returned `out` is the dropped output."""
    from . import tgcn as TG
    U, q, p, wb, w1, w2, w3, Wf, bf = dp
    n, D = t0.shape
    out = torch.empty(n, Wf.shape[1], dtype=torch.float32, device=t0.device)
    bw = torch.empty(n, 3, dtype=torch.float32, device=t0.device)
    if drop is not None:
        pk, seeds, rows, lo1, lo2 = drop
        seeds3 = (_lib.ctypes.c_uint64 * 3)(*[int(x) for x in seeds])
        _lib.check(TG._timed("fuse_fwd", _lib.load().tagrec_tgcn_fuse_fwd_drop_f32, _lib.ptr(t0), _lib.ptr(t1), _lib.ptr(t2), n, D,
                             Wf.shape[1], U.shape[1], wb.shape[0], w1.shape[0], *[_lib.ptr(a) for a in dp], float(pk), seeds3,
                             _lib.ptr(rows[0]), _lib.ptr(rows[1]), _lib.ptr(rows[2]), int(lo1), int(lo2), _lib.ptr(bw),
                             _lib.ptr(out), _lib.stream_ptr()), "tgcn_fuse_fwd_drop")
        return out, bw
    _lib.check(TG._timed("fuse_fwd", _lib.load().tagrec_tgcn_fuse_fwd_f32, _lib.ptr(t0), _lib.ptr(t1), _lib.ptr(t2), n, D,
                         Wf.shape[1], U.shape[1], wb.shape[0], w1.shape[0], *[_lib.ptr(a) for a in dp], _lib.ptr(bw),
                         _lib.ptr(out), _lib.stream_ptr()), "tgcn_fuse_fwd")
    return out, bw


def _dense_views(ps):
    """(U, q, p, wb [C,3], w1 [V,D], w2 [V,2D], w3 [V,3D], Wf, bf) as the fused kernels take them."""
    U, q, p, wb, w1, w2, w3, Wf, bf = ps
    return (U.contiguous(), q.reshape(-1).contiguous(), p.reshape(-1).contiguous(), wb.reshape(wb.shape[0], 3).contiguous(),
            w1.reshape(w1.shape[0], -1).contiguous(), w2.reshape(w2.shape[0], -1).contiguous(),
            w3.reshape(w3.shape[0], -1).contiguous(), Wf.contiguous(), bf.reshape(-1).contiguous())


def tall_wgrad(X, dY):
    """X^T dY for X [n, d], dY [n, a] with n in the millions: slab products + one sum (see tgcn._TallMM)."""
    n = X.shape[0]
    if n < 65536:
        return X.t() @ dY
    S = 256
    m = n // S * S
    dW = torch.bmm(X[:m].view(S, m // S, -1).transpose(1, 2), dY[:m].view(S, m // S, -1)).sum(0)
    if m < n:
        dW = dW + X[m:].t() @ dY[m:]
    return dW


# ------------------------------------------------------------------------------------------------ the step
def step_forward(model, batch, embs, ew, layers_ps, training_drop):
    """Forward of the restricted step.  embs: {type: table}; layers_ps: per layer the 21 detached parameters.
    The three node types of a layer share ONE set of merged buffers -- rows [users | items | tags] of the layer's row
    subsets -- so the fused dense block (and its backward and weight-gradient kernels) is launched once per layer, not
    once per type: the small launches of the upper layers (a few thousand rows: a fraction of one wave of blocks) fill
    more of the chip, and the next layer's input tables are row slices of the merged output.
    Returns (res [2], state)."""
    dev = batch.device
    lib = _lib.load()
    _mark("step start")
    need = model._needed_rows(batch)
    need_pos = model._need_pos
    _mark("plan done")
    sizes = {"user": model.num_user, "item": model.num_item, "tag": model.num_tag}
    L = len(layers_ps)
    ewp = torch.cat([ew.new_zeros(1, ew.shape[1]), ew])
    top = need[L]
    n_top = {t: top[t].numel() for t in ("user", "item")}
    dims = model.dim_layer_list
    dtot = sum(dims)
    cat = {t: torch.empty(n_top[t], dtot, dtype=torch.float32, device=dev) for t in ("user", "item")}
    for t in cat:
        cat[t][:, :dims[0]] = embs[t].index_select(0, top[t])
    X = {t: embs[t].contiguous() for t in TYPES}
    rows_in = {t: None for t in TYPES}
    pos_in = {t: None for t in TYPES}
    saved = []
    off = dims[0]
    drops, seed = training_drop
    for li, ps in enumerate(layers_ps):
        layer = model.layer[str(li)]
        D, A = layer.in_features, layer.U.shape[1]
        att = {t: [x.contiguous() for x in ps[4 * i:4 * i + 4]] for i, t in enumerate(TYPES)}          # W_1, W_2, b, v
        dp = _dense_views(ps[12:])
        d_out = dims[li + 1]
        rows_out = need[li + 1]
        own = PJ.supported(D, A, 2 * A)               # the projections on csrc/proj.hip (else the library GEMMs)
        m = {t: (X[t].shape[0] if rows_out[t] is None else rows_out[t].numel()) for t in TYPES}
        lo, M = {}, 0
        for t in TYPES:
            lo[t], M = M, M + m[t]
        rng = {t: slice(lo[t], lo[t] + m[t]) for t in TYPES}
        T3 = [torch.empty(M, D, dtype=torch.float32, device=dev) for _ in range(3)]      # (user-, item-, tag-side) vectors
        st = {"rows_out": rows_out, "rows_in": rows_in, "pos_in": pos_in, "X": X, "att": att, "dp": dp, "ewp": ewp, "own": own, "m": m,
              "rng": rng, "M": M, "T3": T3}
        sel, Xs, Q, P = {}, {}, {}, {}
        for t, (n1, n2) in OTHERS.items():
            Xs[t] = T3[SLOT[t]][rng[t]]                     # the rows' own vectors: the type's slot of the triple
            if rows_out[t] is None:
                sel[t] = None
                Xs[t].copy_(X[t])
            else:
                sel[t] = rows_out[t] if rows_in[t] is None else PL.lookup(pos_in[t], rows_out[t])
                if m[t]:
                    torch.index_select(X[t], 0, sel[t], out=Xs[t])
            if own:
                Q[t] = PJ.tall_mm(X[t], att[t][1], torch.empty(X[t].shape[0], A, dtype=torch.float32, device=dev))
                P[(t, n1)] = torch.empty(m[t], A, dtype=torch.float32, device=dev)
                P[(t, n2)] = torch.empty(m[t], A, dtype=torch.float32, device=dev)
                PJ.tall_mm(Xs[t], att[n1][0][:D], P[(t, n1)], W2=att[n2][0][:D], w_split=2, b1=att[n1][2], b2=att[n2][2],
                           out2=P[(t, n2)])
            else:
                Q[t] = X[t] @ att[t][1]
                P[(t, n1)] = torch.addmm(att[n1][2], Xs[t], att[n1][0][:D])
                P[(t, n2)] = torch.addmm(att[n2][2], Xs[t], att[n2][0][:D])
        WT = {t: (PJ.small_mm(ewp, att[t][0][D:]) if own else ewp @ att[t][0][D:]) for t in TYPES}
        st.update(sel=sel, Xs=Xs, Q=Q, P=P, WT=WT)
        attns, idxs = {}, {}
        for r, (src, nb) in enumerate(RELATIONS):
            rows = rows_out[src]
            if m[src] == 0:
                continue
            if rows is None and rows_in[nb] is None:
                idx, widx = model.nbr[r]
            elif rows is None:                               # neighbour ids -> positions in the compact table
                idx, widx = model.nbr[r][0], model.nbr[r][1]
                idx = pos_in[nb].index_select(0, idx.flatten().long()).reshape(idx.shape)
            else:                                            # the rows' slices of both tables, renumbered, in one pass
                kk = model.nbr[r][0].shape[1]
                idx = torch.empty(m[src], kk, dtype=torch.int32, device=dev)
                widx = torch.empty(m[src], kk, dtype=torch.int32, device=dev)
                _lib.check(lib.tagrec_nbr_gather_i32(_lib.ptr(model.nbr[r][0]), _lib.ptr(model.nbr[r][1]), _lib.ptr(rows),
                                                     _lib.ptr(pos_in[nb] if rows_in[nb] is not None else None), m[src], kk,
                                                     _lib.ptr(idx), _lib.ptr(widx), _lib.stream_ptr()), "nbr_gather")
            _, attns[r] = attn_fwd(P[(src, nb)], Q[nb], WT[nb], att[nb][3].reshape(-1), X[nb], idx, widx,
                                   out=T3[SLOT[nb]][rng[src]])
            idxs[r] = (idx, widx)
        st.update(attns=attns, idxs=idxs)
        # message dropout of the layer's outputs (tgcn.py:217-219) in the fused kernel's epilogue: the library's counter-based
        # mask keyed by NODE id; the dropped output is all that is stored ([out' > 0] = mask [out > 0], see the backward)
        pk = drops[li] if drops else 0.0
        drop = None
        if pk > 0:
            drop = (pk, [_drop_seed(seed, li, t) for t in TYPES], [rows_out[t] for t in TYPES], lo["item"], lo["tag"])
        Od_all, BW_all = fuse_fwd(T3[0], T3[1], T3[2], dp, drop)
        st.update(O=Od_all, BW=BW_all)
        st["pk"] = pk
        Od = {t: Od_all[rng[t]] for t in TYPES}
        # positions of the batch rows in this layer's output, their normalised rows into the concat buffer
        extra = {}
        pos_out = {t: (need_pos[li + 1][t] if rows_out[t] is not None else None) for t in TYPES}
        norm_saved = {}
        for t in cat:
            extra[t] = top[t] if rows_out[t] is None else PL.lookup(pos_out[t], top[t])
            xr = Od[t].index_select(0, extra[t])
            inv = torch.empty(n_top[t], dtype=torch.float32, device=dev)
            if n_top[t]:
                _lib.check(lib.tagrec_rownorm_fwd_f32(_lib.ptr(xr), _lib.ptr(cat[t][:, off:]), dtot, _lib.ptr(inv), n_top[t], d_out,
                                                      _lib.stream_ptr()), "rownorm_fwd")
            norm_saved[t] = (xr, inv)
        st.update(extra=extra, norm=norm_saved, off=off, pos_out=pos_out)
        saved.append(st)
        _mark(f"forward layer {li} queued")
        off += d_out
        X, rows_in, pos_in = Od, rows_out, pos_out
    B = batch.shape[0]
    trip = model._batch_positions(batch)
    coef = torch.empty(B, dtype=torch.float32, device=dev)
    partials = torch.empty(2 * ((B + 3) // 4), dtype=torch.float32, device=dev)
    res = torch.empty(2, dtype=torch.float32, device=dev)
    U, I = cat["user"], cat["item"]
    _lib.check(lib.tagrec_bpr_fwd_f32(_lib.ptr(U), _lib.ptr(I), dtot, dtot, _lib.ptr(U), _lib.ptr(I), dtot, dtot, _lib.ptr(trip), B,
                                      H.loss_kind_id(model.loss_func), _lib.ptr(coef), _lib.ptr(partials), _lib.ptr(res),
                                      _lib.stream_ptr()), "bpr_fwd")
    return res, {"saved": saved, "cat": cat, "trip": trip, "coef": coef, "top": top, "dims": dims, "sizes": sizes,
                 "seed": seed}


def step_backward(model, g, state, n_weight):
    """Gradients of (embed.user, embed.item, embed.tag, embed.weight, 21 parameters per layer)."""
    from . import tgcn as TG
    lib = _lib.load()
    saved, cat, trip, coef, top, dims = (state[k] for k in ("saved", "cat", "trip", "coef", "top", "dims"))
    dev = trip.device
    dtot = sum(dims)
    B = trip.shape[0]
    d_cat = {t: torch.zeros_like(cat[t]) for t in cat}
    U, I, dU, dI = cat["user"], cat["item"], d_cat["user"], d_cat["item"]
    _lib.check(lib.tagrec_bpr_bwd_f32(_lib.ptr(U), _lib.ptr(I), dtot, dtot, _lib.ptr(U), _lib.ptr(I), dtot, dtot, _lib.ptr(trip), B,
                                      _lib.ptr(coef), _lib.ptr(g.contiguous()), 1.0, _lib.ptr(dU), _lib.ptr(dI), _lib.ptr(dU),
                                      _lib.ptr(dI), _lib.stream_ptr()), "bpr_bwd")
    L = len(saved)
    # Adam.fuse_into(model): the node tables' update rides in the epilogue of the product that forms the last term of their
    # gradient (bottom layer, dQ W_2^T): no gradient tensor for them, no separate optimizer pass over the tables
    from .train import fused_optimizer
    fused_opt = fused_optimizer(model)
    fused_done = set()
    layer_grads = [None] * L
    d_ewp = None
    G_all = None              # gradient w.r.t. the merged output rows of the layer being processed (written by the layer above)
    d_tables = None
    for li in range(L - 1, -1, -1):
        st = saved[li]
        layer = model.layer[str(li)]
        D, A = layer.in_features, layer.U.shape[1]
        att, dp, ewp = st["att"], st["dp"], st["ewp"]
        rows_out, X, Xs, sel, m, rng, M, T3 = (st[k_] for k_ in ("rows_out", "X", "Xs", "sel", "m", "rng", "M", "T3"))
        pk = st["pk"]
        # gradient w.r.t. the (dropped) output rows: what the layer above sent down + the loss's normalised batch rows
        if G_all is None:
            G_all = torch.zeros(M, dims[li + 1], dtype=torch.float32, device=dev)
        for t in cat:
            if m[t] and cat[t].shape[0]:
                xr, inv = st["norm"][t]
                dz = torch.empty_like(xr)
                _lib.check(lib.tagrec_rownorm_bwd_f32(_lib.ptr(xr), _lib.ptr(inv), _lib.ptr(d_cat[t][:, st["off"]:]), dtot, 1.0,
                                                      _lib.ptr(dz), 0, xr.shape[0], xr.shape[1], _lib.stream_ptr()), "rownorm_bwd")
                G_all[rng[t]].index_add_(0, st["extra"][t], dz)          # (the batch rows of a type are distinct)
        if pk > 0:                                # d out = d out' mask / (1 - p), and the kernels' relu mask [out' > 0] holds the mask
            G_all.mul_(1.0 / (1.0 - pk))
        # ---- fused dense block, the three types in one launch: input gradients + its weight gradients
        r_ = TG._FusedDense._backward_rows(T3[0], T3[1], T3[2], *dp[:8], st["O"], st["BW"], G_all)
        dts_all, dense_g = r_[:3], list(r_[3:12])
        G_all = None
        # where this layer's input gradients go: row slices of ONE merged buffer (= the output gradient of the layer below),
        # or, at the bottom, one buffer per embedding table
        n_in = {t: X[t].shape[0] for t in TYPES}
        if li > 0:
            below = saved[li - 1]
            dX_all = torch.empty(below["M"], D, dtype=torch.float32, device=dev)
            dx = {t: dX_all[below["rng"][t]] for t in TYPES}
        else:
            dX_all = None
            dx = {t: torch.empty(n_in[t], D, dtype=torch.float32, device=dev) for t in TYPES}
        have_x = {t: False for t in TYPES}        # dx[t] holds something yet?
        accQ = {t: None for t in TYPES}
        dP = {}
        zs = [att[t][i] for i in (0, 2, 3) for t in TYPES]                 # dW1, db, dv of the three types: one zero fill
        zflat = torch.zeros(sum(z.numel() for z in zs), dtype=torch.float32, device=dev)
        zv, zo = [], 0
        for z in zs:
            zv.append(zflat[zo:zo + z.numel()].view(z.shape))
            zo += z.numel()
        dW1, db, dv = ({t: zv[3 * i + j] for j, t in enumerate(TYPES)} for i in range(3))
        dWTs = {t: None for t in TYPES}
        # ---- neighbour attentions: scatter-form relations first (into zeroed buffers), then the pull-form ones, each adding
        # what has been collected so far in its product's epilogue
        order = sorted(range(6), key=lambda r: (m[RELATIONS[r][0]] >= TG._PULL_MIN_ROWS, r))
        for r in order:
            src, nb = RELATIONS[r]
            if m[src] == 0:
                continue
            d_out = dts_all[SLOT[nb]][rng[src]]
            idx, widx = st["idxs"][r]
            pull = m[src] >= TG._PULL_MIN_ROWS
            k = idx.shape[1]
            if pull and A in (16, 32):
                # destination-centric pull of dEj (+ da on the way), source-centric softmax backward from da (8 bytes per pair
                # out), destination-centric pull of dQ from the compressed pairs
                static = None
                inv_r = model.inv[r] if model.inv is not None else None
                if (inv_r is not None and rows_out[src] is not None
                        and m[src] * k >= STATIC_INVERSION_MIN_SHARE * inv_r.perm32.numel()):
                    static = (inv_r.perm32, inv_r.dest32, st["pos_out"][src], st["pos_in"][nb])
                dP[r], dWT_r, dv_r, accQ[nb], _ = attn_bwd_pulls(
                    st["P"][(src, nb)], st["Q"][nb], st["WT"][nb], att[nb][3].reshape(-1), X[nb], idx, widx, st["attns"][r], d_out,
                    accQ[nb], dx[nb] if have_x[nb] else None, model.w_major[r], out=dx[nb], static=static)
                have_x[nb] = True
            else:
                if pull:
                    dh = torch.empty(m[src] * k, A, dtype=torch.float32, device=dev)
                    dQ = dEj = None
                else:
                    dh = None
                    if not have_x[nb]:
                        dx[nb].zero_()
                        have_x[nb] = True
                    if accQ[nb] is None:
                        accQ[nb] = torch.zeros(n_in[nb], A, dtype=torch.float32, device=dev)
                    dQ, dEj = accQ[nb], dx[nb]
                dP[r], dWT_r, dv_r = attn_bwd(st["P"][(src, nb)], st["Q"][nb], st["WT"][nb], att[nb][3].reshape(-1), X[nb], idx, widx,
                                              st["attns"][r], d_out, dQ, dEj, dh)
                if pull:
                    accQ[nb], px = _pull_add(idx, st["attns"][r], d_out, dh, n_in[nb], accQ[nb], dx[nb] if have_x[nb] else None)
                    dx[nb].copy_(px)
                    have_x[nb] = True
            dWTs[nb] = dWT_r if dWTs[nb] is None else dWTs[nb] + dWT_r
            dv[nb] += dv_r.reshape(dv[nb].shape)
        # ---- projections backward
        own = st["own"]
        dW2s = {}
        if li == 0:                               # the ego slot of the concat at the batch rows (layer 0's "output" is the table)
            for t in cat:
                if cat[t].shape[0]:
                    if not have_x[t]:
                        dx[t].zero_()
                        have_x[t] = True
                    dx[t].index_add_(0, top[t], d_cat[t][:, :dims[0]])
        for t, (n1, n2) in OTHERS.items():
            r1, r2 = RELATIONS.index((t, n1)), RELATIONS.index((t, n2))
            dXs = None
            if m[t]:
                dXs = dts_all[SLOT[t]][rng[t]]
                if own:
                    PJ.tall_mm(dP[r1], att[n1][0][:D], dXs, X2=dP[r2], W2=att[n2][0][:D], w_split=1, transposed=True, accumulate=True)
                    dW = PJ.tall_wgrad(Xs[t], dP[r1], dP[r2], db1=db[n1], db2=db[n2], acc_b=True)
                    dW1[n1][:D] += dW[:, :A]
                    dW1[n2][:D] += dW[:, A:]
                else:
                    for r_i, n_ in ((r1, n1), (r2, n2)):
                        dXs.addmm_(dP[r_i], att[n_][0][:D].t())
                        dW1[n_][:D] += tall_wgrad(Xs[t], dP[r_i])
                        db[n_] += dP[r_i].sum(0, keepdim=True)
            fuse_t = (li == 0 and fused_opt is not None and own and accQ[t] is not None
                      and X[t].data_ptr() == model.embed[t].data_ptr() and X[t].shape[1] in PJ.DIMS)
            if fuse_t:
                # every other term first (the product below is the LAST one), the weight gradient before the table changes
                dW2s[t] = PJ.tall_wgrad(X[t], accQ[t])
                if dXs is not None:
                    if not have_x[t]:
                        dx[t].zero_()
                        have_x[t] = True
                    if sel[t] is None:
                        dx[t] += dXs
                    else:
                        PJ.row_add_at(dx[t], sel[t], dXs)
                    dXs = None
                prm = model.embed[t]
                m_, v_, t_ = fused_opt.fused_state(prm)
                PJ.tall_mm_adam(accQ[t], att[t][1], dx[t] if have_x[t] else None, prm.data, m_, v_, fused_opt.lr, fused_opt.betas,
                                fused_opt.eps, t_, transposed=True)
                fused_opt.fused_commit(prm)
                fused_done.add(t)
                have_x[t] = True
            elif accQ[t] is not None:
                if own:
                    PJ.tall_mm(accQ[t], att[t][1], dx[t], transposed=True, accumulate=have_x[t])
                    dW2s[t] = PJ.tall_wgrad(X[t], accQ[t])
                else:
                    if have_x[t]:
                        dx[t].addmm_(accQ[t], att[t][1].t())
                    else:
                        torch.mm(accQ[t], att[t][1].t(), out=dx[t])
                    dW2s[t] = tall_wgrad(X[t], accQ[t])
                have_x[t] = True
            else:
                dW2s[t] = torch.zeros_like(att[t][1])
            if not have_x[t]:
                dx[t].zero_()
            if dXs is not None:
                if sel[t] is None:
                    dx[t] += dXs
                elif own:
                    PJ.row_add_at(dx[t], sel[t], dXs)
                else:
                    dx[t].index_add_(0, sel[t], dXs)
            if dWTs[t] is not None:
                if own:
                    PJ.small_mm(ewp.t(), dWTs[t], out=dW1[t][D:], accumulate=True)
                    if d_ewp is None:
                        d_ewp = PJ.small_mm(dWTs[t], att[t][0][D:].t())
                    else:
                        PJ.small_mm(dWTs[t], att[t][0][D:].t(), out=d_ewp, accumulate=True)
                else:
                    dW1[t][D:] += ewp.t() @ dWTs[t]
                    contrib = dWTs[t] @ att[t][0][D:].t()
                    d_ewp = contrib if d_ewp is None else d_ewp + contrib
        lg = []
        for t in TYPES:
            lg += [dW1[t], dW2s[t], db[t], dv[t]]
        shapes = [x.shape for x in layer_params(layer)[12:]]
        lg += [x.reshape(s_) for x, s_ in zip(dense_g, shapes)]
        layer_grads[li] = lg
        saved[li] = None
        _mark(f"backward layer {li} queued")
        if li > 0:
            G_all = dX_all
        else:
            d_tables = dx
    for t in fused_done:                          # updated above: the optimizer gets no gradient for them
        d_tables[t] = None
    d_weight = d_ewp[1:] if d_ewp is not None else torch.zeros(n_weight, model.dim_weight, device=dev)
    flat = [d_tables["user"], d_tables["item"], d_tables["tag"], d_weight]
    for lg in layer_grads:
        flat += lg
    return flat


def _pull_add(idxc, attnc, doc, dh, n_dst, addQ, addX):
    """(addQ + S dh, addX + G dOut) over the relation's table inverted on the spot (one radix sort of the n k neighbour
    ids; pad slots sort behind the last destination row and are never read); add* may be None."""
    from .graph import Graph
    nc, k = idxc.shape
    flat = idxc.flatten()
    key = torch.where(flat > 0, flat - 1, n_dst)
    skey, order = torch.sort(key, stable=True)
    rowptr = torch.searchsorted(skey, torch.arange(n_dst + 1, dtype=skey.dtype, device=skey.device))
    Gm = Graph(rowptr, torch.div(order, k, rounding_mode="floor").to(torch.int32), attnc.flatten().index_select(0, order),
               (n_dst, nc), workspace=True)
    Sm = Gm.like(order.to(torch.int32), torch.ones_like(Gm.val), nc * k)
    if addQ is None:
        outQ = Sm.spmm(dh)
    else:
        outQ = torch.empty_like(addQ)
        Sm.spmm_axpy(dh, addQ, 1.0, outQ)
    if addX is None:
        outX = Gm.spmm(doc)
    else:
        outX = torch.empty_like(addX)
        Gm.spmm_axpy(doc, addX, 1.0, outX)
    return outQ, outX


class TgcnBprLoss(torch.autograd.Function):
    """(embedding tables, layer parameters) -> [mul_loss, l2reg_loss (unweighted)] for one BPR batch."""

    @staticmethod
    def forward(ctx, model, batch, drops, seed, eu, ei, et, ew, *flat):
        n_l = len(model.layer)
        layers_ps = [[p.detach() for p in flat[N_LAYER_PARAMS * i:N_LAYER_PARAMS * (i + 1)]] for i in range(n_l)]
        embs = {"user": eu.detach(), "item": ei.detach(), "tag": et.detach()}
        res, state = step_forward(model, batch, embs, ew.detach(), layers_ps, (drops, seed))
        ctx.model, ctx.state, ctx.n_weight = model, state, ew.shape[0]
        return res

    @staticmethod
    def backward(ctx, g):
        grads = step_backward(ctx.model, g, ctx.state, ctx.n_weight)
        ctx.state = None
        return (None, None, None, None, *grads)
