"""LightGCN behind the reference's model surface (/root/reference/model/lightgcn.py).

    LightGCN(data)            ctor reads data.num / data.ui_adj[/ut_adj/it_adj] and the config   (:11-35)
    .forward()                -> tuple(user_emb, item_emb[, tag_emb])                              (:49-63)
    .loss(batch[B,3])         -> (mul_loss, reg * l2reg_loss on EGO rows), both differentiable     (:68-82)
    .predict_rating(users)    -> sigmoid(U_b I^T), [b, n_item]                                     (:84-89)
    .state_dict()             keys embed.0 / embed.1 [/ embed.2], as the reference's ParameterList

Differences in mechanism, not in results:
  * the per-type tables are row slices of ONE contiguous N x D parameter (`table`), so the
    `torch.cat` / `torch.split` copies of :52,62 disappear and Adam is one launch;
  * each layer is one fused HIP kernel (SpMM + L2-normalise + running layer mean), and the whole
    backward is hand-derived: BPR scatter, then one fused SpMM + normalise-backward per layer.
"""
import torch

from . import _lib, help as H
from .base import StepWorkspace, TableModel, _Token, step_buffer, xavier_tables  # noqa: F401  (xavier_tables re-exported)
from .config import CFG as _GLOBAL_CFG
from .graph import Graph, creat_adj
from .train import fused_optimizer


def _layer_seed(seed, k):
    return (int(seed) * 64 + k) & 0xFFFFFFFFFFFFFFFF


def propagate_forward(graph, x0, n_layer, drops=None, seed=0, loss_rows=None, masks_out=None):
    """x0 -> (out, raws, invs): out = mean(x0, z1..zL), raws[k] = dropout(A raws[k-1]) (un-normalised),
    invs[k][r] = 1/max(||raws[k][r]||, 1e-12).  One fused kernel per layer.  drops[k] > 0 = message dropout of
    layer k's product (lightgcn.py:56), drawn inside the kernel from (seed, layer, element).

    loss_rows (int64 node ids): `out` will be read at these rows only (the batch rows of the BPR loss).  Then the last
    layer is computed on them alone and the layer below it on their neighbours (anything further down reaches nearly
    every node through the popular items, so it runs in full); the rows left out stay zero in raws / invs and are
    never read with a non-zero gradient in the backward pass.  Needs a square graph and a vector-kernel width.
    masks_out (dict): receives {layer: row mask} of the restricted layers, for `propagate_backward`."""
    s = 1.0 / (n_layer + 1)
    out = x0 * s
    raws, invs = [], []
    x = x0
    masks = {}
    if (loss_rows is not None and n_layer >= 1 and graph.shape[0] == graph.shape[1] and x0.shape[1] in (8, 16, 32, 64, 128, 256)
            and loss_rows.numel() * 16 <= x0.shape[0]):              # a batch that touches most rows gains nothing
        top = torch.zeros(x0.shape[0], dtype=torch.uint8, device=x0.device)
        top.index_fill_(0, loss_rows, 1)
        masks[n_layer - 1] = top
        if n_layer >= 2:
            masks[n_layer - 2] = graph.mark_rows(loss_rows, torch.zeros_like(top))
    if masks_out is not None:
        masks_out.update(masks)
    for k in range(n_layer):
        if k in masks:
            y = torch.zeros_like(x0)
            inv = torch.zeros(x0.shape[0], dtype=torch.float32, device=x0.device)
            graph.spmm_norm_acc_rows(x, y, inv, out, s, masks[k], drops[k] if drops else 0.0, _layer_seed(seed, k))
            raws.append(y)
            invs.append(inv)
            x = y
            continue
        y = torch.empty_like(x0)
        inv = torch.empty(x0.shape[0], dtype=torch.float32, device=x0.device)
        graph.spmm_norm_acc(x, y, inv, out, s, drops[k] if drops else 0.0, _layer_seed(seed, k))
        raws.append(y)
        invs.append(inv)
        x = y
    return out, raws, invs


def propagate_backward(graph_t, d_out, raws, invs, drops=None, seed=0, masks=None, fused=None):
    """Gradient of `propagate_forward` w.r.t. x0 given d_out (dense [N,D]).
    G^L = nb(X^L);  G^k = A^T G^(k+1) + nb(X^k);  G^0 = A^T G^1 + s*d_out,  nb = normalise-backward of s*d_out.
    With dropout each G^k (k >= 1) is multiplied by layer k's mask / (1 - p) before it travels on (it is the gradient
    w.r.t. the product the mask was applied to).
    masks = the row masks of a restricted forward (`propagate_forward(masks_out=...)`): a layer that was computed on
    masks[k] only has raws[k] = invs[k] = 0 elsewhere, and the gradient arriving from the layer above lives on rows
    whose neighbours all lie inside masks[k] -- so G^k is exactly zero outside masks[k] and those rows are not visited."""
    L = len(raws)
    s = 1.0 / (L + 1)
    if L == 0:
        return d_out.clone()
    n, D = d_out.shape
    lib = _lib.load()
    g = torch.empty_like(d_out)
    # The gradient is non-zero on the batch rows only at the head of the chain and spreads by one hop per layer:
    # every product is told which rows of its operand hold a non-zero and leaves the others unfetched (bit-identical
    # result; the kernel ignores the flags once they cover 4/5 of the rows, without a host round trip).
    sparse = D in (8, 16, 32, 64, 128, 256)
    if sparse:
        flags = [torch.empty(n, dtype=torch.uint8, device=d_out.device) for _ in range(2)]
        counts = torch.zeros(2, dtype=torch.int32, device=d_out.device)
        _lib.check(lib.tagrec_rownorm_bwd_flags_f32(_lib.ptr(raws[L - 1]), _lib.ptr(invs[L - 1]), _lib.ptr(d_out), D, s,
                                                    _lib.ptr(g), 0, n, D, _lib.ptr(flags[0]), _lib.ptr(counts[0:1]),
                                                    _lib.stream_ptr()), "rownorm_bwd_flags")
    else:
        _lib.check(lib.tagrec_rownorm_bwd_f32(_lib.ptr(raws[L - 1]), _lib.ptr(invs[L - 1]), _lib.ptr(d_out), D, s,
                                              _lib.ptr(g), 0, n, D, _lib.stream_ptr()), "rownorm_bwd")
    if drops and drops[L - 1] > 0:
        H.message_drop(g, drops[L - 1], _layer_seed(seed, L - 1), out=g)     # flags stay a superset of the non-zero rows
    cur = 0
    for k in range(L - 2, -1, -1):
        mask = masks.get(k) if (masks and sparse and (k + 1) in masks) else None
        gn = torch.empty_like(d_out) if mask is None else torch.zeros_like(d_out)
        if sparse:
            if mask is not None:
                flags[1 - cur].zero_()
            graph_t.spmm_normbwd_sparse(g, flags[cur], counts[cur:cur + 1], raws[k], invs[k], d_out, s, gn, flags[1 - cur],
                                        counts[1 - cur:2 - cur], drops[k] if drops else 0.0, _layer_seed(seed, k), mask)
            cur = 1 - cur
        else:
            graph_t.spmm_normbwd(g, raws[k], invs[k], d_out, s, gn, drops[k] if drops else 0.0, _layer_seed(seed, k))
        g = gn
    if fused is not None and sparse:      # (table, optimizer): Adam in the epilogue of the last hop, no gradient tensor
        table, opt = fused
        m, v, step = opt.fused_state(table)
        graph_t.spmm_axpy_adam(g, flags[cur], counts[cur:cur + 1], d_out, s, None, table.data, m, v, opt.lr, opt.betas, opt.eps, step,
                               opt.fused_dev(table))
        opt.fused_commit(table)
        return None
    g0 = torch.empty_like(d_out)
    if sparse:
        graph_t.spmm_axpy_sparse(g, flags[cur], counts[cur:cur + 1], d_out, s, g0)
    else:
        graph_t.spmm_axpy(g, d_out, s, g0)
    return g0


VEC_WIDTHS = (8, 16, 32, 64, 128, 256)


def spmm_listed(graph, rows, x, out=None):
    """(A x)[rows] as a compact [len(rows), D] tensor (rows int64, may repeat)."""
    rows = rows.contiguous()
    lib = _lib.load()
    if out is None:
        out = torch.empty(rows.numel(), x.shape[1], dtype=torch.float32, device=x.device)
    ws_n = lib.tagrec_spmm_listed_workspace(rows.numel(), x.shape[1])
    ws = torch.empty(max(ws_n, 1), dtype=torch.float32, device=x.device)
    graph._call("spmm_listed", lib.tagrec_spmm_listed_f32, graph.handle, _lib.ptr(rows), rows.numel(), _lib.ptr(x),
                _lib.ptr(out), x.shape[1], _lib.ptr(ws), ws_n, _lib.stream_ptr())
    return out


def restricted_forward(graph, x0, n_layer, rows, ws=None):
    """The forward pass of a training step whose loss reads the layer mean at `rows` (int64 node ids [T], may repeat)
    only -- the BPR batch rows (lightgcn.py:71-75).  Layers below L-1 run on all rows (through the popular items every
    row is within two hops of the batch), layer L-1 on the batch rows and their neighbours (row-masked kernel), layer L
    in compact form on the batch rows alone (`spmm_listed`); no layer accumulates the mean, which is formed at the end
    on the T rows.  Returns (out_b [T, D], state for `restricted_backward`).  Rows a layer did not compute are left
    UNWRITTEN in its output; every later reader is told which rows are valid."""
    L, s = n_layer, 1.0 / (n_layer + 1)
    n, D = x0.shape
    T = rows.numel()
    dev = x0.device
    mid = graph.mark_rows(rows, step_buffer(ws, "mid", (n,), torch.uint8, dev).zero_()) if L >= 2 else None
    raws, invs = [], []
    x = x0
    for k in range(L - 1):
        y = step_buffer(ws, f"y{k}", (n, D), torch.float32, dev)
        inv = step_buffer(ws, f"inv{k}", (n,), torch.float32, dev)
        graph.spmm_norm_acc_rows(x, y, inv, None, 0.0, mid if k == L - 2 else None)
        raws.append(y)
        invs.append(inv)
        x = y
    y_top = spmm_listed(graph, rows, x)
    z_top = torch.empty_like(y_top)
    inv_top = torch.empty(T, dtype=torch.float32, device=x0.device)
    _lib.check(_lib.load().tagrec_rownorm_fwd_f32(_lib.ptr(y_top), _lib.ptr(z_top), D, _lib.ptr(inv_top), T, D, _lib.stream_ptr()),
               "rownorm_fwd")
    out_b = x0.index_select(0, rows) * s
    for y, inv in zip(raws, invs):
        out_b.addcmul_(y.index_select(0, rows), inv.index_select(0, rows)[:, None], value=s)
    out_b.add_(z_top, alpha=s)
    return out_b, (raws, invs, mid, y_top, inv_top)


def restricted_backward(graph_t, rows, d_out_b, state, shape, fused=None, ws=None):
    """Gradient w.r.t. x0 of `restricted_forward` given d_out_b [T, D] = d loss / d out_b.  The chain starts on the batch
    rows (compact), lands on their neighbours (row-masked hop: G is non-zero there only) and spreads from there; every
    operand travels with one flag byte per row and zero rows are not gathered; the normalize-backward / mean terms exist
    on the batch rows only (dz_flags), so no other row's epilogue reads X_raw or dZ."""
    raws, invs, mid, y_top, inv_top = state
    n, D = shape
    L = len(raws) + 1
    s = 1.0 / (L + 1)
    T = rows.numel()
    dev = d_out_b.device
    lib = _lib.load()
    tflag = step_buffer(ws, "tflag", (n,), torch.uint8, dev).zero_()
    tflag.index_fill_(0, rows, 1)
    dz = step_buffer(ws, "dz", (n, D), torch.float32, dev)              # d loss / d out, valid on the batch rows only
    dz.index_fill_(0, rows, 0.0)
    dz.index_add_(0, rows, d_out_b)
    g_top = torch.empty(T, D, dtype=torch.float32, device=dev)
    _lib.check(lib.tagrec_rownorm_bwd_f32(_lib.ptr(y_top), _lib.ptr(inv_top), _lib.ptr(d_out_b), D, s, _lib.ptr(g_top), 0, T, D,
                                          _lib.stream_ptr()), "rownorm_bwd")
    g = step_buffer(ws, "g_top", (n, D), torch.float32, dev)             # G^L: valid on the batch rows only (flags = tflag)
    g.index_fill_(0, rows, 0.0)
    g.index_add_(0, rows, g_top)
    flags, count = tflag, None                                           # count None: the flags are always consulted
    for k in range(L - 2, -1, -1):
        masked = k == L - 2
        gn = step_buffer(ws, f"g{k & 1}", (n, D), torch.float32, dev)
        fo = step_buffer(ws, f"fo{k & 1}", (n,), torch.uint8, dev)
        if masked:
            fo.zero_()
        cnt = step_buffer(ws, f"cnt{k & 1}", (1,), torch.int32, dev).zero_()
        graph_t.spmm_normbwd_sparse(g, flags, count, raws[k], invs[k], dz, s, gn, fo, cnt, row_mask=mid if masked else None,
                                    dz_flags=tflag)
        raws[k] = invs[k] = None          # last use: the 4 N D bytes go back to the allocator before the next hop allocates
        # a masked hop wrote the rows of `mid` only: its flags must always be honoured; a full hop wrote every row
        g, flags, count = gn, fo, (None if masked else cnt)
    if fused is not None:             # (table, optimizer): the last hop applies Adam to the table, no gradient is written
        table, opt = fused
        m, v, step = opt.fused_state(table)
        graph_t.spmm_axpy_adam(g, flags, count, dz, s, tflag, table.data, m, v, opt.lr, opt.betas, opt.eps, step, opt.fused_dev(table))
        opt.fused_commit(table)
        return None
    g0 = torch.empty(n, D, dtype=torch.float32, device=dev)              # (handed to the optimizer: not a workspace buffer)
    graph_t.spmm_axpy_sparse(g, flags, count, dz, s, g0, b_flags=tflag)
    return g0


class _Propagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, graph, n_layer, drops=None, seed=0):
        out, raws, invs = propagate_forward(graph, table.detach(), n_layer, drops, seed)
        ctx.graph, ctx.raws, ctx.invs, ctx.drops, ctx.seed = graph, raws, invs, drops, seed
        return out

    @staticmethod
    def backward(ctx, d_out):
        g0 = propagate_backward(ctx.graph.transpose(), d_out.contiguous(), ctx.raws, ctx.invs, ctx.drops, ctx.seed)
        ctx.raws = ctx.invs = None
        return g0, None, None, None, None


class _PropagateBprLoss(torch.autograd.Function):
    """table -> [mul_loss, l2reg_loss(ego rows)] in one autograd node."""

    @staticmethod
    def forward(ctx, table, graph, n_layer, n_user, n_item, trip, loss_kind, reg_active, drops=None, seed=0, restrict=True,
                fused_opt=None, ws=None):
        x0 = table.detach()
        ctx.fused = (table, fused_opt) if fused_opt is not None else None
        ctx.ws, ctx.token = None, _Token()
        if ws is not None and ws.acquire(ctx.token):
            ctx.ws = ws
        B, D = trip.shape[0], x0.shape[1]
        n = x0.shape[0]
        lib = _lib.load()
        coef = torch.empty(B, dtype=torch.float32, device=x0.device)
        partials = torch.empty(2 * ((B + 3) // 4), dtype=torch.float32, device=x0.device)
        res = torch.empty(2, dtype=torch.float32, device=x0.device)
        ctx.n_user, ctx.n_item, ctx.reg_active, ctx.trip, ctx.coef, ctx.graph = n_user, n_item, reg_active, trip, coef, graph
        # the loss reads `out` at the batch rows only: users, and items offset by n_user
        rows = torch.cat([trip[:, 0], trip[:, 1] + n_user, trip[:, 2] + n_user]) if restrict else None
        ctx.compact = bool(restrict and drops is None and n_layer >= 1 and graph.shape[0] == graph.shape[1] and D in VEC_WIDTHS
                           and 3 * B * 16 <= n)                       # a batch that touches most rows gains nothing
        if ctx.compact:
            out_b, ctx.state = restricted_forward(graph, x0, n_layer, rows, ctx.ws)
            ego_b = x0.index_select(0, rows)
            ar = torch.arange(B, device=x0.device)
            ctrip = torch.stack([ar, ar, ar + B], dim=1).contiguous()
            _lib.check(lib.tagrec_bpr_fwd_f32(_lib.ptr(out_b[:B]), _lib.ptr(out_b[B:]), D, D, _lib.ptr(ego_b[:B]), _lib.ptr(ego_b[B:]),
                                              D, D, _lib.ptr(ctrip), B, loss_kind, _lib.ptr(coef), _lib.ptr(partials),
                                              _lib.ptr(res), _lib.stream_ptr()), "bpr_fwd")
            ctx.rows, ctx.out_b, ctx.ego_b, ctx.ctrip, ctx.shape = rows, out_b, ego_b, ctrip, x0.shape
            return res
        ctx.masks = {}
        out, raws, invs = propagate_forward(graph, x0, n_layer, drops, seed, rows, ctx.masks)
        ctx.drops, ctx.seed = drops, seed
        U, I = out[:n_user], out[n_user:n_user + n_item]
        Ue, Ie = x0[:n_user], x0[n_user:n_user + n_item]
        _lib.check(lib.tagrec_bpr_fwd_f32(_lib.ptr(U), _lib.ptr(I), D, D, _lib.ptr(Ue), _lib.ptr(Ie), D, D,
                                          _lib.ptr(trip), B, loss_kind, _lib.ptr(coef), _lib.ptr(partials),
                                          _lib.ptr(res), _lib.stream_ptr()), "bpr_fwd")
        ctx.raws, ctx.invs = raws, invs
        ctx.out, ctx.x0 = out, x0
        return res

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        lib = _lib.load()
        null = _lib.c_void_p(0)
        if ctx.compact:
            out_b, ego_b, ctrip, rows = ctx.out_b, ctx.ego_b, ctx.ctrip, ctx.rows
            B, D = ctrip.shape[0], out_b.shape[1]
            d_b = torch.zeros(2, 3 * B, D, dtype=torch.float32, device=out_b.device)   # d / d out_b, d / d ego_b
            reg = ctx.reg_active
            _lib.check(lib.tagrec_bpr_bwd_f32(_lib.ptr(out_b[:B]), _lib.ptr(out_b[B:]), D, D,
                                              _lib.ptr(ego_b[:B]) if reg else null, _lib.ptr(ego_b[B:]) if reg else null,
                                              D if reg else 0, D if reg else 0, _lib.ptr(ctrip), B, _lib.ptr(ctx.coef), _lib.ptr(g), 1.0,
                                              _lib.ptr(d_b[0][:B]), _lib.ptr(d_b[0][B:]),
                                              _lib.ptr(d_b[1][:B]) if reg else null, _lib.ptr(d_b[1][B:]) if reg else null,
                                              _lib.stream_ptr()), "bpr_bwd")
            fused = ctx.fused if (ctx.fused is not None and not ctx.reg_active) else None
            g0 = restricted_backward(ctx.graph.transpose(), rows, d_b[0], ctx.state, ctx.shape, fused, ctx.ws)
            if ctx.reg_active:
                g0.index_add_(0, rows, d_b[1])                            # L2 term on the ego rows
            ctx.state = ctx.out_b = None
            if ctx.ws is not None:
                ctx.ws.release(ctx.token)
            return g0, None, None, None, None, None, None, None, None, None, None, None, None
        out, x0, trip = ctx.out, ctx.x0, ctx.trip
        nu, ni, D, B = ctx.n_user, ctx.n_item, x0.shape[1], trip.shape[0]
        d_out = torch.zeros_like(out)
        U, I = out[:nu], out[nu:nu + ni]
        _lib.check(lib.tagrec_bpr_bwd_f32(_lib.ptr(U), _lib.ptr(I), D, D, null, null, 0, 0, _lib.ptr(trip), B,
                                          _lib.ptr(ctx.coef), _lib.ptr(g), 1.0,
                                          _lib.ptr(d_out[:nu]), _lib.ptr(d_out[nu:nu + ni]), null, null,
                                          _lib.stream_ptr()), "bpr_bwd")
        fused = ctx.fused if (ctx.fused is not None and not ctx.reg_active and len(ctx.raws) >= 1) else None
        g0 = propagate_backward(ctx.graph.transpose(), d_out, ctx.raws, ctx.invs, ctx.drops, ctx.seed, ctx.masks, fused)
        # L2 term on the ego rows: added after the propagation hop has written g0
        if ctx.reg_active:
            Ue, Ie = x0[:nu], x0[nu:nu + ni]
            _lib.check(lib.tagrec_bpr_bwd_f32(_lib.ptr(U), _lib.ptr(I), D, D, _lib.ptr(Ue), _lib.ptr(Ie), D, D,
                                              _lib.ptr(trip), B, _lib.ptr(ctx.coef), _lib.ptr(g), 1.0,
                                              null, null, _lib.ptr(g0[:nu]), _lib.ptr(g0[nu:nu + ni]),
                                              _lib.stream_ptr()), "bpr_bwd(reg)")
        ctx.raws = ctx.invs = ctx.out = None
        if ctx.ws is not None:
            ctx.ws.release(ctx.token)
        return g0, None, None, None, None, None, None, None, None, None, None, None, None


class LightGCN(TableModel):
    def __init__(self, data, args=None, config=None, graph=None):
        super().__init__()
        self._config(config if config is not None else _GLOBAL_CFG)
        self._init_table(data, self.use_tag, self.dim_latent, self.device)
        self.norm_adj = graph if graph is not None else creat_adj(data, self.use_tag, self.norm_type,
                                                                  self.split_adj_k, self.device)

    def _config(self, config):
        self.dim_latent = config["dim_latent"]
        self.num_layer = len(config["dim_layer_list"])   # only the LENGTH matters (lightgcn.py:27)
        self.device = torch.device(config["device"])
        self.norm_type = config["norm_type"]
        self.split_adj_k = config["split_adj_k"]
        self.reg = config["reg"]
        self.loss_func = config["mul_loss_func"]
        self.use_tag = config["use_tag"]
        self.message_drop_list = config["message_drop_list"]
        self.node_drop = config["node_drop"]
        self.drop_seed = config.get("seed", 2020)
        # loss(): compute the top two layers only on the rows the batch's loss depends on (propagate_forward)
        self.restrict_forward = bool(config.get("restrict_forward", True))
        # persistent buffers of the restricted training step (base.StepWorkspace); config["step_workspace"] = False: allocate per step
        self.step_ws = StepWorkspace() if config.get("step_workspace", True) else None

    def _fused_ok(self):
        return isinstance(self.norm_adj, Graph)

    fused_capturable = True        # Adam(capturable=True).fuse_into(model): the fused update advances its counter on the device

    def set_fused_optimizer(self, opt):
        """`Adam.fuse_into(model)`: the compact restricted step (reg == 0) applies the table's Adam update in the epilogue of
        its last backward product; every other path hands the optimizer a gradient as usual.  None switches it off."""
        self._fused_opt = opt

    def _drops(self):
        """(per-layer drop rates, seed of this forward pass) when message dropout is active, else (None, 0).  The
        seed advances with every training-mode forward pass; masks are functions of (seed, layer, element)."""
        drops = [float(p) for p in self.message_drop_list[:self.num_layer]]
        if not (self.training and any(p > 0 for p in drops)):
            return None, 0
        if self.dim_latent % 4 or self.dim_latent > 256 or self.dim_latent & (self.dim_latent - 1) or self.dim_latent < 8:
            raise _lib.TagrecError("LightGCN: fused message dropout needs dim_latent in {8,16,...,256}")
        if torch.cuda.is_current_stream_capturing():
            raise _lib.TagrecError("LightGCN: message dropout draws a new seed on the host every step and cannot be captured "
                                   "in a HIP graph")
        self._drop_calls = getattr(self, "_drop_calls", 0) + 1
        return drops + [0.0] * (self.num_layer - len(drops)), (int(self.drop_seed) << 24) + self._drop_calls

    def _graph(self):
        return H.node_drop(self.norm_adj, self.node_drop, self.training)

    def _propagate(self):
        graph = self._graph()
        if self._fused_ok():
            drops, seed = self._drops()
            return _Propagate.apply(self.table, graph, self.num_layer, drops, seed)
        # operator-by-operator path (row folds / message dropout), same order as lightgcn.py:52-60
        x = self.table
        layers = [x]
        for k in range(self.num_layer):
            x = H.split_mm(graph, x)
            x = torch.nn.functional.dropout(x, p=self.message_drop_list[k], training=self.training)
            layers.append(H.normalize_rows(x))
        return torch.mean(torch.stack(layers, dim=1), dim=1)

    def forward(self):
        return self._split(self._propagate())

    def loss(self, batch_data):
        batch_data = batch_data.to(self.device, torch.int64).contiguous()
        nu, ni = self.num_list[0], self.num_list[1]
        if self._fused_ok():
            drops, seed = self._drops()
            fused = fused_optimizer(self) if (self.training and torch.is_grad_enabled()) else None
            res = _PropagateBprLoss.apply(self.table, self._graph(), self.num_layer, nu, ni, batch_data,
                                          H.loss_kind_id(self.loss_func), self.reg != 0, drops, seed, self.restrict_forward,
                                          fused, self.step_ws if (self.training and torch.is_grad_enabled()) else None)
            return res[0], self.reg * res[1]
        all_users, all_items = self.forward()[:2]
        ego = self.embed
        loss, reg_loss = H.triplet_loss(all_users, all_items, ego[0], ego[1], batch_data, self.loss_func)
        return loss, self.reg * reg_loss
