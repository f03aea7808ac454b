"""DisenGCN behind the reference's model surface (/root/reference/model/disengcn.py) -- SURVEY.md 8f, N4.

    DisenGCN(data)            three tables (user, item, tag -- always, :51) + per layer W [K, D, D/K], b [K, 1, D/K]
    .forward()                -> tuple(user_emb, item_emb, tag_emb): the LAST layer's output only          (:90-103)
    .loss((batch[B,3], cor))  -> (mul_loss, reg * l2reg_loss on the PROPAGATED rows)                       (:105-131)

Layer (:23-46): f = normalize(LeakyReLU_0.2(x (W + b))) per factor; `iterate_k` rounds of
p = softmax_k <new_f[head], f[tail]>;  new_f_k = normalize(f_k + A(p_k) f_k).  The K [D, D/K] projections are ONE
[D, D] GEMM on the [n, D] layout (factor k = columns k D/K ..); every routing round is one score pass + one routed
product with the self term and the per-slice normalisation in its epilogue.  p is detached in the reference
(:37), so only the last round's product is differentiated: dF = dRaw + A(p)^T dRaw."""
import torch
import torch.nn as nn

from . import _lib, help as H
from . import routing as R
from .base import TableModel
from .config import CFG as _GLOBAL_CFG
from .graph import creat_adj


class _Route(torch.autograd.Function):
    """f [n, D] (per-slice normalised) -> new_f after `iterate_k` routing rounds (disengcn.py:29-44)."""

    @staticmethod
    def forward(ctx, f, rg, K, iterate_k, row_mask=None):
        """row_mask (last layer of a loss): only these rows of the result are read, so every routing round scores and
        aggregates their entries alone (a row's rounds depend on its own earlier rounds and on f, not on other rows')."""
        f = f.detach().contiguous()
        logits = (torch.zeros if row_mask is not None else torch.empty)(rg.nnz, K, dtype=torch.float32, device=f.device)
        new_f = f
        for _ in range(iterate_k):
            rg.score(new_f, f, logits, accumulate=False, row_mask=row_mask)   # <head, tail> per factor (:31-33)
            w = rg.softmax(logits)                                            # :34
            raw, new_f, inv = rg.spmm(w, f, self_add=f, raw=True, normed=True, row_mask=row_mask)   # f + A(p) f, normalize (:40-42)
        ctx.rg = rg
        ctx.save_for_backward(raw, inv, w)
        return new_f

    @staticmethod
    def backward(ctx, g):
        raw, inv, w = ctx.saved_tensors
        rg = ctx.rg
        draw = R.slice_norm_bwd(raw, inv, g.contiguous())
        df, _, _ = rg.spmm(w, draw, self_add=draw, transposed=True, sparse_x=True)
        return df, None, None, None, None


class Layer(nn.Module):
    def __init__(self, fac_k, iter_k, in_dim, out_dim):
        super().__init__()
        self.fac_k, self.iter_k, self.in_dim, self.out_dim = fac_k, iter_k, in_dim, out_dim
        dim_k = out_dim // fac_k
        self.W = nn.Parameter(torch.empty(fac_k, in_dim, dim_k))
        self.b = nn.Parameter(torch.empty(fac_k, 1, dim_k))

    def forward(self, rg, all_emb, row_mask=None):
        # K projections x (W_k + b_k) as one GEMM: column block k of the [in, out] matrix is W_k + b_k (:24)
        wb = (self.W + self.b).permute(1, 0, 2).reshape(self.in_dim, self.out_dim)
        f = torch.nn.functional.leaky_relu(torch.matmul(all_emb, wb), 0.2)
        f = R.slice_normalize(f, self.fac_k)
        return _Route.apply(f, rg, self.fac_k, self.iter_k, row_mask)


class DisenGCN(TableModel):
    def __init__(self, data, args=None, config=None, graph=None):
        super().__init__()
        self._config(config if config is not None else _GLOBAL_CFG)
        self._init_table(data, True, self.dim_latent, self.device)          # num_list always has the tag table (:51)
        self.norm_adj = graph if graph is not None else creat_adj(data, self.use_tag, self.norm_type, 1, self.device)
        if self.norm_adj.shape[0] != self.table.shape[0]:
            raise _lib.TagrecError("DisenGCN: the adjacency must cover users, items and tags (use_tag=True), as in the reference")
        self.routing = R.RoutingGraph(self.norm_adj)
        self.layer = nn.ModuleList(Layer(self.factor_k, self.iterate_k, self.dim_latent, self.dim_latent)
                                   for _ in range(self.num_layer))
        for lyr in self.layer:                                              # xavier over all parameters, in order (:80-82)
            nn.init.xavier_uniform_(lyr.W)
            nn.init.xavier_uniform_(lyr.b)
        self.layer.to(self.device)

    def _config(self, config):
        self.dim_latent = config["dim_latent"]
        self.num_layer = len(config["dim_layer_list"])
        self.device = torch.device(config["device"])
        self.norm_type = config["norm_type"]
        self.factor_k = config["factor_k"]
        self.iterate_k = config["iterate_k"]
        self.dim_k = self.dim_latent // self.factor_k
        self.reg = config["reg"]
        self.loss_func = config["mul_loss_func"]
        self.use_tag = config["use_tag"]
        self.message_drop_list = config["message_drop_list"]
        self.restrict_forward = bool(config.get("restrict_forward", True))

    def forward(self, loss_rows=None):
        """loss_rows (node ids): the caller reads the result at these rows only (`loss`): the last layer's routing
        then runs on them alone."""
        x = self.table
        for i, lyr in enumerate(self.layer):
            mask = self.routing.loss_row_mask(loss_rows) if i == len(self.layer) - 1 else None
            x = lyr(self.routing, x, mask)
            x = torch.nn.functional.dropout(x, p=self.message_drop_list[i], training=self.training)
        return self._split(x)

    def loss(self, batch_data):
        data = batch_data[0] if isinstance(batch_data, (tuple, list)) else batch_data
        data = data.to(self.device, torch.int64).contiguous()
        nu = self.num_list[0]
        rows = torch.cat([data[:, 0], data[:, 1] + nu, data[:, 2] + nu]) if self.restrict_forward else None
        all_users, all_items = self.forward(loss_rows=rows)[:2]
        loss, reg_loss = H.triplet_loss(all_users, all_items, all_users, all_items, data, self.loss_func)
        return loss, self.reg * reg_loss
