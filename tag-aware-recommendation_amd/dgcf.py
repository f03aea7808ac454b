"""DGCF behind the reference's model surface (/root/reference/model/dgcf.py) -- SURVEY.md 8f, N4.

    DGCF(data)                ctor reads data.num / adjacency blocks and the config                    (:11-35)
    .forward(out_A=False)     -> tuple(user_emb, item_emb[, tag_emb]); out_A=True -> per layer, per factor, the
                                 routing weights of the last iteration as sparse tensors                (:51-68)
    .loss((batch[B,3], cor))  -> (mul_loss, reg * l2reg_loss on EGO rows)                               (:115-145)
    .predict_rating(users)    -> sigmoid(U_b I^T)                                                       (:147-152)

Per layer and routing iteration the reference builds K sparse tensors and runs 3K sparse products plus two
[nnz, D/K] gathers per factor; here one iteration is: softmax over factors -> row sums -> ONE routed product over
full rows (all K factors at once) with the per-slice normalisation in its epilogue -> ONE score pass.  The routing
weights carry no gradient in the reference (`A_factor.detach()`, :93), so backward per layer is the per-slice
normalise-backward followed by one routed product with the transposed weights of the LAST iteration."""
import torch

from . import help as H
from . import routing as R
from .base import TableModel
from .config import CFG as _GLOBAL_CFG
from .graph import creat_adj


class _RoutedLayer(torch.autograd.Function):
    """`iterate_update` (dgcf.py:70-90): ego [N, D] -> per-slice normalised factor embeddings [N, D].
    `logits` ([nnz, K], the reference's A_values) is updated in place; `keep` collects the last iteration's weights."""

    @staticmethod
    def forward(ctx, ego, rg, logits, iterate_k, update_last, keep, row_masks=None):
        """row_masks (top layer of a loss; one uint8 [n] mask or None per routing iteration): the rows whose product and
        scores that iteration must form.  The last iteration's output is read at the batch rows only; the iteration
        before it must be right on their neighbours too (a row's output uses the neighbours' row sums, which come from
        the neighbours' own scores of the previous iteration); earlier ones run in full."""
        ego = ego.detach().contiguous()
        K = logits.shape[1]
        t_emb, _ = R.slice_norm_fwd(ego, K, tanh=True, want_inv=False)        # tanh(normalize(ego_split[tail])), :107-108
        for t in range(iterate_k):
            row_mask = row_masks[t] if row_masks is not None else None
            w = rg.softmax(logits)                                            # :75
            d = rg.rowsum_rsqrt(w)                                            # :95-99
            xs = R.slice_scale(ego, d)                                        # D x           (:101)
            f, h, inv = rg.spmm(w, xs, post=d, raw=True, normed=True, row_mask=row_mask)   # D A D x (:102-103), normalize (:106)
            if t < iterate_k - 1 or update_last:
                rg.score(h, t_emb, logits, accumulate=True, row_mask=row_mask)   # A_values += A_score (:84-85)
        if keep is not None:
            keep.append(w)
        ctx.rg = rg
        ctx.save_for_backward(f, inv, d, w)
        return h

    @staticmethod
    def backward(ctx, g):
        f, inv, d, w = ctx.saved_tensors
        rg = ctx.rg
        df = R.slice_norm_bwd(f, inv, g.contiguous())                         # through F.normalize (:87)
        # (D A D)^T = D A^T D; df is non-zero only on the rows the batch gradient has reached
        dx, _, _ = rg.spmm(w, R.slice_scale(df, d), post=d, transposed=True, sparse_x=True)
        return dx, None, None, None, None, None, None


class DGCF(TableModel):
    def __init__(self, data, args=None, config=None, graph=None):
        super().__init__()
        self._config(config if config is not None else _GLOBAL_CFG)
        self._init_table(data, self.use_tag, self.dim_latent, self.device)
        self.norm_adj = graph if graph is not None else creat_adj(data, self.use_tag, self.norm_type, 1, self.device)
        self.routing = R.RoutingGraph(self.norm_adj)

    def _config(self, config):
        self.dim_latent = config["dim_latent"]
        self.num_layer = len(config["dim_layer_list"])
        self.device = torch.device(config["device"])
        self.norm_type = config["norm_type"]
        self.factor_k = config["factor_k"]
        self.iterate_k = config["iterate_k"]
        self.dim_k = self.dim_latent // self.factor_k
        self.reg = config["reg"]
        self.cor_reg = config.get("cor_reg", 0)
        self.loss_func = config["mul_loss_func"]
        self.use_tag = config["use_tag"]
        self.restrict_forward = bool(config.get("restrict_forward", True))

    def forward(self, out_A=False, loss_rows=None):
        """loss_rows (node ids): the caller reads the result at these rows only (`loss`); the top layer is then
        computed on them alone."""
        rg = self.routing
        logits = torch.ones(rg.nnz, self.factor_k, dtype=torch.float32, device=self.device)    # A_values (:52)
        ego = self.table
        layers = [ego]
        keep = [] if out_A else None
        for k in range(self.num_layer):
            last = k == self.num_layer - 1
            masks = None
            if last and not out_A:
                top = rg.loss_row_mask(loss_rows)
                if top is not None:
                    masks = [None] * self.iterate_k
                    masks[-1] = top
                    if self.iterate_k >= 2:
                        masks[-2] = self.norm_adj.mark_rows(loss_rows, torch.zeros_like(top))
            ego = _RoutedLayer.apply(ego, rg, logits, self.iterate_k, not last, keep, masks)
            layers.append(ego)
        if out_A:
            idx = torch.stack([rg.rows, rg.cols])
            return [[torch.sparse_coo_tensor(idx, w[:, i].contiguous(), self.norm_adj.shape) for i in range(self.factor_k)]
                    for w in keep]
        out = torch.mean(torch.stack(layers, dim=1), dim=1)
        return self._split(out)

    def loss(self, batch_data):
        data = batch_data[0] if isinstance(batch_data, (tuple, list)) else batch_data       # (triplets, cor), :116
        data = data.to(self.device, torch.int64).contiguous()
        nu = self.num_list[0]
        rows = torch.cat([data[:, 0], data[:, 1] + nu, data[:, 2] + nu]) if self.restrict_forward else None
        all_users, all_items = self.forward(loss_rows=rows)[:2]
        ego = self.embed
        loss, reg_loss = H.triplet_loss(all_users, all_items, ego[0], ego[1], data, self.loss_func)
        return loss, self.reg * reg_loss
