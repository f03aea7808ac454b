"""KGAT behind the reference's model surface (/root/reference/model/kgat.py) -- SURVEY.md 8f, N4.

    KGAT(data)               data.num + data.create_edge(): dict relation -> 2-D integer array             (:11-22)
    .forward()               -> (user_embed, entity_embed); entity = items then tags                        (:63-104)
    .loss(batch[B,3])        -> (mul_loss, reg * l2reg_loss on the propagated rows); items index entities   (:143-153)
    .transe_loss(batch[B,4]) -> (mean softplus(|h W_r + r - t+ W_r|^2 - |h W_r + r - t- W_r|^2), cor_reg * l2reg)   (:155-162)
    .predict_rating(users)   -> sigmoid(U_b E^T), [b, n_entity]                                             (:164-169)

The edge arrays are read exactly as the reference reads them, head = e[:, 0], tail = e[:, 1] (:71-72): the [E, 2]
arrays of `KGAT_load.get_relation_dict` give E edges per relation; the [2, E] arrays of `TGCN_load.create_edge`,
which is what com.py:78-79 passes, give two.  `agg_type` other than "bi_inter" (the reference's own default is
"bi_agg") means no propagation at all (:100-101).

Mechanism.  The reference gathers [E, D] head / tail rows per relation and multiplies each by W_r; here every node is
projected once per relation (P_r = E W_r, one batched GEMM) and the attention logit of an edge is the per-entry
score < tanh(P_r[head] + e_r), P_r[tail] > (csrc/routing.hip `route_score`).  The six edge lists are merged into one
CSR whose duplicate entries are summed (what torch.sparse.softmax's coalesce does), `row_softmax` normalises each
row, and every layer's `split_mm` is the routed product with those values.  Unlike DGCF the attention values carry
gradient (:96 has no detach): d value = < dY[row], X[col] > is another score pass, d logits the row-softmax
backward, and the score's own backward two routed products (one over the transposed relation graph)."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, help as H
from . import routing as R
from .config import CFG as _GLOBAL_CFG


class KGAT(nn.Module):
    def __init__(self, data, args=None, config=None):
        super().__init__()
        self._config(config if config is not None else _GLOBAL_CFG)
        if self.device.type != "cuda":
            raise _lib.TagrecError("KGAT: tagrec_amd needs a GPU device (no CPU path)")
        _lib.load()
        self.num_user = data.num["user"]
        self.num_entity = data.num["item"] + data.num["tag"]
        self.num_relation = 6
        self.edge_index_dict = {k: torch.as_tensor(np.asarray(v)).to(self.device) for k, v in data.create_edge().items()}
        self._init_weight()
        self._build_structure()
        self._eval_cache = None

    def _config(self, config):
        self.dim_latent = config["dim_latent"]
        self.dim_relation = config["dim_relation"]
        self.dim_layer_list = list(config["dim_layer_list"])
        self.num_layer = len(self.dim_layer_list)
        self.dim_layer_list = [self.dim_latent] + self.dim_layer_list
        self.agg_type = config["agg_type"]
        self.device = torch.device(config["device"])
        self.message_drop_list = config["message_drop_list"]
        self.reg = config["reg"]
        self.cor_reg = config["cor_reg"]
        self.loss_func = config["mul_loss_func"]

    def _init_weight(self):
        D, Dr = self.dim_latent, self.dim_relation
        self.embed = nn.ParameterDict({
            "user": nn.Parameter(torch.empty(self.num_user, D)),
            "entity": nn.Parameter(torch.empty(self.num_entity, D)),
            "relation": nn.Parameter(torch.empty(self.num_relation, Dr)),
        })
        self.mat = nn.ParameterDict({"transE": nn.Parameter(torch.empty(self.num_relation, D, Dr))})
        for k in range(self.num_layer):
            din, dout = self.dim_layer_list[k], self.dim_layer_list[k + 1]
            self.mat[f"W1_{k}"] = nn.Parameter(torch.empty(din, dout))
            self.mat[f"b1_{k}"] = nn.Parameter(torch.empty(1, dout))
            if self.agg_type == "bi_inter":
                self.mat[f"W2_{k}"] = nn.Parameter(torch.empty(din, dout))
                self.mat[f"b2_{k}"] = nn.Parameter(torch.empty(1, dout))
        for p in self.parameters():                     # xavier on everything, in parameter order (:58-60)
            nn.init.xavier_uniform_(p)
        self.to(self.device)

    def _build_structure(self):
        """Per relation: a CSR over its edges (duplicates kept).  Merged: one CSR over the distinct (head, tail) pairs
        of all relations, plus each relation entry's position in it (duplicates are summed, as coalesce does)."""
        n = self.num_user + self.num_entity
        self._rel = []
        keys = []
        for k in sorted(self.edge_index_dict.keys()):
            e = self.edge_index_dict[k]
            rows, cols = e[:, 0].long(), e[:, 1].long()            # kgat.py:71-72, whatever the array's layout
            if rows.numel() == 0:
                continue
            rg, order = R.RoutingGraph.from_edges(rows, cols, n, self.device)
            self._rel.append((k, rg))
            keys.append(rg.rows * n + rg.cols)
        if not keys:
            self._merged, self._maps = None, []
            return
        uniq, inverse = torch.unique(torch.cat(keys), return_inverse=True)
        self._merged, _ = R.RoutingGraph.from_edges(torch.div(uniq, n, rounding_mode="floor"), uniq % n, n, self.device)
        self._maps = list(torch.split(inverse, [int(k.numel()) for k in keys]))

    # ---------------------------------------------------------------------------------------------- forward
    def _attention(self, all_embed):
        logits = torch.zeros(self._merged.nnz, dtype=torch.float32, device=self.device)
        for (k, rg), pos in zip(self._rel, self._maps):
            proj = torch.matmul(all_embed, self.mat["transE"][k])                     # every node under relation k
            head = torch.tanh(proj + self.embed["relation"][k])
            logits = logits.index_add(0, pos, R.edge_score(head, proj, rg))
        return R.row_softmax(logits, self._merged)

    def forward(self):
        all_embed = torch.cat([self.embed["user"], self.embed["entity"]], dim=0)
        if self.agg_type == "bi_inter" and self._merged is not None:
            att = self._attention(all_embed)
            outs = [all_embed]
            for k in range(self.num_layer):
                nei = R.valued_spmm(att, all_embed, self._merged)
                s = torch.nn.functional.leaky_relu(torch.matmul(nei + all_embed, self.mat[f"W1_{k}"] + self.mat[f"b1_{k}"]), 0.2)
                b = torch.nn.functional.leaky_relu(torch.matmul(nei * all_embed, self.mat[f"W2_{k}"] + self.mat[f"b2_{k}"]), 0.2)
                all_embed = s + b
                all_embed = torch.nn.functional.dropout(all_embed, p=self.message_drop_list[k], training=self.training)
                outs.append(H.normalize_rows(all_embed))
            all_embed = torch.cat(outs, dim=1)
        return all_embed[:self.num_user], all_embed[self.num_user:]

    def get_embed(self, batch_data):
        head, rela, pos_tail, neg_tail = batch_data.to(self.device, torch.int64).T
        all_embed = torch.cat([self.embed["user"], self.embed["entity"]], dim=0)
        r_e = self.embed["relation"][rela]
        trans = self.mat["transE"][rela]
        proj = lambda idx: torch.matmul(all_embed[idx].unsqueeze(1), trans).squeeze(1)
        return proj(head), r_e, proj(pos_tail), proj(neg_tail)

    def loss(self, batch_data):
        batch_data = batch_data.to(self.device, torch.int64).contiguous()
        all_users, all_items = self.forward()[:2]
        all_users, all_items = all_users.contiguous(), all_items.contiguous()
        loss, reg_loss = H.triplet_loss(all_users, all_items, all_users, all_items, batch_data, self.loss_func)
        return loss, self.reg * reg_loss

    def transe_loss(self, batch_data):
        h_e, r_e, pos_t_e, neg_t_e = self.get_embed(batch_data)
        pos_score = torch.norm(h_e + r_e - pos_t_e, p=2, dim=1).pow(2)
        neg_score = torch.norm(h_e + r_e - neg_t_e, p=2, dim=1).pow(2)
        kg_loss = torch.mean(torch.nn.functional.softplus(pos_score - neg_score))
        return kg_loss, self.cor_reg * H.l2reg_loss(h_e, r_e, pos_t_e, neg_t_e)

    def train(self, mode=True):
        self._eval_cache = None
        return super().train(mode)

    def predict_rating(self, users):
        if self.training or self._eval_cache is None:
            with torch.no_grad():
                cache = self.forward()[:2]
            if not self.training:
                self._eval_cache = cache
        else:
            cache = self._eval_cache
        return torch.sigmoid(torch.matmul(cache[0][users.to(self.device)], cache[1].t()))
