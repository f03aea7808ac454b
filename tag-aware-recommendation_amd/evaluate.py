"""Evaluation with the reference's protocol and result layout
(/root/reference/training/basic_test.py:30-111, training/utils.py:7-54):

    Basic_test(data).run(model) -> {"recall": [@k...], "precision": [...], "hr": [...],
                                    "ndcg": [...], "auc": [x]}          means over the test users

Per 512-user batch: `model.predict_rating` -> train positives masked with -1024
(basic_test.py:47) -> torch.topk(max(topks)) -> hit labels.  The reference ships
the top-k to a multiprocessing pool and runs sklearn's AUC per user on the host;
here labels, recall/precision/hr/ndcg and a rank-based AUC are computed on the
device from sorted (user, item) keys, one host read at the end.
"""
import numpy as np
import torch

from . import _lib
from .config import CFG as _GLOBAL_CFG

FUSED_WIDTHS = (16, 32, 64, 128, 192, 256, 384, 512)


def fused_topk(user_table, item_table, users, train_ptr, train_items, k):
    """Top-k item ids per user by sigmoid(u . i), train positives excluded -- one fused HIP pass over the item
    table (csrc/eval.hip) instead of a [users, n_item] rating matrix + mask + torch.topk."""
    U = _lib.require_gpu_tensor(user_table.contiguous(), torch.float32, "eval user table")
    I = _lib.require_gpu_tensor(item_table.contiguous(), torch.float32, "eval item table")
    users = users.to(U.device, torch.int64).contiguous()
    top = torch.empty(users.numel(), k, dtype=torch.int64, device=U.device)
    val = torch.empty(users.numel(), k, dtype=torch.float32, device=U.device)
    _lib.check(_lib.load().tagrec_eval_topk_f32(_lib.ptr(U), _lib.ptr(I), I.shape[0], U.shape[1], _lib.ptr(users),
                                                users.numel(), _lib.ptr(train_ptr), _lib.ptr(train_items), k,
                                                _lib.ptr(top), _lib.ptr(val), _lib.stream_ptr()), "eval_topk")
    return top, val


def _edge_keys(user_items, n_item, device):
    if isinstance(user_items, dict):
        us = np.fromiter((u for u, its in user_items.items() for _ in its), dtype=np.int64)
        its = np.fromiter((i for its in user_items.values() for i in its), dtype=np.int64)
    elif isinstance(user_items, torch.Tensor):       # [E,2] tensor, possibly already on the device
        u = user_items[:, 0].to(device, torch.int64)
        i = user_items[:, 1].to(device, torch.int64)
        return u, i, torch.sort(u * n_item + i).values
    else:                                   # [E,2] array
        arr = np.asarray(user_items)
        us, its = arr[:, 0].astype(np.int64), arr[:, 1].astype(np.int64)
    u = torch.from_numpy(us).to(device)
    i = torch.from_numpy(its).to(device)
    key = torch.sort(u * n_item + i).values
    return u, i, key


def _member(keys_sorted, k):
    if keys_sorted.numel() == 0:
        return torch.zeros_like(k, dtype=torch.bool)
    pos = torch.searchsorted(keys_sorted, k).clamp_(max=keys_sorted.numel() - 1)
    return keys_sorted[pos] == k


def minibatch(data, batch_size):
    """training/utils.py:48-54 (a trailing empty slice is produced when len % batch == 0)."""
    step = len(data) // batch_size + 1
    for i in range(step):
        yield data[i * batch_size:(i + 1) * batch_size]


class Basic_test:
    def __init__(self, data, args=None, config=None, with_auc=None):
        self.cfg = config if config is not None else _GLOBAL_CFG
        self.device = torch.device(self.cfg["device"])
        self.n_item = data.num["item"]
        self.n_user = data.num["user"]
        self.train_u, self.train_i, _ = _edge_keys(data.user_items["train"], self.n_item, self.device)
        self.sets = {}
        names = ["test"] + (["val"] if self.cfg.get("has_val") else [])
        for name in names:
            u, i, key = _edge_keys(data.user_items[name], self.n_item, self.device)
            cnt = torch.bincount(u, minlength=self.n_user)
            self.sets[name] = (key, cnt)
        order = torch.argsort(self.train_u, stable=True)
        self.train_u, self.train_i = self.train_u[order], self.train_i[order]
        self.train_ptr = torch.zeros(self.n_user + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(torch.bincount(self.train_u, minlength=self.n_user), 0, out=self.train_ptr[1:])
        self.with_auc = (self.n_item <= 50_000) if with_auc is None else with_auc
        # per-user SORTED train items for the fused kernel's mask look-up
        skey = torch.sort(self.train_u * self.n_item + self.train_i).values
        self.train_items_sorted = (skey % self.n_item).to(torch.int32).contiguous()
        self.fused = bool(self.cfg.get("eval_fused", True))

    @torch.no_grad()
    def run(self, model, istest=False, group_k=0, all_users=None):
        model.eval()
        name = "val" if (not istest and self.cfg.get("has_val")) else "test"
        key, cnt = self.sets[name]
        if all_users is None:
            all_users = torch.nonzero(cnt > 0).flatten()
        else:
            all_users = torch.as_tensor(all_users, dtype=torch.int64, device=self.device)
        topks = list(self.cfg["topks"])
        kmax = max(topks)
        disc = 1.0 / torch.log2(torch.arange(2, kmax + 2, device=self.device, dtype=torch.float64))
        sums = {m: torch.zeros(len(topks), dtype=torch.float64, device=self.device)
                for m in ("recall", "precision", "hr", "ndcg")}
        auc_sum = torch.zeros((), dtype=torch.float64, device=self.device)

        def score(users, top):
            label = _member(key, users[:, None] * self.n_item + top.clamp_min(0))            # get_label
            # the fused kernel pads a list with id -1 when a user has fewer than k un-masked items: never a hit
            label = (label & (top >= 0)).to(torch.float64)
            n_true = cnt[users].to(torch.float64)
            for j, k in enumerate(topks):
                right = label[:, :k].sum(1)
                sums["recall"][j] += (right / n_true).sum()
                sums["precision"][j] += right.sum() / k
                sums["hr"][j] += (right > 0).sum()
                ideal = torch.cumsum(disc[:k], 0)[(torch.clamp(n_true, max=k) - 1).long()]
                sums["ndcg"][j] += ((label[:, :k] * disc[:k]).sum(1) / ideal).sum()

        tables = model.forward()[:2] if (self.fused and not self.with_auc and hasattr(model, "forward")) else None
        if tables is not None and tables[0].shape[1] in FUSED_WIDTHS and tables[0].is_cuda and kmax <= 64:
            # one propagation, one fused score/mask/top-k pass; metrics in user chunks to bound temporaries
            top, _ = fused_topk(tables[0], tables[1], all_users, self.train_ptr, self.train_items_sorted, kmax)
            for lo in range(0, all_users.numel(), 1 << 18):
                score(all_users[lo:lo + (1 << 18)], top[lo:lo + (1 << 18)])
            n = float(all_users.numel())
            out = {m: (v / n).cpu().tolist() for m, v in sums.items()}
            out["auc"] = [float("nan")]
            return out
        for users in minibatch(all_users, self.cfg["test_batch"]):
            if users.numel() == 0:
                continue
            rating = model.predict_rating(users)
            # mask the users' train items (basic_test.py:42-47)
            lo, hi = self.train_ptr[users], self.train_ptr[users + 1]
            deg = hi - lo
            row = torch.repeat_interleave(torch.arange(users.numel(), device=self.device), deg)
            start = torch.repeat_interleave(lo - torch.cumsum(deg, 0) + deg, deg)
            col = self.train_i[start + torch.arange(row.numel(), device=self.device)]
            rating[row, col] = -(1 << 10)
            _, top = torch.topk(rating, k=kmax)
            score(users, top)
            if self.with_auc:
                auc_sum += self._auc(rating, users, key)
        n = float(all_users.numel())
        out = {m: (v / n).cpu().tolist() for m, v in sums.items()}
        out["auc"] = [float(auc_sum.cpu()) / n] if self.with_auc else [float("nan")]
        return out

    def _auc(self, rating, users, key):
        """Per-user ROC AUC over the un-masked items (training/utils.py:37-45), as the
        Mann-Whitney statistic with average ranks for ties (what sklearn computes)."""
        n_item = rating.shape[1]
        items = torch.arange(n_item, device=self.device)
        pos = _member(key, users[:, None] * self.n_item + items[None, :])
        valid = rating >= 0
        r = rating.double().masked_fill(~valid, -1.0)
        srt, idx = torch.sort(r, dim=1)
        # average rank of ties: (first index + last index)/2 + 1 over equal values
        first = torch.searchsorted(srt, srt, right=False)
        last = torch.searchsorted(srt, srt, right=True)
        avg_rank_sorted = (first + last + 1).double() / 2.0
        ranks = torch.empty_like(avg_rank_sorted).scatter_(1, idx, avg_rank_sorted)
        n_invalid = (~valid).sum(1, keepdim=True).double()
        ranks = ranks - n_invalid                       # ranks among valid items only
        posv = pos & valid
        n_pos = posv.sum(1).double()
        n_neg = valid.sum(1).double() - n_pos
        u_stat = (ranks * posv).sum(1) - n_pos * (n_pos + 1) / 2.0
        return (u_stat / (n_pos * n_neg)).sum()
