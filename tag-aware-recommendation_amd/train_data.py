"""Triplet producers with the reference's protocol (/root/reference/train_data):

    Abstract_training_data.reset() / .mini_batch()      abstract.py:4-23
    BPR_training_data(data)                              bpr_training_data.py:12-45
    TransTag_training_data(data)                         transe_training_data.py:42-70

The reference samples negatives in a forked `multiprocessing.Pool` with a Python
rejection loop per edge (train_data/utils.py:19-28) and copies the epoch's
[E,3] array to the device.  Here the epoch is sampled ON the device by a HIP
kernel (csrc/rowops.hip `sample_negative_kernel`): per edge, counter-based
uniform draws, membership test by binary search in the user's sorted positive
list, re-draw until a non-positive comes up; then a device-side shuffle.
Same distribution (one uniform non-train item per train edge); the reference's
own stream is not reproducible from its seed, so parity runs use
`Fixed_training_data` with arrays shared by both sides.
"""
import numpy as np
import torch

from . import _lib
from .config import CFG as _GLOBAL_CFG


class Abstract_training_data:
    def __init__(self, args=None, config=None):
        cfg = config if config is not None else _GLOBAL_CFG
        self.device = torch.device(cfg["device"])
        self.cpu_core = cfg["cpu_core"]
        self.all_train_data = None
        self.batch_size = cfg["train_batch"]

    def get_all_training_data(self):
        raise NotImplementedError

    def reset(self):
        self.all_train_data = self.get_all_training_data()

    def mini_batch(self):
        """abstract.py:17-23, the loop as written: once fewer than 2*batch rows remain the slice
        runs to the end -- and the loop still advances, so a short final slice repeats the tail."""
        n = self.all_train_data.shape[0]
        for i in range(0, n, self.batch_size):
            if i + 2 * self.batch_size > n:
                yield self.all_train_data[i:]
            else:
                yield self.all_train_data[i:i + self.batch_size]


class _Positives:
    """Sorted positive lists per left id as a device CSR (rowptr int64, cols int32), built once."""

    def __init__(self, left, right, n_left, n_right):
        key = torch.unique(left * n_right + right)
        l = torch.div(key, n_right, rounding_mode="floor")
        self.rowptr = torch.zeros(n_left + 1, dtype=torch.int64, device=left.device)
        torch.cumsum(torch.bincount(l, minlength=n_left), 0, out=self.rowptr[1:])
        self.cols = (key - l * n_right).to(torch.int32).contiguous()
        self.n_left, self.n_right = int(n_left), int(n_right)

    def sample(self, left, seed):
        """One uniform non-positive draw per entry of `left` (HIP kernel, counter-based generator)."""
        left = left.contiguous()
        neg = torch.empty_like(left)
        _lib.check(_lib.load().tagrec_sample_negative_i64(_lib.ptr(left), left.numel(), _lib.ptr(self.rowptr),
                                                          _lib.ptr(self.cols), self.n_left, self.n_right, int(seed),
                                                          _lib.ptr(neg), _lib.stream_ptr()), "sample_negative")
        return neg


class BPR_training_data(Abstract_training_data):
    def __init__(self, data, args=None, config=None, seed=None):
        super().__init__(args, config)
        cfg = config if config is not None else _GLOBAL_CFG
        self.num = data.num["item"]
        self.num_user = data.num["user"]
        pos = data.edge_index["train"]
        self.pos_inter = (pos if isinstance(pos, torch.Tensor) else torch.from_numpy(np.asarray(pos))).to(
            self.device, torch.int64)
        self._pos = _Positives(self.pos_inter[:, 0], self.pos_inter[:, 1], self.num_user, self.num)
        self._seed = int(cfg["seed"] if seed is None else seed)
        self._epoch = 0
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self._seed)
        self.all_train_data = self.get_all_training_data()
        self.tot_inter = self.all_train_data.shape[0] // self.batch_size

    def get_all_training_data(self):
        u, i = self.pos_inter[:, 0].contiguous(), self.pos_inter[:, 1]
        neg = self._pos.sample(u, (self._seed << 20) + self._epoch)      # a fresh stream every epoch
        self._epoch += 1
        data = torch.stack([u, i, neg], dim=1)
        perm = torch.randperm(data.shape[0], device=self.device, generator=self._gen)
        return data[perm].contiguous()


class DGCF_training_data(Abstract_training_data):
    """The per-batch sampler DGCF / DisenGCN train with (train_data/bpr_training_data.py:47-83, utils.py:58-80):
    every mini-batch draws `train_batch` users (without replacement when there are more users than that), one of
    the user's train items and one rejected-negative item, plus `cor_batch` random ids per node type (the unused
    `cor` half).  E // B + 1 batches per epoch; `reset()` does nothing.  Drawn on the device; statistical parity
    only (the reference draws from Python's and numpy's global generators)."""

    def __init__(self, data, args=None, config=None, seed=None):
        super().__init__(args, config)
        cfg = config if config is not None else _GLOBAL_CFG
        self.cor_batch = cfg.get("cor_batch", 100)
        self.use_tag = cfg["use_tag"]
        self.num_item, self.num_user, self.num_tag = data.num["item"], data.num["user"], data.num.get("tag", 0)
        pos = data.edge_index["train"]
        pos = (pos if isinstance(pos, torch.Tensor) else torch.from_numpy(np.asarray(pos))).to(self.device, torch.int64)
        self._pos = _Positives(pos[:, 0], pos[:, 1], self.num_user, self.num_item)
        deg = self._pos.rowptr[1:] - self._pos.rowptr[:-1]
        self._users = torch.nonzero(deg > 0).flatten()                 # keys of user_items['train']
        self.tot_inter = pos.shape[0] // self.batch_size + 1
        self._seed = int(cfg["seed"] if seed is None else seed)
        self._draws = 0
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self._seed)

    def mini_sample(self):
        B, nu = self.batch_size, self._users.numel()
        if nu > B:
            pick = torch.randperm(nu, device=self.device, generator=self._gen)[:B]
        else:
            pick = torch.randint(0, nu, (B,), device=self.device, generator=self._gen)
        u = self._users[pick]
        lo, hi = self._pos.rowptr[u], self._pos.rowptr[u + 1]
        r = torch.rand(B, device=self.device, generator=self._gen)
        at = torch.minimum(lo + (r * (hi - lo)).long(), hi - 1)
        pos_i = self._pos.cols[at].long()
        neg_i = self._pos.sample(u, (self._seed << 20) + self._draws)
        self._draws += 1
        cor = [torch.randperm(n, device=self.device, generator=self._gen)[:self.cor_batch]
               for n in ([self.num_user, self.num_item] + ([self.num_tag] if self.use_tag else []))]
        k = min(c.numel() for c in cor)
        return torch.stack([u, pos_i, neg_i], dim=1), torch.stack([c[:k] for c in cor])

    def reset(self):
        pass

    def mini_batch(self):
        for _ in range(self.tot_inter):
            yield self.mini_sample()


class KGAT_training_data(Abstract_training_data):
    """TransE-phase producer of KGAT (train_data/transe_training_data.py:12-41): all (head, relation, tail) triplets of
    `data.create_edge()` in relation order; batch i is the window `all_triplet[i : i + transe_batch]` -- consecutive
    windows overlap in all but one row, as the reference's loop is written (:37-38) -- with one uniform negative
    tail per row, rejected while (head, relation, tail) is a known triplet.  `reset()` does nothing."""

    def __init__(self, data, args=None, config=None, seed=None):
        super().__init__(args, config)
        cfg = config if config is not None else _GLOBAL_CFG
        self.batch_size = cfg["transe_batch"]
        self.num = data.num["user"] + data.num["item"] + data.num["tag"]
        parts = []
        for k, e in data.create_edge().items():
            e = torch.as_tensor(np.asarray(e), dtype=torch.int64)      # [2, E] (TGCN_load.create_edge); rows: head, tail
            parts.append(torch.stack([e[0], torch.full_like(e[0], int(k)), e[1]], dim=1))
        self.all_triplet = torch.cat(parts).to(self.device)
        n_rel = int(self.all_triplet[:, 1].max()) + 1 if self.all_triplet.numel() else 1
        self._n_rel = n_rel
        self._pos = _Positives(self.all_triplet[:, 0] * n_rel + self.all_triplet[:, 1], self.all_triplet[:, 2], self.num * n_rel,
                               self.num)
        self.tot_inter = self.all_triplet.shape[0] // self.batch_size
        self._seed = int(cfg["seed"] if seed is None else seed) + 2
        self._draws = 0

    def reset(self):
        pass

    def mini_batch(self):
        for i in range(self.tot_inter):
            batch = self.all_triplet[i:i + self.batch_size]
            left = (batch[:, 0] * self._n_rel + batch[:, 1]).contiguous()
            neg = self._pos.sample(left, (self._seed << 20) + self._draws)
            self._draws += 1
            yield torch.cat([batch, neg[:, None]], dim=1)


class Fixed_training_data(Abstract_training_data):
    """Replays given per-epoch triplet arrays (parity runs against the CPU oracle)."""

    def __init__(self, epochs, batch_size, device):
        self.device = torch.device(device)
        self.batch_size = batch_size
        self._epochs = [torch.as_tensor(e, dtype=torch.int64).to(self.device) for e in epochs]
        self._next = 0
        self.all_train_data = self._epochs[0]

    def get_all_training_data(self):
        out = self._epochs[self._next % len(self._epochs)]
        self._next += 1
        return out


class TransTag_training_data(Abstract_training_data):
    """(user, tag, pos_item, neg_item) rows: one negative item per (u, i, t) assignment, rejected
    while (u, t, item) is an assignment (transe_training_data.py:42-70, utils.py:31-40).  Not
    shuffled, re-sampled on every reset() (the reference inherits `reset`)."""

    def __init__(self, data, args=None, config=None, seed=None):
        super().__init__(args, config)
        cfg = config if config is not None else _GLOBAL_CFG
        self.batch_size = cfg["transtag_batch"]
        self.num = data.num["item"]
        n_tag = data.num["tag"]
        uit = data.uit_data if isinstance(data.uit_data, torch.Tensor) else torch.from_numpy(np.asarray(data.uit_data))
        uit = uit.to(self.device, torch.int64)
        self.uti = uit[:, [0, 2, 1]].contiguous()
        # positives of a (user, tag) pair = the items it was assigned to; pair ids are compacted so the CSR stays small
        pair = self.uti[:, 0] * n_tag + self.uti[:, 1]
        upair, self._left = torch.unique(pair, return_inverse=True)
        self._pos = _Positives(self._left, self.uti[:, 2], upair.numel(), self.num)
        self._seed = int(cfg["seed"] if seed is None else seed) + 1
        self._epoch = 0
        self.all_train_data = self.get_all_training_data()
        self.tot_inter = self.all_train_data.shape[0] // self.batch_size

    def get_all_training_data(self):
        neg = self._pos.sample(self._left, (self._seed << 20) + self._epoch)
        self._epoch += 1
        return torch.cat([self.uti, neg[:, None]], dim=1).contiguous()
