"""Triplet producers with the reference's protocol (/root/reference/train_data):

    Abstract_training_data.reset() / .mini_batch()      abstract.py:4-23
    BPR_training_data(data)                              bpr_training_data.py:12-45
    TransTag_training_data(data)                         transe_training_data.py:42-70

The reference samples negatives in a forked `multiprocessing.Pool` with a Python
rejection loop per edge (train_data/utils.py:19-28) and copies the epoch's
[E,3] array to the device.  Here the epoch is sampled ON the device: uniform
draws, membership test by binary search in the sorted (user, item) keys,
re-draw of the collisions until none is left, then a device-side shuffle.
Same distribution (one uniform non-train item per train edge); the reference's
own stream is not reproducible from its seed, so parity runs use
`Fixed_training_data` with arrays shared by both sides.
"""
import numpy as np
import torch

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


def _reject_resample(keys_sorted, left, n_right, gen):
    """One uniform draw in [0, n_right) per row of `left`, re-drawn while left*n_right+draw is in keys_sorted."""
    dev = left.device
    neg = torch.randint(0, n_right, left.shape, device=dev, generator=gen)
    todo = torch.arange(left.numel(), device=dev)
    while todo.numel():
        k = left[todo] * n_right + neg[todo]
        pos = torch.searchsorted(keys_sorted, k).clamp_(max=keys_sorted.numel() - 1)
        bad = keys_sorted[pos] == k
        todo = todo[bad]
        if todo.numel():
            neg[todo] = torch.randint(0, n_right, todo.shape, device=dev, generator=gen)
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
        self._keys = torch.sort(self.pos_inter[:, 0] * self.num + self.pos_inter[:, 1]).values
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(cfg["seed"] if seed is None else seed)
        self.all_train_data = self.get_all_training_data()
        self.tot_inter = self.all_train_data.shape[0] // self.batch_size

    def get_all_training_data(self):
        u, i = self.pos_inter[:, 0], self.pos_inter[:, 1]
        neg = _reject_resample(self._keys, u, self.num, self._gen)
        data = torch.stack([u, i, neg], dim=1)
        perm = torch.randperm(data.shape[0], device=self.device, generator=self._gen)
        return data[perm].contiguous()


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
        self._left = self.uti[:, 0] * n_tag + self.uti[:, 1]            # (u, t) pair id
        self._keys = torch.sort(self._left * self.num + self.uti[:, 2]).values
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed((cfg["seed"] if seed is None else seed) + 1)
        self.all_train_data = self.get_all_training_data()
        self.tot_inter = self.all_train_data.shape[0] // self.batch_size

    def get_all_training_data(self):
        neg = _reject_resample(self._keys, self._left, self.num, self._gen)
        return torch.cat([self.uti, neg[:, None]], dim=1).contiguous()
