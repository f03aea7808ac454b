"""Dataset ingest in the reference's on-disk formats (row N3 of SURVEY.md 8f).

    CF_load(root, dataset)     train.txt / test.txt [/ val.txt]: one line per user, `u i1 i2 ...`
                               (/root/reference/data/cf_load.py:9-28, data/utils.py:23-46)
    TGCN_load(root, dataset)   + user_item_tag.txt: one `u i t` triple per line, duplicates removed
                               (/root/reference/data/tgcn_load.py:12-25, data/utils.py:9-20)

Same attributes as the reference's loader objects: `.num` (user/item[/tag/weight]), `.user_items`
(dict split -> {user: [items]}), `.edge_index` (dict split -> int array [E,2]), `.ui_adj/.ut_adj/.it_adj`
(COO with unit data; repeated (user,tag)/(item,tag) pairs are summed when the adjacency is built, which
is where the integer edge weights come from), `.uit_data`.  Host-side text parsing with numpy; the
result feeds `creat_adj`, the producers and `Basic_test` unchanged.
"""
import os

import numpy as np

from .synth import Coo, Dataset


def read_interaction_data(path):
    """`u i1 i2 ...` per line -> {u: [distinct items]}; a user on several lines is merged; users with no
    item are dropped (data/utils.py:23-46).  Item order inside a list is not meaningful in the reference
    (it goes through `set`); here it is first-appearance order."""
    out = {}
    with open(path, "r") as f:
        for line in f:
            parts = line.split()
            if len(parts) < 2:
                continue
            u = int(parts[0])
            seen = out.setdefault(u, [])
            have = set(seen)
            for tok in parts[1:]:
                i = int(tok)
                if i not in have:
                    have.add(i)
                    seen.append(i)
    return {u: its for u, its in out.items() if its}


def dict_to_edges(d):
    us = np.fromiter((u for u, its in d.items() for _ in its), dtype=np.int64)
    its = np.fromiter((i for its in d.values() for i in its), dtype=np.int64)
    return np.stack([us, its], axis=1) if len(us) else np.zeros((0, 2), np.int64)


def read_knowledge_data(path):
    """`u i t` per line, int32, duplicate rows removed and rows sorted (np.unique(axis=0), data/utils.py:9-20)."""
    data = np.loadtxt(path, dtype=np.int32, ndmin=2)
    return np.unique(data, axis=0)


class CF_load(Dataset):
    def __init__(self, data_root, dataset, has_val=False):
        super().__init__()
        self.file_dir = os.path.join(data_root, dataset)
        splits = ["train"] + (["val"] if has_val else []) + ["test"]
        maxes = []
        for s in splits:
            self.user_items[s] = read_interaction_data(os.path.join(self.file_dir, s + ".txt"))
            self.edge_index[s] = dict_to_edges(self.user_items[s])
            if len(self.edge_index[s]):
                maxes.append(self.edge_index[s].max(axis=0))
        mx = np.max(np.stack(maxes), axis=0)
        self.num = {"user": int(mx[0]) + 1, "item": int(mx[1]) + 1}          # cf_load.py:23
        tr = self.edge_index["train"]
        self.ui_adj = Coo(tr[:, 0], tr[:, 1], np.ones(len(tr), np.float32), (self.num["user"], self.num["item"]))


class TGCN_load(CF_load):
    def __init__(self, data_root, dataset, has_val=False):
        super().__init__(data_root, dataset, has_val)
        self.uit_data = read_knowledge_data(os.path.join(self.file_dir, "user_item_tag.txt"))
        uit = self.uit_data.astype(np.int64)
        self.num["tag"] = int(uit[:, 2].max()) + 1                            # tgcn_load.py:19
        ones = np.ones(len(uit), np.float32)
        self.ut_adj = Coo(uit[:, 0], uit[:, 2], ones, (self.num["user"], self.num["tag"]))
        self.it_adj = Coo(uit[:, 1], uit[:, 2], ones, (self.num["item"], self.num["tag"]))

        def max_dup(a, b, nb):
            _, cnt = np.unique(a * nb + b, return_counts=True)
            return int(cnt.max())
        # tgcn_load.py:23: the largest entry of the three (duplicate-summed) adjacencies
        self.num["weight"] = max(1, max_dup(uit[:, 0], uit[:, 2], self.num["tag"]),
                                 max_dup(uit[:, 1], uit[:, 2], self.num["tag"]))


def write_dataset(ds, data_root, dataset):
    """Inverse of the loaders: dump a `Dataset` in the reference's text formats."""
    d = os.path.join(data_root, dataset)
    os.makedirs(d, exist_ok=True)
    for split, ui in ds.user_items.items():
        with open(os.path.join(d, split + ".txt"), "w") as f:
            for u in sorted(ui):
                f.write(" ".join(str(x) for x in [u] + list(ui[u])) + "\n")
    if ds.uit_data is not None:
        np.savetxt(os.path.join(d, "user_item_tag.txt"), np.asarray(ds.uit_data), fmt="%d")
    return d
