"""Synthetic datasets in the shapes the reference's loaders produce.

`Dataset` carries exactly the attributes the reference's models, producers and
test harness read from `TGCN_load` (/root/reference/data/tgcn_load.py:12-25,
data/cf_load.py:9-28): `.num`, `.ui_adj/.ut_adj/.it_adj` (COO: `.row .col .data
.shape`; a scipy COO matrix is accepted anywhere a `Coo` is), `.user_items`,
`.edge_index`, `.uit_data`.

Host generator (numpy) for C1-scale graphs; device generator (torch) for the
million-node configs of SURVEY.md section 8(d), which never leave the GPU.
"""
from collections import namedtuple

import numpy as np

Coo = namedtuple("Coo", ["row", "col", "data", "shape"])


class Dataset:
    def __init__(self):
        self.num = {}
        self.ui_adj = self.ut_adj = self.it_adj = None
        self.user_items = {}
        self.edge_index = {}
        self.uit_data = None
        self.rel = None           # device CSR relations of the tripartite generator

    def create_edge(self):
        """`TGCN_load.create_edge` (data/tgcn_load.py:55-71): dict relation -> [2, E] arrays (head row, tail row) for
        ui, iu, ut, tu, it, ti with node ids offset into [users | items | tags].  The reference's COO matrices hold
        one entry per (u, i, t) assignment; a block stored here with summed counts is expanded back."""
        def entries(c, row_off, col_off):
            cnt = np.rint(np.asarray(c.data)).astype(np.int64)
            return (np.repeat(np.asarray(c.row).astype(np.int64), cnt) + row_off,
                    np.repeat(np.asarray(c.col).astype(np.int64), cnt) + col_off)
        nu, ni = self.num["user"], self.num["item"]
        out = {}
        for k, (c, ro, co) in enumerate(((self.ui_adj, 0, nu), (self.ut_adj, 0, nu + ni), (self.it_adj, nu, nu + ni))):
            r, cc = entries(c, ro, co)
            out[2 * k], out[2 * k + 1] = np.stack([r, cc]), np.stack([cc, r])
        return out


def _zipf_weights(n, alpha):
    w = np.arange(1, n + 1, dtype=np.float64) ** (-alpha)
    return w / w.sum()


def make_cf_dataset(n_user=943, n_item=1682, n_edge=100_000, item_alpha=0.8, seed=0,
                    n_tag=0, n_assign=0, tag_alpha=1.0, train_ratio=0.8, max_weight=16):
    """C1-style user-item(-tag) dataset: items Zipf(item_alpha),
    users uniform, de-duplicated, every user and item has degree >= 1; per-user
    80/20 split with the `int(len * ratio)` rule of data/preprocess/help.py:79-112
    (a user with one item keeps it for test, then every user/item missing from
    train is patched back in -- the reference's `change_dict` intent)."""
    rng = np.random.RandomState(seed)
    pi = _zipf_weights(n_item, item_alpha)
    perm = rng.permutation(n_item)                 # item ids are not sorted by popularity
    # degree >= 1 for everyone, then top up with fresh draws until n_edge distinct pairs exist
    u = np.concatenate([np.arange(n_user), rng.randint(0, n_user, size=n_item)])
    i = np.concatenate([perm[rng.choice(n_item, size=n_user, p=pi)], np.arange(n_item)])
    key = np.unique(u.astype(np.int64) * n_item + i)
    base = key
    while len(key) < n_edge:
        m = int((n_edge - len(key)) * 1.3) + 16
        uu = rng.randint(0, n_user, size=m).astype(np.int64)
        ii = perm[rng.choice(n_item, size=m, p=pi)]
        key = np.unique(np.concatenate([key, uu * n_item + ii]))
    if len(key) > n_edge:                          # trim extras, never the degree-guarantee pairs
        extra = np.setdiff1d(key, base)
        drop = rng.choice(extra, size=len(key) - n_edge, replace=False)
        key = np.setdiff1d(key, drop)
    u, i = key // n_item, key % n_item

    ds = Dataset()
    train_mask = np.zeros(len(u), dtype=bool)
    bounds = np.flatnonzero(np.diff(u)) + 1
    for seg in np.split(np.arange(len(u)), bounds):
        k = int(len(seg) * train_ratio)
        if k > 0:
            train_mask[rng.choice(seg, size=k, replace=False)] = True
    # every user and every item must appear in train
    seen_u = np.zeros(n_user, bool); seen_u[u[train_mask]] = True
    seen_i = np.zeros(n_item, bool); seen_i[i[train_mask]] = True
    for e in range(len(u)):
        if not train_mask[e] and (not seen_u[u[e]] or not seen_i[i[e]]):
            train_mask[e] = True
            seen_u[u[e]] = True
            seen_i[i[e]] = True
    tr = np.stack([u[train_mask], i[train_mask]], axis=1)
    te = np.stack([u[~train_mask], i[~train_mask]], axis=1)
    ds.edge_index = {"train": tr, "test": te}
    for name, arr in ds.edge_index.items():
        d = {}
        for a, b in arr:
            d.setdefault(int(a), []).append(int(b))
        ds.user_items[name] = d
    ds.num = {"user": n_user, "item": n_item}
    ds.ui_adj = Coo(tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64),
                    np.ones(len(tr), np.float32), (n_user, n_item))
    if n_tag:
        pt = _zipf_weights(n_tag, tag_alpha)
        tperm = rng.permutation(n_tag)
        e = rng.randint(0, len(tr), size=n_assign)            # tags annotate train interactions
        t = tperm[rng.choice(n_tag, size=n_assign, p=pt)]
        uit = np.stack([tr[e, 0], tr[e, 1], t], axis=1)
        base = tr[np.arange(n_tag) % len(tr)]                 # every tag id is used at least once
        uit = np.concatenate([uit, np.stack([base[:, 0], base[:, 1], np.arange(n_tag)], axis=1)])
        uit = np.unique(uit, axis=0)                          # read_knowledge_data uniq (data/utils.py:9-20)
        ds.uit_data = uit.astype(np.int32)
        ds.num["tag"] = n_tag
        # duplicates are summed when the COO is densified -> integer co-occurrence weights
        ds.ut_adj = Coo(uit[:, 0].astype(np.int64), uit[:, 2].astype(np.int64),
                        np.ones(len(uit), np.float32), (n_user, n_tag))
        ds.it_adj = Coo(uit[:, 1].astype(np.int64), uit[:, 2].astype(np.int64),
                        np.ones(len(uit), np.float32), (n_item, n_tag))
        ds.num["weight"] = int(max(1, _max_dup(ds.ut_adj), _max_dup(ds.it_adj)))
    return ds


def _max_dup(coo):
    key = coo.row.astype(np.int64) * coo.shape[1] + coo.col
    _, cnt = np.unique(key, return_counts=True)
    return int(cnt.max()) if cnt.size else 0


def sample_bpr_epoch(ds, seed, shuffle=True):
    """One epoch of BPR triplets, [E,3] int64: one uniform negative per train
    edge, rejected while it is a train item of the user
    (train_data/utils.py:19-28), then a global shuffle (utils.py:52-55).
    Vectorised rejection sampling on the host; deterministic in `seed`
    (numpy seed 2020 + epoch)."""
    rng = np.random.RandomState(seed)
    tr = ds.edge_index["train"]
    n_item = ds.num["item"]
    pos_key = np.sort(tr[:, 0].astype(np.int64) * n_item + tr[:, 1])
    neg = rng.randint(0, n_item, size=len(tr))
    while True:
        k = tr[:, 0].astype(np.int64) * n_item + neg
        j = np.searchsorted(pos_key, k)
        j[j >= len(pos_key)] = len(pos_key) - 1
        bad = pos_key[j] == k
        if not bad.any():
            break
        neg[bad] = rng.randint(0, n_item, size=int(bad.sum()))
    out = np.stack([tr[:, 0], tr[:, 1], neg], axis=1).astype(np.int64)
    if shuffle:
        out = out[rng.permutation(len(out))]
    return out


# ---------------------------------------------------------------------------------- device generator
def make_bipartite_device(n_user, n_item, n_edge, seed, device, item_alpha=0.8, item_cap=100_000,
                          user_mu=3.5, user_sigma=0.8):
    """C2/C5-style user-item graph generated ON the GPU: item popularity
    Zipf(item_alpha) capped at `item_cap` interactions, user activity log-normal(mu, sigma),
    both rescaled to `n_edge` DISTINCT pairs, every user and item has degree >= 1, ids randomly
    permuted (so a contiguous row range is a statistically balanced shard).
    Returns a `Dataset` whose `edge_index['train']` is an int64 [E,2] tensor on `device`
    (sorted by (user, item)); `user_items` is left empty -- at this size it never visits the host."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)

    wi = torch.arange(1, n_item + 1, device=dev, dtype=torch.float64).pow(-item_alpha)
    wi = wi / wi.sum()
    wi = torch.clamp(wi, max=float(item_cap) / float(n_edge))
    wu = torch.exp(user_mu + user_sigma * torch.randn(n_user, device=dev, dtype=torch.float64, generator=gen))
    cdf_i = torch.cumsum(wi / wi.sum(), 0)
    cdf_u = torch.cumsum(wu / wu.sum(), 0)
    perm_i = torch.randperm(n_item, device=dev, generator=gen)

    def draw(m):
        u = torch.searchsorted(cdf_u, torch.rand(m, device=dev, dtype=torch.float64, generator=gen)).clamp_(max=n_user - 1)
        i = torch.searchsorted(cdf_i, torch.rand(m, device=dev, dtype=torch.float64, generator=gen)).clamp_(max=n_item - 1)
        return u, perm_i[i]

    # degree >= 1 for everyone
    u0, i0 = draw(n_user + n_item)
    u0[:n_user] = torch.arange(n_user, device=dev)
    i0[n_user:] = torch.arange(n_item, device=dev)
    key = torch.unique(u0 * n_item + i0)
    base = key
    while key.numel() < n_edge:
        m = int((n_edge - key.numel()) * 1.15) + 1024
        u, i = draw(m)
        key = torch.unique(torch.cat([key, u * n_item + i]))
    if key.numel() > n_edge:
        is_base = torch.isin(key, base)
        extra = torch.nonzero(~is_base).flatten()
        drop = extra[torch.randperm(extra.numel(), device=dev, generator=gen)[:key.numel() - n_edge]]
        keep = torch.ones(key.numel(), dtype=torch.bool, device=dev)
        keep[drop] = False
        key = key[keep]
    ds = Dataset()
    ds.num = {"user": int(n_user), "item": int(n_item)}
    u = torch.div(key, n_item, rounding_mode="floor")
    ds.edge_index = {"train": torch.stack([u, key - u * n_item], dim=1).contiguous()}
    return ds


def make_tripartite_device(n_user, n_item, n_tag, n_assign, seed, device, item_alpha=0.8, tag_alpha=1.0,
                           user_mu=3.5, user_sigma=0.8, max_weight=16):
    """C4-style (user, item, tag) assignments generated on the GPU: `n_assign` draws with
    item popularity Zipf(item_alpha), tag popularity Zipf(tag_alpha), user activity log-normal; every user,
    item and tag appears at least once.  Duplicate triples are removed (data/utils.py:9-20), repeated
    (user, tag) / (item, tag) pairs become integer weights clipped to `max_weight` (num['weight']).
    Returns a `Dataset` with `uit_data` [T,3] and `edge_index['train']` [E,2] (distinct (u,i)) as int64 device
    tensors and `rel`: the six relations ui, ut, iu, it, tu, ti as device CSR (rowptr, col, int weight)."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)

    def cdf(w):
        return torch.cumsum(w / w.sum(), 0)

    ci = cdf(torch.arange(1, n_item + 1, device=dev, dtype=torch.float64).pow(-item_alpha))
    ct = cdf(torch.arange(1, n_tag + 1, device=dev, dtype=torch.float64).pow(-tag_alpha))
    cu = cdf(torch.exp(user_mu + user_sigma * torch.randn(n_user, device=dev, dtype=torch.float64, generator=gen)))
    pi = torch.randperm(n_item, device=dev, generator=gen)
    pt = torch.randperm(n_tag, device=dev, generator=gen)

    def draw(c, n, m):
        return torch.searchsorted(c, torch.rand(m, device=dev, dtype=torch.float64, generator=gen)).clamp_(max=n - 1)

    m = n_assign
    u, i, t = draw(cu, n_user, m), pi[draw(ci, n_item, m)], pt[draw(ct, n_tag, m)]
    # everyone appears at least once
    k = max(n_user, n_item, n_tag)
    ar = torch.arange(k, device=dev)
    u = torch.cat([u, ar % n_user]); i = torch.cat([i, ar % n_item]); t = torch.cat([t, ar % n_tag])
    key = torch.unique((u * n_item + i) * n_tag + t)
    t = key % n_tag
    ui = torch.div(key, n_tag, rounding_mode="floor")
    u, i = torch.div(ui, n_item, rounding_mode="floor"), ui % n_item
    ds = Dataset()
    ds.uit_data = torch.stack([u, i, t], dim=1).contiguous()
    uik = torch.unique(ui)
    eu = torch.div(uik, n_item, rounding_mode="floor")
    ds.edge_index = {"train": torch.stack([eu, uik - eu * n_item], dim=1).contiguous()}

    def rel(a, b, na, nb, unit=False):
        k2, cnt = torch.unique(a * nb + b, return_counts=True)
        r = torch.div(k2, nb, rounding_mode="floor")
        ptr = torch.zeros(na + 1, dtype=torch.int64, device=dev)
        torch.cumsum(torch.bincount(r, minlength=na), 0, out=ptr[1:])
        w = torch.ones_like(cnt) if unit else cnt.clamp(max=max_weight)
        return ptr, (k2 - r * nb).to(torch.int32), w.to(torch.int32)

    ds.rel = {"ui": rel(u, i, n_user, n_item, unit=True), "ut": rel(u, t, n_user, n_tag),
              "iu": rel(i, u, n_item, n_user, unit=True), "it": rel(i, t, n_item, n_tag),
              "tu": rel(t, u, n_tag, n_user), "ti": rel(t, i, n_tag, n_item)}
    wmax = max(int(ds.rel[r][2].max()) for r in ("ut", "it"))
    ds.num = {"user": int(n_user), "item": int(n_item), "tag": int(n_tag), "weight": int(wmax)}
    return ds
