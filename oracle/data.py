"""Oracle: triplet producer, neighbour tables and ranking metrics.  TEST INFRASTRUCTURE.

Restates (numpy, python loops -- small cases only):
  * `Abstract_training_data.mini_batch` (train_data/abstract.py:17-23)
  * `sample_neg_item`, `sample_neg_tail`, `split_data`, `shuffle`
    (train_data/utils.py:5-28, 31-40, 52-55)
  * `all_neighbor_sample` (data/utils.py:87-106)
  * `get_label`, `pre_rec_k`, `ndcg_k`, `minibatch` (training/utils.py:7-35, 48-54)
  * the mask -> top-k step of `epoch_test` (training/basic_test.py:30-80)
"""
import numpy as np


def mini_batch_bounds(n_rows, batch):
    """abstract.py:17-23, loop restated as written: for i in range(0, n, batch) the
    slice is [i, i+batch) unless fewer than 2*batch rows remain, in which case it
    is [i, n).  The loop does NOT stop after that merged slice, so the rows past
    the next multiple of `batch` are yielded once more as a short final slice:
    ceil(n/batch) slices in all (n=1000, batch=512 -> [0,1000) then [512,1000))."""
    out = []
    for i in range(0, n_rows, batch):
        if i + 2 * batch > n_rows:
            out.append((i, n_rows))
        else:
            out.append((i, i + batch))
    return out


def split_data(arr, k):
    """train_data/utils.py:5-16: k contiguous chunks, last takes the remainder."""
    size = len(arr) // k
    return [arr[i * size: len(arr) if i == k - 1 else (i + 1) * size] for i in range(k)]


def sample_neg_item(pos_inter, user_items, num_item, rng):
    """train_data/utils.py:19-28: one uniform negative per positive edge, rejected
    while it is one of the user's train items.  `rng` is a numpy RandomState
    (the reference uses the global one)."""
    out = np.empty((len(pos_inter), 3), dtype=np.int64)
    for k, (u, i) in enumerate(pos_inter):
        bad = user_items[int(u)]
        while True:
            j = rng.randint(0, num_item)
            if j not in bad:
                break
        out[k] = (u, i, j)
    return out


def sample_neg_tail(tri, hr_dict, num, rng):
    """train_data/utils.py:31-40: (h, r, t) -> (h, r, t, t-) with t- not in hr_dict[h][r]."""
    out = np.empty((len(tri), 4), dtype=np.int64)
    for k, (h, r, t) in enumerate(tri):
        bad = hr_dict[int(h)][int(r)]
        while True:
            j = rng.randint(0, num)
            if j not in bad:
                break
        out[k] = (h, r, t, j)
    return out


def neighbor_table(csr, max_deg, rng):
    """data/utils.py:87-106 `all_neighbor_sample`: (n_rows, max_deg) table of
    neighbour id + 1 (0 = pad) and the integer edge weight; rows shorter than
    max_deg are filled by sampling WITH replacement, a full row is a permutation."""
    n = csr.shape[0]
    ids = np.zeros((n, max_deg), dtype=np.int64)
    wts = np.zeros((n, max_deg), dtype=np.int64)
    for r in range(n):
        lo, hi = int(csr.rowptr[r]), int(csr.rowptr[r + 1])
        if hi == lo:
            continue
        pick = rng.choice(np.arange(lo, hi), max_deg, replace=(hi - lo) < max_deg)
        ids[r] = csr.col[pick].astype(np.int64) + 1
        wts[r] = csr.val[pick].astype(np.int64)
    return ids, wts


# --------------------------------------------------------------------------- metrics
def get_label(true_items, topk):
    """training/utils.py:7-12: label[i,j] = 1 iff topk[i,j] is a test item of user i."""
    return np.array([[1.0 if it in true_items[i] else 0.0 for it in row] for i, row in enumerate(topk)],
                    dtype=np.float32)


def recall_precision_hr(label, true_items, k):
    """training/utils.py:15-21 (sums over the users of the batch, not means)."""
    right = label[:, :k].sum(1)
    n_true = np.array([len(t) for t in true_items])
    return {"recall": float(np.sum(right / n_true)), "precision": float(np.sum(right) / k),
            "hr": float(np.sum(right > 0))}


def ndcg(label, true_items, k):
    """training/utils.py:24-35."""
    disc = 1.0 / np.log2(np.arange(2, k + 2))
    ideal = np.array([disc[:min(k, len(t))].sum() for t in true_items])
    ideal[ideal == 0.0] = 1.0
    got = (label[:, :k] * disc).sum(1)
    return float(np.sum(got / ideal))


def rank_metrics(rating, train_items, test_items, users, topks):
    """One `epoch_test` batch (basic_test.py:36-55 + :12-27) minus the AUC:
    mask the user's train items with -1024, take top max(topks), score.
    rating: float array [len(users), n_item]; returns dict of per-k SUMS."""
    rating = np.array(rating, dtype=np.float32, copy=True)
    for r, u in enumerate(users):
        its = train_items.get(u, [])
        if len(its):
            rating[r, np.asarray(its, dtype=np.int64)] = -(1 << 10)
    kmax = max(topks)
    # torch.topk order: descending value; ties are resolved by torch, so the
    # caller passes torch's own top-k when bit-faithful tie behaviour matters
    top = np.argsort(-rating, axis=1, kind="stable")[:, :kmax]
    truth = [test_items[u] for u in users]
    label = get_label(truth, top)
    out = {"recall": [], "precision": [], "hr": [], "ndcg": []}
    for k in topks:
        rp = recall_precision_hr(label, truth, k)
        for key, v in rp.items():
            out[key].append(v)
        out["ndcg"].append(ndcg(label, truth, k))
    return out
