"""Does relabelling the C2 graph change the fused forward layer?  (VERDICT r01 item 6 / DESIGN section 5)

    python tools/reorder_experiment.py [scale]

Times `spmm_norm_acc` (all rows, D = 64) on the C2 graph under four node orders and prints one line each:
  random      : the generator's order (ids randomly permuted: what bench.py runs)
  degree      : users and items each sorted by descending degree (popular rows adjacent in memory)
  degree-rows : rows processed in descending-degree order, columns unchanged (launch order only)
  bfs-like    : items by descending degree, each user placed by its most popular item (users that share a hot item adjacent)
Results are equal up to summation order (the products are compared after mapping back).  L2 hit rates come from a
separate `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` pass of this script (tools/profile_reorder.sh)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import tagrec_amd as T  # noqa: E402


def permuted(rp, col, val, perm):
    """CSR of P A P^T where new id = perm[old id] (perm: old -> new)."""
    n = rp.numel() - 1
    deg = rp[1:] - rp[:-1]
    rows = torch.repeat_interleave(torch.arange(n, device=rp.device), deg)
    return T.graph.coalesce_device(perm[rows], perm[col.long()], val, n, n)


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    dev = torch.device("cuda:0")
    nu = ni = int(1_000_000 * scale)
    ds = T.synth.make_bipartite_device(nu, ni, int(50_000_000 * scale), seed=1, device=dev)
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], nu, ni)
    deg = rp[1:] - rp[:-1]
    D = 64
    x = torch.randn(n, D, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    ident = torch.arange(n, device=dev)

    def rank_desc(key):                       # old id -> position when sorted by descending key (stable)
        order = torch.argsort(key, descending=True, stable=True)
        r = torch.empty_like(order)
        r[order] = torch.arange(order.numel(), device=dev)
        return r

    perms = {"random": ident}
    pu, pi = rank_desc(deg[:nu]), rank_desc(deg[nu:]) + nu
    perms["degree"] = torch.cat([pu, pi])
    # users keyed by the rank of their most popular item (then by degree): co-accessed hot items -> adjacent users
    best = torch.full((nu,), n, dtype=torch.int64, device=dev)
    best.scatter_reduce_(0, e[:, 0], pi[e[:, 1]], reduce="amin")
    perms["bfs-like"] = torch.cat([rank_desc(-(best * 4096 - deg[:nu].clamp(max=4095))), pi])
    ref = None
    for name, perm in perms.items():
        if name == "random":
            g = T.Graph(rp, col, val, (n, n), symmetric=True)
            xp = x
        else:
            prp, pcol, pval = permuted(rp, col, val, perm)
            g = T.Graph(prp, pcol, pval, (n, n), symmetric=True)
            xp = torch.empty_like(x)
            xp[perm] = x
        y, inv, acc = torch.empty_like(x), torch.empty(n, device=dev), torch.zeros_like(x)
        for _ in range(2):
            g.spmm_norm_acc(xp, y, inv, acc, 0.25)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            g.spmm_norm_acc(xp, y, inv, acc, 0.25)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        back = y[perm] if name != "random" else y
        if ref is None:
            ref = back.clone()
        err = float((back - ref).abs().max() / ref.abs().max())
        print(f"{name:12s} fused forward layer {ms:7.3f} ms   max rel diff vs random order {err:.2e}", flush=True)
        del g


if __name__ == "__main__":
    main()
