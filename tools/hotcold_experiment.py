"""Would splitting the product into a HOT-column pass and a COLD-column pass raise the L2 hit rate?  

    python tools/hotcold_experiment.py

On the C2 graph (D = 64) the all-rows product is timed as it is, and as two products over a column split of the same
matrix -- entries whose column is one of the H highest-degree nodes of its type (their rows = H x 256 B per type) and the
rest -- for several H.  If the hot pass ran at L2 speed the two passes together would be faster than the one pass."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import tagrec_amd as T  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def split(rp, col, val, keep, n):
    """CSR of the entries with keep[entry] (same row order, entry order kept)."""
    deg = rp[1:] - rp[:-1]
    rows = torch.repeat_interleave(torch.arange(n, device=rp.device), deg)
    r = rows[keep]
    nrp = torch.zeros(n + 1, dtype=torch.int64, device=rp.device)
    torch.cumsum(torch.bincount(r, minlength=n), 0, out=nrp[1:])
    return nrp, col[keep].contiguous(), val[keep].contiguous()


def main():
    dev = torch.device("cuda:0")
    nu = ni = 1_000_000
    ds = T.synth.make_bipartite_device(nu, ni, 50_000_000, seed=1, device=dev)
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], nu, ni)
    deg = rp[1:] - rp[:-1]
    D = 64
    x = torch.randn(n, D, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    g = T.Graph(rp, col, val, (n, n), symmetric=True)
    y = torch.empty_like(x)
    t_all = timed(lambda: g.spmm(x, out=y))
    print(f"one pass: {t_all:.3f} ms  (nnz {int(rp[-1])})")
    ref = y.clone()
    for H in (4096, 16384, 65536, 262144):
        hot = torch.zeros(n, dtype=torch.bool, device=dev)
        hot[torch.topk(deg[:nu], H).indices] = True
        hot[nu + torch.topk(deg[nu:], H).indices] = True
        keep = hot[col.long()]
        frac = float(keep.float().mean())
        gh = T.Graph(*split(rp, col, val, keep, n), (n, n))
        gc = T.Graph(*split(rp, col, val, ~keep, n), (n, n))
        yh, yc = torch.empty_like(x), torch.empty_like(x)
        th = timed(lambda: gh.spmm(x, out=yh))
        tc = timed(lambda: gc.spmm_axpy(x, yh, 1.0, yc))           # cold pass adds the hot pass's result
        err = float((yc - ref).abs().max())
        print(f"H = {H:7d} per type: hot entries {frac:.3f}  hot pass {th:.3f} ms  cold pass (+ add) {tc:.3f} ms  "
              f"sum {th + tc:.3f} ms  vs {t_all:.3f}   max |diff| {err:.2e}")


if __name__ == "__main__":
    main()
