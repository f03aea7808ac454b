#!/usr/bin/env python3
"""Time the kernels of the TGCN attention backward (pull form) on one C4-sized relation:
n source rows (compact), n_dst destination rows, k neighbours, D = 128, A = 32.
    python tools/attn_bwd_microbench.py [n] [n_dst]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tagrec_amd as T  # noqa: E402,F401
from tagrec_amd import tgcn as TG, tgcn_step as TS  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 380_000
n_dst = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
k, D, A, n_wt = 25, 128, 32, 17
g = torch.Generator(device=dev).manual_seed(1)
# destinations drawn with a popularity skew (a few very popular ones), ids + 1, a few pads
pop = (torch.rand(n, k, device=dev, generator=g) ** 3 * n_dst).long().clamp_(max=n_dst - 1)
idx = (pop + 1).to(torch.int32)
idx[torch.rand(n, k, device=dev, generator=g) < 0.02] = 0
widx = (1 + (torch.rand(n, k, device=dev, generator=g) ** 6 * (n_wt - 1)).long().clamp_(max=n_wt - 2)).to(torch.int32)   # mostly 1
rnd = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.3
P, Q, WT, v, Ej, d_out = rnd(n, A), rnd(n_dst, A), rnd(n_wt, A), rnd(A), rnd(n_dst, D), rnd(n, D)
_, attn = TS.attn_fwd(P, Q, WT, v, Ej, idx, widx)
for rep in range(3):
    TG.timing = {}
    TS.attn_bwd_pulls(P, Q, WT, v, Ej, idx, widx, attn, d_out, None, None, w_major=1)
    torch.cuda.synchronize()
    ms = {kk: sum(a.elapsed_time(b) for a, b in vv) for kk, vv in TG.timing.items()}
TG.timing = {}
dh = torch.empty(n * k, A, device=dev)
TS.attn_bwd(P, Q, WT, v, Ej, idx, widx, attn, d_out, None, None, dh)
TS._pull_add(idx, attn, d_out, dh, n_dst, None, None)
torch.cuda.synchronize()
old = {kk: sum(a.elapsed_time(b) for a, b in vv) for kk, vv in TG.timing.items()}
TG.timing = None
pairs = n * k
print(f"pairs {pairs / 1e6:.1f} M; new form: " + ", ".join(f"{kk} {vv:.3f} ms" for kk, vv in ms.items()))
print("old form: " + ", ".join(f"{kk} {vv:.3f} ms" for kk, vv in old.items()) + " (+ its two SpMM pulls, untimed here)")
