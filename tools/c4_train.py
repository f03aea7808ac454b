#!/usr/bin/env python3
"""C4-sized TGCN, a few dozen BPR steps with and without the node tables' Adam update folded into the backward pass
(`T.Adam(...).fuse_into(model)`): the loss trajectories, step time, and how far the tables drift apart -- against the drift
of two un-fused runs, whose only difference is the order of float atomics (Adam turns a rounding-level difference of a
~1e-9 gradient into a step of +-lr).

    python tools/c4_train.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tagrec_amd as T                                   # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, k, D = 512, 25, 128
cfg = T.get_config("tgcn", dim_latent=D, dim_layer_list=[D] * 3, device=dev, train_batch=B, neighbor_k=k)
ds = T.synth.make_tripartite_device(1_000_000, 1_000_000, 2_000_000, 100_000_000, seed=2, device=dev)
epoch = T.BPR_training_data(ds, config=cfg, seed=2020).all_train_data
runs = {}
for run_id, fuse in enumerate((False, False, True)):                 # the un-fused run twice: how far float-atomic order alone drifts
    torch.manual_seed(cfg["seed"])
    model = T.TGCN(ds, config=cfg)
    opt = T.Adam(model.parameters(), lr=cfg["lr"])
    if fuse:
        opt.fuse_into(model)
    model.train()
    losses = []
    for i in range(steps):
        if i == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        lossx = model.loss(epoch[i * B:(i + 1) * B])
        opt.zero_grad()
        sum(lossx).backward()
        opt.step()
        losses.append(lossx[0].detach())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (steps - 3) * 1e3
    runs[run_id] = ([float(x) for x in losses], model.embed["user"].detach().clone())
    print(f"fused Adam {fuse}: {ms:.2f} ms/step; BPR loss " + " ".join(f"{x:.5f}" for x in runs[run_id][0][::max(1, steps // 8)]))
    del model, opt
for name, (i, j) in (("un-fused vs un-fused (atomic order only)", (0, 1)), ("un-fused vs fused", (0, 2))):
    a, b = runs[i], runs[j]
    d = (a[1] - b[1]).abs()
    print(f"{name}: largest loss difference {max(abs(x - y) for x, y in zip(a[0], b[0])):.5f}; user table max |difference| "
          f"{float(d.max()):.4f}, mean {float(d.mean()):.2e} (values up to {float(a[1].abs().max()):.3f})")
