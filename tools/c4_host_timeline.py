#!/usr/bin/env python3
"""Where the host and the GPU are at a few points of the C4 (TGCN) training step: for each mark, the host time at which the
launches before it had been QUEUED and the GPU time at which they had been EXECUTED, both from the start of the step.  A mark
whose two times are close means the GPU was waiting for the host there.

    python tools/c4_host_timeline.py [steps [pull_min_rows]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tagrec_amd as T                                   # noqa: E402
from tagrec_amd import tgcn_step as TS                   # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
B, k, D = 512, 25, 128
cfg = T.get_config("tgcn", dim_latent=D, dim_layer_list=[D] * 3, device=dev, train_batch=B, neighbor_k=k)
ds = T.synth.make_tripartite_device(1_000_000, 1_000_000, 2_000_000, 100_000_000, seed=2, device=dev)
torch.manual_seed(cfg["seed"])
model = T.TGCN(ds, config=cfg)
opt = T.Adam(model.parameters(), lr=cfg["lr"])
epoch = T.BPR_training_data(ds, config=cfg, seed=2020).all_train_data
model.train()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 2:                                    # experiment: threshold between the scatter and the pull form
    from tagrec_amd import tgcn as TG
    TG._PULL_MIN_ROWS = int(sys.argv[2])
for i in range(3 + steps):
    b = epoch[i * B:(i + 1) * B]
    if i >= 3:
        torch.cuda.synchronize()
        TS.MARKS = []
        t0 = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    lossx = model.loss(b)
    TS._mark("forward returned")
    opt.zero_grad()
    sum(lossx).backward()
    TS._mark("backward returned")
    opt.step()
    TS._mark("optimizer queued")
    if i >= 3:
        torch.cuda.synchronize()
        print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms")
        for name, th, ev in TS.MARKS:
            print(f"  {name:28s} host +{(th - t0) * 1e3:7.2f} ms   gpu +{e0.elapsed_time(ev):7.2f} ms")
        TS.MARKS = None
