"""Train LightGCN on the C2-shaped synthetic graph for a few hundred steps and evaluate Recall@20 / NDCG@20 with the
fused scoring kernel: an end-to-end run at the headline size (sampler -> steps -> full-propagation forward -> top-K).
A random 1 % of the interactions is held out as the test set.  Usage: python tools/c2_train_eval.py [steps] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tagrec_amd as T

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
nu = ni = 1_000_000
t0 = time.perf_counter()
ds = T.synth.make_bipartite_device(nu, ni, 50_000_000, seed=1, device=dev)
e = ds.edge_index["train"]
g = torch.Generator(device=dev).manual_seed(7)
hold = torch.rand(e.shape[0], device=dev, generator=g) < 0.01
test, train = e[hold], e[~hold]
ds.edge_index = {"train": train, "test": test}
ds.user_items = {"train": train, "test": test}            # Basic_test takes [E, 2] arrays as well as dicts
cfg = T.get_config("lightgcn", use_tag=False, dim_latent=64, dim_layer_list=[64] * 3, device=dev, train_batch=B, topks=[10, 20])
rp, col, val, n = T.graph.bipartite_norm_device(train[:, 0], train[:, 1], nu, ni, "bi_norm")
G = T.Graph(rp, col, val, (n, n), symmetric=True)
torch.manual_seed(2020)
model = T.LightGCN(ds, config=cfg, graph=G)
opt = T.Adam(model.parameters(), lr=0.01).fuse_into(model)      # the table update runs inside the last backward kernel
prod = T.BPR_training_data(ds, config=cfg, seed=2020)
tester = T.Basic_test(ds, config=cfg, with_auc=False)
users = torch.unique(test[:, 0])[:100_000]
torch.cuda.synchronize()
print(f"setup {time.perf_counter() - t0:.1f} s: train edges {train.shape[0]}, test edges {test.shape[0]}", flush=True)


def evaluate(tag):
    t = time.perf_counter()
    r = tester.run(model, istest=True, all_users=users)
    torch.cuda.synchronize()
    print(f"[{tag}] recall@20 {r['recall'][1]:.5f} ndcg@20 {r['ndcg'][1]:.5f} hr@20 {r['hr'][1]:.4f} "
          f"({users.numel()} users x {ni} items in {time.perf_counter() - t:.2f} s)", flush=True)
    return r


evaluate("init")
model.train()
ep = prod.all_train_data
t = time.perf_counter()
parts = []
for s in range(steps):
    lo = (s * B) % (ep.shape[0] - B)
    lossx = model.loss(ep[lo:lo + B])
    parts.append(lossx[0].detach())
    opt.zero_grad()
    sum(lossx).backward()
    opt.step()
torch.cuda.synchronize()
dt = time.perf_counter() - t
ls = torch.stack(parts).cpu().numpy()
print(f"{steps} steps of B={B}: {dt / steps * 1e3:.2f} ms/step, {steps * B / dt / 1e6:.2f} M triplets/s; "
      f"loss {ls[:5].mean():.4f} -> {ls[-5:].mean():.4f}", flush=True)
r = evaluate("trained")
assert ls[-5:].mean() < ls[:5].mean()
