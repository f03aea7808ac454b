"""python tools/c4_rows.py [steps]: the C4 TGCN step taken apart -- how many rows of every type each layer computes for one
batch (`TGCN._needed_rows`), the rows whose gradient is non-zero per backward call, and the per-call time of the HIP
kernels in launch order."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tagrec_amd as T
from tagrec_amd import tgcn as TG

dev = torch.device("cuda", 0)
D, L, B, k = 128, 3, 512, 25
cfg = T.get_config("tgcn", dim_latent=D, dim_layer_list=[D] * L, device=dev, train_batch=B, neighbor_k=k)
ds = T.synth.make_tripartite_device(1_000_000, 1_000_000, 2_000_000, 100_000_000, seed=2, device=dev)
torch.manual_seed(cfg["seed"])
model = T.TGCN(ds, config=cfg)
opt = T.Adam(model.parameters(), lr=cfg["lr"])
prod = T.BPR_training_data(ds, config=cfg, seed=2020)
epoch = prod.all_train_data
model.train()
need = model._needed_rows(epoch[:B].to(dev))
for l in range(1, L + 1):
    print("layer", l, {t: ("all" if r is None else int(r.numel())) for t, r in need[l].items()})
calls = []
orig = TG._timed
def rec(name, fn, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rc = fn(*args); e1.record()
    n = next((a for a in args if isinstance(a, int) and a > 64), None)
    calls.append((name, n, e0, e1))
    return rc
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for i in range(steps):
    if i == steps - 1:
        TG._timed = rec
        torch.cuda.synchronize(); t = time.perf_counter()
    lossx = model.loss(epoch[i * B:(i + 1) * B])
    opt.zero_grad(); sum(lossx).backward(); opt.step()
torch.cuda.synchronize()
print("step ms", (time.perf_counter() - t) * 1e3)
tot = {}
for name, n, a, b in calls:
    ms = a.elapsed_time(b)
    tot[name] = tot.get(name, 0) + ms
    print(f"{name:10s} n={n!s:>9s} {ms:8.3f} ms")
print(tot)
