"""Step time at the reference's own dataset scale (C1: ml-100k-sized), where the step is launch-bound.
Usage: python tools/small_scale_step.py   (needs a GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tagrec_amd as T

dev = torch.device("cuda:0")
ds = T.synth.make_cf_dataset(943, 1682, 100000, seed=0, n_tag=400, n_assign=30000)
for name, kw in (("lightgcn", dict(use_tag=False, dim_layer_list=[64, 64])),
                 ("ngcf", dict(use_tag=False, dim_layer_list=[64, 64])),
                 ("tgcn", dict(dim_layer_list=[64, 64], neighbor_k=25))):
    cfg = T.get_config(name, dim_latent=64, device=dev, train_batch=512, **kw)
    for graphs in (None, {}):
        torch.manual_seed(0)
        model = {"lightgcn": T.LightGCN, "ngcf": T.NGCF, "tgcn": T.TGCN}[name](ds, config=cfg)
        opt = T.Adam(model.parameters(), lr=0.01, capturable=graphs is not None)
        prod = T.BPR_training_data(ds, config=cfg, seed=1)
        model.train()
        T.epoch_training(prod, model.loss, opt, verbose=False, graphs=graphs)      # warm-up epoch (captures the graph)
        torch.cuda.synchronize()
        t = time.perf_counter()
        losses = T.epoch_training(prod, model.loss, opt, verbose=False, graphs=graphs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        mode = "eager" if graphs is None else ("HIP graph" if not graphs.get("errors") else f"graph FAILED {graphs['errors'][:1]}")
        print(f"{name} [{mode}]: {len(losses)} steps/epoch, {dt / len(losses) * 1e3:.3f} ms/step, epoch {dt:.3f} s, "
              f"loss {losses[-1]:.4f}", flush=True)
