"""The C5 SHAPE (LightGCN dim 256, 10 M x 10 M users x items, 500 M edges: nnz = 1 G, table 20.5 GB) on ONE MI355X.

    python tools/c5_one_gpu.py [steps]

C5 proper is the 8-GPU row-sharded configuration; this shows that the whole problem is resident in one GPU's 288 GB
(table + Adam state 61 GB, CSR 12 GB, activations and gradient tables of the restricted step) and what a step costs
there, with the allocator's view of memory per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tagrec_amd as T  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda:0")
    nu = ni = 10_000_000
    t0 = time.time()
    ds = T.synth.make_bipartite_device(nu, ni, 500_000_000, seed=1, device=dev)
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], nu, ni)
    G = T.Graph(rp, col, val, (n, n), symmetric=True)
    cfg = T.get_config("lightgcn", use_tag=False, dim_latent=256, dim_layer_list=[256] * 3, device=dev, train_batch=512)
    torch.manual_seed(2020)
    model = T.LightGCN(ds, config=cfg, graph=G)
    opt = T.Adam(model.parameters(), lr=0.01)
    gen = torch.Generator(device=dev).manual_seed(1)
    gb = lambda x: x / 1e9
    print(f"build {time.time() - t0:.1f} s, nnz {G.nnz}, allocated {gb(torch.cuda.memory_allocated()):.0f} GB, "
          f"reserved {gb(torch.cuda.memory_reserved()):.0f} GB", flush=True)
    model.train()
    for step in range(steps):
        pick = torch.randint(0, e.shape[0], (512,), device=dev, generator=gen)
        b = torch.stack([e[pick, 0], e[pick, 1], torch.randint(0, ni, (512,), device=dev, generator=gen)], 1)
        torch.cuda.synchronize()
        t = time.time()
        lossx = model.loss(b)
        torch.cuda.synchronize()
        t1 = time.time()
        opt.zero_grad()
        sum(lossx).backward()
        torch.cuda.synchronize()
        t2 = time.time()
        opt.step()
        torch.cuda.synchronize()
        t3 = time.time()
        loss = float(lossx[0])
        del lossx
        st = torch.cuda.memory_stats()
        print(f"step {step}: loss {loss:.5f}  forward {1e3 * (t1 - t):.0f} ms, backward {1e3 * (t2 - t1):.0f} ms, adam "
              f"{1e3 * (t3 - t2):.0f} ms | allocated {gb(torch.cuda.memory_allocated()):.0f} GB, peak "
              f"{gb(torch.cuda.max_memory_allocated()):.0f} GB, reserved {gb(torch.cuda.memory_reserved()):.0f} GB, "
              f"allocator retries {st.get('num_alloc_retries', 0)}, segments freed {st.get('segment.all.freed', 0)}", flush=True)


if __name__ == "__main__":
    main()
