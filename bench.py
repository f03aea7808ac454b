#!/usr/bin/env python3
"""Headline benchmark: BPR triplets/s of a LightGCN training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scale S]

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): LightGCN, 3 layers, dim 64, synthetic
1M users x 1M items x 50M edges (N = 2M nodes, nnz = 100M), bi_norm adjacency, BPR softplus loss,
Adam lr 0.01, train_batch 512 (the reference's default, utility/utils.py:24).  One "step" is what
`epoch_training` does per mini-batch (training/basic_train.py:14-25): full-graph propagation,
BPR loss, backward, Adam.  Everything is resident in HBM before the timed region; triplets are
sampled on the device beforehand (the reference also times its sampler separately).

With N > 1 (launched by torch.distributed.run, one rank per GPU) the SAME graph and batch are split
over the ranks (strong scaling).  Default `--parallel feature`: every rank holds D/N columns of the
node table, Adam state and activations plus the whole CSR; only row norms, row dot products and the
B triplet scores are all-reduced over RCCL.  `--parallel row`: rows are sharded and every layer
all-gathers the shard outputs (the reference's split_adj_k folds, one per GPU).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, the fused forward layer
(`spmm_rows_kernel<16, NORM_ACC>`): algorithmic bytes per launch (SURVEY.md 8d:
(8+4D) per stored entry + (8+4D) per row + 8D per row for the fused normalise/mean epilogue)
divided by its mean duration from HIP events recorded on the launch stream inside the timed region.
`cpu_baseline` times the CPU oracle (PyTorch CPU restatement of the reference path, checked against
the reference in tests/golden) on a 1/8-scale graph of the same shape and scales by stored entries.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (nodes and edges) for quick runs")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--big-batch", type=int, default=786432, help="also report triplets/s at this batch (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--force-shard", action="store_true",
                    help="run the sharded model even with one rank (exercises dist.py + RCCL init on one GPU)")
    ap.add_argument("--parallel", choices=["auto", "feature", "row"], default="auto",
                    help="multi-GPU sharding of the node table: feature = columns (default when dim %% N == 0 and "
                         "dim / N >= 8), row = row ranges with an all-gather per layer")
    ap.add_argument("--model", choices=["lightgcn", "ngcf", "tgcn", "dgcf", "disengcn"], default="lightgcn",
                    help="lightgcn = C2 (headline); ngcf = C3 (same graph, D^-1 A + I, MFMA dense layers); "
                         "tgcn = C4 (tripartite, 1M/1M/2M nodes, D=128, k=25; use --steps 3 --warmup 1); "
                         "dgcf / disengcn = the C2 graph with dynamic per-factor edge weights (4 factors, 2 routing iterations)")
    return ap.parse_args()


def spmm_bytes(nnz, n_rows, D, epilogue_row_bytes):
    return nnz * (8 + 4 * D) + n_rows * (8 + 4 * D) + n_rows * epilogue_row_bytes


def cpu_baseline(args, full_nnz):
    """CPU oracle step time on a 1/8-scale C2-shaped graph (about 15 s of CPU work), scaled to the full graph by stored
    entries."""
    import tagrec_amd as T
    from oracle import adj as oadj, models as om
    frac = 8 if args.scale >= 0.5 else 1
    nu = max(int(1_000_000 * args.scale) // frac, 1000)
    ne = max(int(50_000_000 * args.scale) // frac, 20000)
    # the GPU box gives one-GPU jobs a 16-core share of the host; more threads only oversubscribe it
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    torch.set_num_threads(cores)
    ds = T.synth.make_bipartite_device(nu, nu, ne, seed=11, device="cpu")
    e = ds.edge_index["train"]
    rp, c, v, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], nu, nu)
    A = om.csr_to_torch(oadj.CSR(rp.numpy(), c.numpy(), v.numpy(), (n, n)))
    torch.manual_seed(2020)
    tabs = [t.requires_grad_() for t in om.xavier_tables([(nu, args.dim), (nu, args.dim)], 2020)]
    opt = torch.optim.Adam(tabs, lr=0.01)
    g = torch.Generator().manual_seed(5)
    n_warm, n_timed = 1, 3
    pick = torch.randint(0, e.shape[0], ((n_warm + n_timed) * args.batch,), generator=g)
    tri = torch.stack([e[pick, 0], e[pick, 1], torch.randint(0, nu, (pick.numel(),), generator=g)], 1)
    batches = [tri[k * args.batch:(k + 1) * args.batch] for k in range(n_warm + n_timed)]
    fn = lambda b: om.lightgcn_loss(tabs, A, args.layers, b, 0.0, "softplus")
    om.adam_epoch(tabs, fn, batches[:n_warm], opt)
    t0 = time.perf_counter()
    om.adam_epoch(tabs, fn, batches[n_warm:], opt)
    dt = (time.perf_counter() - t0) / n_timed
    nnz_s = int(rp[-1])
    scaled = dt * full_nnz / nnz_s
    return {"value": args.batch / scaled, "unit": "triplets/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (torch {torch.__version__} CPU, {cores} threads), LightGCN L={args.layers} D={args.dim} "
                      f"B={args.batch} on a {nu}x{nu} graph with nnz={nnz_s}: {dt * 1e3:.1f} ms/step measured over {n_timed} steps; "
                      f"scaled by nnz ratio {full_nnz / nnz_s:.1f} to the full graph"}


def bench_tgcn(args):
    """C4: TGCN 3-layer dim 128 on a synthetic tripartite graph (1M users, 1M items, 2M tags, 100M
    assignments, k=25 neighbours per relation).  A step = one BPR mini-batch of `epoch_training` phase 0."""
    import tagrec_amd as T
    from tagrec_amd import tgcn as TG
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sc = args.scale
    nu, ni, nt, na = int(1e6 * sc), int(1e6 * sc), int(2e6 * sc), int(1e8 * sc)
    D = 128 if args.dim == 64 else args.dim
    L, B, k = args.layers, args.batch, 25
    cfg = T.get_config("tgcn", dim_latent=D, dim_layer_list=[D] * L, device=dev, train_batch=B, neighbor_k=k)
    t0 = time.perf_counter()
    ds = T.synth.make_tripartite_device(nu, ni, nt, na, seed=2, device=dev)
    torch.manual_seed(cfg["seed"])
    model = T.TGCN(ds, config=cfg)
    opt = T.Adam(model.parameters(), lr=cfg["lr"])
    prod = T.BPR_training_data(ds, config=cfg, seed=2020)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    epoch = prod.all_train_data

    def run(batches, loss_fn):
        for b in batches:
            lossx = loss_fn(b)
            opt.zero_grad()
            sum(lossx).backward()
            opt.step()
        return lossx

    model.train()
    W, K = args.warmup, args.steps
    batches = [epoch[i * B:(i + 1) * B] for i in range(W + K)]
    run(batches[:W], model.loss)
    TG.timing = {}
    torch.cuda.synchronize()
    t = time.perf_counter()
    last = run(batches[W:], model.loss)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    ms = {kk: [a.elapsed_time(b) for a, b in v] for kk, v in TG.timing.items()}
    # roofline kernel on ALL rows: the training step runs every layer only on the rows the batch's loss depends on
    # (TGCN._forward_rows), so its launches vary in size; one full forward pass gives the kernel's all-rows duration
    TG.timing = {}
    with torch.no_grad():
        model.forward()
    torch.cuda.synchronize()
    full_ms = {kk: [a.elapsed_time(b) for a, b in v] for kk, v in TG.timing.items()}
    TG.timing = None
    # transtag phase, one step, for the record
    tt = T.TransTag_training_data(ds, config=cfg, seed=1)
    tb = tt.all_train_data[:cfg["transtag_batch"]]
    run([tb], model.transtag_loss)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run([tt.all_train_data[512:1024]], model.transtag_loss)
    torch.cuda.synchronize()
    t_tt = time.perf_counter() - t
    n_pairs = sum(int(p[0].shape[0]) for p in model.nbr)          # (node, relation) pairs per layer
    A = cfg["dim_atten"]
    alg = n_pairs / 6 * k * (4 * D + 4 * A + 8) + n_pairs / 6 * (4 * D + 4 * A + 4 * k)   # mean per launch (one relation)
    fwd = full_ms.get("attn_fwd", [])
    roof = None
    if fwd:
        m = sum(fwd) / len(fwd)
        ach = alg / (m * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "tgcn_attn_fwd_kernel<32> (mean over the six relations)", "achieved": ach,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": alg, "mean_launch_ms": m, "launches_timed": len(fwd),
                "measured_on": "one full forward pass over all rows, after the timed steps",
                "all_rows_kernels_ms": {kk: sum(v) / len(v) for kk, v in full_ms.items() if kk != "attn_fwd"},
                "step_kernels_ms_per_step": {kk: sum(v) / K for kk, v in ms.items()}}
    n_nodes = nu + ni + nt
    out = {"metric": f"BPR triplets/sec, TGCN {L}-layer dim{D}, tripartite {nu}/{ni}/{nt} nodes, k={k}",
           "value": K * B / dt, "unit": "triplets/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"C4 TGCN L={L} D={D} users={nu} items={ni} tags={nt} assignments={int(ds.uit_data.shape[0])} "
                                  f"k={k} train_batch={B} adam lr=0.01 logsigmoid", "train_batch": B, "parallelism": "single"},
           "roofline": roof, "cpu_baseline": None,
           "extra": {"build_s": round(t_build, 1), "last_loss": [float(x) for x in last],
                     "transtag_step_ms": t_tt * 1e3,
                     "attention_ms_per_step": (sum(ms.get("attn_fwd", [])) + sum(ms.get("attn_bwd", []))) / K,
                     "pruned_forward": bool(model.prune_forward),
                     "fused_dense_ms_per_step": sum(sum(ms.get(kk, [])) for kk in ("fuse_fwd", "fuse_bwd", "fuse_wf")) / K,
                     "fused_fwd_all_rows_tflops": L * 2 * n_nodes * (32 * D + 48) * D / 1e12 / max(1e-9, sum(full_ms.get("fuse_fwd", [])) * 1e-3),
                     "note": "type attention + convolutions + fusion layer = fused MFMA kernels (csrc/tgcn_fuse.hip: fwd, bwd-data, "
                             "bwd-Wf); neighbour attention = csrc/tgcn.hip with pull-form backward through the SpMM kernel; "
                             "the projections P/Q/WT and the small weight gradients are plain rocBLAS GEMMs; "
                             "the CPU reference cannot materialise this size"}}
    print(json.dumps(out))


def main():
    args = parse()
    if args.model == "tgcn":
        return bench_tgcn(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched by torch.distributed.run with N ranks")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (tagrec_amd has no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import tagrec_amd as T

    sharded = world > 1 or args.force_shard
    if sharded:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
        from tagrec_amd import dist as TD

    nu = ni = max(int(1_000_000 * args.scale), 2000)
    ne = max(int(50_000_000 * args.scale), 40000)
    D, L, B = args.dim, args.layers, args.batch
    cfg = T.get_config(args.model, use_tag=False, dim_latent=D, dim_layer_list=[D] * L, device=dev, train_batch=B)
    parallel = args.parallel
    if parallel == "auto":
        parallel = "feature" if (D % world == 0 and D // world >= 8) else "row"
    Dl = D // world if (sharded and parallel == "feature") else D
    if args.model != "lightgcn" and world > 1:
        sys.exit(f"bench.py: the sharded path covers LightGCN (C2/C5); run --model {args.model} on one GPU")
    routed = args.model in ("dgcf", "disengcn")

    t0 = time.perf_counter()
    ds = T.synth.make_bipartite_device(nu, ni, ne, seed=1, device=dev)
    if args.model == "disengcn":
        ds.num["tag"] = 0                   # the reference's DisenGCN always carries a tag table; empty here
    e = ds.edge_index["train"]
    rp, col, val, n = T.graph.bipartite_norm_device(e[:, 0], e[:, 1], nu, ni, cfg["norm_type"])
    nnz = int(rp[-1])
    torch.manual_seed(cfg["seed"])
    if not sharded:
        G = T.Graph(rp, col, val, (n, n), symmetric=(cfg["norm_type"] in ("bi_norm", "plain")))
        model = {"lightgcn": T.LightGCN, "ngcf": T.NGCF, "dgcf": T.DGCF, "disengcn": T.DisenGCN}[args.model](ds, config=cfg, graph=G)
        timed_graph = G
        if not routed:
            G.transpose()                  # NGCF: build A^T once, outside the timed region
    else:
        if parallel == "feature":
            model = TD.FeatureShardedLightGCN(ds, cfg, rp, col, val, n)
        else:
            model = TD.ShardedLightGCN(ds, cfg, rp, col, val, n)
        timed_graph = model.graph
        del rp, col, val
    opt = T.Adam(model.parameters(), lr=cfg["lr"])
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0

    t0 = time.perf_counter()
    prod = T.BPR_training_data(ds, config=cfg, seed=2020)
    torch.cuda.synchronize()
    t_sample = time.perf_counter() - t0
    epoch = prod.all_train_data

    def run_steps(batches):
        for b in batches:
            lossx = model.loss(b)
            opt.zero_grad()
            sum(lossx).backward()
            opt.step()
        return lossx

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(batches):
        barrier()
        t = time.perf_counter()
        last = run_steps(batches)
        barrier()
        dt = time.perf_counter() - t
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        return dt, last

    model.train()
    W, K = args.warmup, args.steps
    batches = [epoch[k * B:(k + 1) * B] for k in range(W + K)]
    run_steps(batches[:W])
    timed_graph.timing = {}
    dt, last = timed(batches[W:])
    kernel_ms = timed_graph.timing_ms()
    timed_graph.timing = None
    loss_val = [float(x) for x in last]

    extra = {"graph_build_s": round(t_build, 2), "epoch_sampling_s": round(t_sample, 3),
             "epoch_triplets": int(epoch.shape[0]), "last_loss": loss_val,
             "edge_layers_per_s": nnz * L * 2 * K / dt}
    if args.model == "lightgcn" and not sharded and getattr(model, "restrict_forward", False):
        # the same step with every forward layer computed on ALL rows (the timed step above computes the top two
        # layers only on the rows the batch's loss depends on -- same loss and gradients)
        model.restrict_forward = False
        run_steps(batches[:W])
        dtf, _ = timed(batches[W:])
        model.restrict_forward = True
        extra["ms_per_step_all_rows_forward"] = dtf / K * 1e3
        extra["triplets_per_s_all_rows_forward"] = K * B / dtf
    if args.big_batch and epoch.shape[0] >= args.big_batch * 3 and not sharded:
        BB = args.big_batch
        bb = [epoch[k * BB:(k + 1) * BB] for k in range(3)]
        run_steps(bb[:1])
        dtb, _ = timed(bb[1:])
        extra[f"triplets_per_s_at_B{BB}"] = 2 * BB / dtb
        extra[f"ms_per_step_at_B{BB}"] = dtb / 2 * 1e3

    # roofline of the dominant kernel: fused forward layer (local rows of this rank)
    dom = "spmm_norm_acc" if args.model == "lightgcn" else "spmm"
    epi_row_bytes = 8 * D if args.model == "lightgcn" else 0
    KF = cfg.get("factor_k", 1)
    if routed:
        dom = "route_spmm"
    if sharded and parallel == "feature":
        dom, epi_row_bytes = "spmm_ss", 4                  # Y = A X on D/N columns + one float of row sum-of-squares
    fwd = kernel_ms.get(dom, [])
    n_local_rows = timed_graph.shape[0]
    local_nnz = timed_graph.nnz
    alg = spmm_bytes(local_nnz, n_local_rows, Dl, epi_row_bytes)
    if routed:
        # routed product: per stored entry col + K weights + one D-wide row; per row rowptr + the outputs it writes
        # (forward: raw + normalised + K inverse norms; backward: one); mean over the launches of a step
        # (the all-rows forward launches; the top layer's restricted launches and the row-sparse backward ones are
        # timed under their own key)
        per_row = 8 + 8 * D + 8 * KF + (4 * D if args.model == "disengcn" else 0)
        alg = local_nnz * (4 + 4 * KF + 4 * D) + n_local_rows * per_row
    roof = None
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_c2_lightgcn.json")
    if world == 1 and args.scale == 1.0 and D == 64 and args.model == "lightgcn" and os.path.exists(pmc_path):
        # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command
        # ((2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 correction; see profiles/README.md)
        with open(pmc_path) as f:
            ks = json.load(f)["kernels"]
            k = ks.get("spmm_rows_kernel<16, 1, false>") or ks.get("spmm_rows_kernel<16, 1>")
            traffic = k["traffic_bytes_per_launch"]
    if fwd:
        ms = sum(fwd) / len(fwd)
        ach = alg / (ms * 1e-3) / 1e9
        epi_name = "SS" if dom == "spmm_ss" else ("NORM_ACC" if args.model == "lightgcn" else "NONE")
        kname = f"route_spmm_kernel<{D // 4}, {KF}>" if routed else f"spmm_rows_kernel<{Dl // 4}, {epi_name}>"
        roof = {"bound": "hbm", "kernel": kname + " (+ long-row finish)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "algorithmic_bytes_per_launch": alg, "mean_launch_ms": ms, "launches_timed": len(fwd),
                "other_kernels_ms": {k: sum(v) / len(v) for k, v in kernel_ms.items() if k != dom}}
        if traffic:
            # the algorithmic figure credits every gathered row as an HBM read; rows served by L2 make the counted
            # traffic smaller, which is how `frac` can touch 1.0 -- this is the rate of the bytes that did cross the fabric
            roof["traffic_frac"] = traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if args.model == "lightgcn":
            # nominal traffic of a step that touches every row in every layer, for reference only: the
            # timed step reads less (rows the loss does not depend on / rows whose gradient is zero are not touched)
            extra["nominal_step_bytes_all_rows"] = 2 * L * spmm_bytes(nnz, n, D, 0) + L * n * 20 * D + 28 * n * D
        elif routed:
            extra["routing"] = {"factor_k": KF, "iterate_k": cfg["iterate_k"],
                                "routed_products_per_step": len(fwd) // K, "score_passes_per_step": len(kernel_ms.get("route_score", [])) // K}
        else:
            extra["dense_gflop_per_step"] = L * 3 * 2 * 2 * n * D * D / 1e9      # fwd + recompute/dA + dW, two matrices each

    if rank == 0:
        mname = {"lightgcn": "LightGCN", "ngcf": "NGCF", "dgcf": "DGCF", "disengcn": "DisenGCN"}[args.model]
        size = "1M users x 1M items x 50M edges" if args.scale == 1.0 else f"{nu} users x {ni} items x {ne} edges"
        out = {"metric": f"BPR triplets/sec, {mname} {L}-layer dim{D}, {size}",
               "value": K * B / dt, "unit": "triplets/s", "n_gpus": world, "steps": K, "warmup": W,
               "ms_per_step": dt / K * 1e3, "higher_is_better": True,
               "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{ {'lightgcn': 'C2', 'ngcf': 'C3'}.get(args.model, 'C2-graph') } {mname} L={L} D={D} users={nu} items={ni} "
                                      f"edges={ne} nnz={nnz} train_batch={B} adam lr=0.01 {cfg['norm_type']} {cfg['mul_loss_func']}",
                          "train_batch": B, "parallelism": f"{parallel}-shard x{world}" if sharded else "single",
                          "step": "loss -> backward -> Adam on one batch; same loss and gradients as the all-rows step: rows "
                                  "of the top two forward layers that the batch's loss does not read are not computed, "
                                  "backward products do not fetch operand rows that are exactly zero "
                                  "(extra.ms_per_step_all_rows_forward = every forward layer on all rows)"},
               "roofline": roof, "extra": extra}
        if not args.no_cpu and world == 1 and args.model == "lightgcn":
            out["cpu_baseline"] = cpu_baseline(args, nnz)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
