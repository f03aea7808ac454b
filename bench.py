#!/usr/bin/env python3
"""Headline benchmark: BPR triplets/s of a LightGCN training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scale S]

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): LightGCN, 3 layers, dim 64, synthetic
1M users x 1M items x 50M edges (N = 2M nodes, nnz = 100M), bi_norm adjacency, BPR softplus loss,
Adam lr 0.01, train_batch 512 (the reference's default, utility/utils.py:24).  One "step" is what
`epoch_training` does per mini-batch (training/basic_train.py:14-25): full-graph propagation,
BPR loss, backward, Adam.  Everything is resident in HBM before the timed region; triplets are
sampled on the device beforehand (the reference also times its sampler separately).

With N > 1 the SAME graph and batch are split over N ranks, one per GPU (strong scaling).  Launch either
as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` or as plain
`python bench.py --gpus N`: the latter starts the N ranks itself (fresh child processes; the parent
never touches the GPU).  `--parallel row` (default from 8 ranks at dim 64): the node table, Adam state and CSR
rows are sharded by row range -- the reference's split_adj_k folds (adj.py:114-140,158-164), one per GPU
-- and the step is the restricted one of the single-GPU model: block-wise pipelined all-gathers of the
layers that need every row, the top layer in push form on the batch rows, flagged gradient tables
(tagrec_amd/dist.py).  `--parallel feature` (default up to 4 ranks, and whenever dim / ranks >= 32 as at C5: a column slice then still
gathers whole 128-byte rows): every rank holds D/N columns of every row plus the whole CSR; only row
norms, row dot products and the B triplet scores are all-reduced.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, the fused forward layer on all rows
(`spmm_rows_kernel<16, NORM_ACC>`, run without the layer-mean accumulator: the training step forms the mean on the
batch rows): algorithmic bytes per launch ((8+4D) per stored entry + (8+4D) per row + 4 per row for the
norm) divided by its mean duration from HIP events recorded on the launch stream inside the timed region.
`roofline.measured_ceilings` are probe kernels of the library run before the timed region (stream triad / copy / read,
random 256-byte-row gathers from 256 MB / 512 MB / 4 GB tables), `frac_of_gather_ceiling` prices the kernel against them.
`cpu_baseline` times the CPU oracle (PyTorch CPU restatement of the reference path, checked against
the reference in tests/golden) on the same graph and batch size, on the box's host cores: 1 warm-up + 3 timed steps.
At N = 1 the headline run also carries `extra.configs.C3` (NGCF, same graph) and `.C4` (TGCN, 1 M / 1 M / 2 M nodes; with
the shader clock measured under its fused kernel) as short legs in the same process, and `extra.hip_graph_replay` (the same
C2 step as one captured HIP graph).  `--model ngcf|tgcn|dgcf|...` run those models on their own.
NOTE: the multi-rank path has been verified over gloo and in an RCCL group of one; it has never run over xGMI.
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
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: exact-fp32 MFMA = the fp32 vector rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (nodes and edges) for quick runs")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--big-batch", type=int, default=786432, help="also report triplets/s at this batch (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-steps", type=int, default=3,
                    help="timed steps of the CPU baseline on the full workload graph after one warm-up step (SURVEY.md 8d: "
                         ">= 3 timed steps at C2/C3; a C2 step takes the 16 host threads about 70 s)")
    ap.add_argument("--cpu-budget-s", type=float, default=330.0,
                    help="wall-clock cap of the CPU baseline leg: if the warm-up step shows that --cpu-steps timed steps "
                         "would not fit, fewer are timed (never fewer than one) and the line says so")
    ap.add_argument("--no-probes", action="store_true", help="skip the measured-ceiling probes (stream triad, row gathers)")
    ap.add_argument("--no-graph-replay", action="store_true", help="skip the HIP-graph replay leg of the C2 / C3 step")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short C3 (NGCF) and C4 (TGCN) legs that the default C2 run appends under extra.configs")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--force-shard", action="store_true",
                    help="run the sharded model even with one rank (exercises dist.py + RCCL init on one GPU)")
    ap.add_argument("--parallel", choices=["auto", "feature", "row", "both"], default="auto",
                    help="multi-GPU sharding of the node table: row = row ranges (the reference's folds; default from 8 "
                         "ranks), feature = columns (default up to 4 ranks when dim / ranks >= 16); both = time the two "
                         "partitions one after the other in the same job: the line is the one `auto` picks, the other is "
                         "printed under extra.partitions")
    ap.add_argument("--all-gather", choices=["collective", "direct"], default="collective",
                    help="row sharding: how a row block reaches the other ranks -- collective = all_gather_into_tensor (RCCL's "
                         "schedule for the communicator), direct = one grouped send / receive pair per peer (every GPU pushes its "
                         "block over the xGMI link it shares with each peer, SURVEY.md 8e); extra.collectives.probe times both")
    ap.add_argument("--chunks", type=int, default=0, help="row blocks per shard for the pipelined all-gathers (0 = default)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: every rank uses cuda:0 and the ranks exchange "
                         "through gloo (RCCL wants one device per rank); numbers from such a run are not scaling results")
    ap.add_argument("--no-fused-adam", action="store_true",
                    help="keep the optimizer a separate launch (default: LightGCN applies the table's Adam update inside its "
                         "last backward kernel, same arithmetic)")
    ap.add_argument("--model", choices=["lightgcn", "ngcf", "tgcn", "dgcf", "disengcn"], default="lightgcn",
                    help="lightgcn = C2 (headline); ngcf = C3 (same graph, D^-1 A + I, MFMA dense layers); "
                         "tgcn = C4 (tripartite, 1M/1M/2M nodes, D=128, k=25; use --steps 3 --warmup 1); "
                         "dgcf / disengcn = the C2 graph with dynamic per-factor edge weights (4 factors, 2 routing iterations)")
    return ap.parse_args()


def spmm_bytes(nnz, n_rows, D, epilogue_row_bytes):
    return nnz * (8 + 4 * D) + n_rows * (8 + 4 * D) + n_rows * epilogue_row_bytes


def kernel_source_sha16():
    """Identity of the SpMM kernel sources, so that counter data collected for another version is refused."""
    import hashlib
    h = hashlib.sha256()
    for f in ("spmm.hip", "graph.h", "common.h"):
        with open(os.path.join(ROOT, "tag-aware-recommendation_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_traffic(model, D, kname):
    """HBM-side bytes per launch of `kname` from the newest committed rocprofv3 PMC summary of this command
    (profiles/rNN_pmc_<config>.json, written by tools/profile.sh + tools/pmc_summary.py: separate --pmc passes,
    (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 note of MI355X_MICROARCH.md).  NOT measured in this run -- counters
    need the profiler -- so the source is named, and data collected for other kernel sources is refused."""
    import glob
    tag = {"lightgcn": "c2_lightgcn", "ngcf": "c3_ngcf"}.get(model)
    if tag is None or D != 64:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{tag}.json")))
    if not files:
        return None, "no committed PMC summary for this configuration"
    with open(files[-1]) as f:
        js = json.load(f)
    rel = os.path.relpath(files[-1], ROOT)
    sha = kernel_source_sha16()
    if js.get("kernel_source_sha16") != sha:
        return None, (f"refused: {rel} was collected for kernel sources {js.get('kernel_source_sha16', 'unrecorded')}, "
                      f"this tree has {sha}; re-run tools/profile.sh")
    k = js["kernels"].get(kname)
    if k is None:
        return None, f"refused: {rel} has no kernel named {kname}"
    return k["traffic_bytes_per_launch"], (f"{rel}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `{js.get('command', '?')}`, "
                                           f"collected {js.get('collected', '?')} at commit {js.get('head', '?')}, kernel sources {sha}; "
                                           "not re-measured by this run")


def _cpu_steps(om, model_kind, tabs, mats, A, layers, batches, budget_s):
    """One warm-up step, then up to len(batches) - 1 timed steps; fewer if the warm-up shows they would not fit `budget_s`."""
    params = tabs + mats
    opt = torch.optim.Adam(params, lr=0.01)
    if model_kind == "ngcf":
        W = {k: m for k, m in zip(_ngcf_keys(layers), mats)}
        fn = lambda b: om.ngcf_loss(tabs, W, A, layers, b, 0.0, "logsigmoid")
    else:
        fn = lambda b: om.lightgcn_loss(tabs, A, layers, b, 0.0, "softplus")
    t0 = time.perf_counter()
    om.adam_epoch(params, fn, batches[:1], opt)
    t_warm = time.perf_counter() - t0
    n_timed = max(1, min(len(batches) - 1, int((budget_s - t_warm) / max(t_warm, 1e-3))))
    t0 = time.perf_counter()
    om.adam_epoch(params, fn, batches[1:1 + n_timed], opt)
    return (time.perf_counter() - t0) / n_timed, n_timed, t_warm


def _ngcf_keys(layers):
    return [f"{w}_{k}" for k in range(layers) for w in ("W1", "b1", "W2", "b2")]


def cpu_baseline(args, model_kind, rp, col, val, n, nu, ni, epoch):
    """The CPU oracle (oracle/models.py: torch.sparse.mm on a COO tensor built like the reference's sp2tensor, F.normalize,
    softplus / logsigmoid, autograd, torch.optim.Adam -- checked op for op against the imported reference, tests/golden)
    timed on THIS workload: the same graph the GPU leg just ran (its CSR copied to the host), the same batch size,
    1 warm-up + `--cpu-steps` timed steps on the host cores of the box."""
    from oracle import adj as oadj, models as om
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))   # the box gives a one-GPU job a 16-core share of the host
    torch.set_num_threads(cores)
    D, L, B = args.dim, args.layers, args.batch
    want = max(1, args.cpu_steps)
    print(f"[bench] cpu_baseline: {model_kind} oracle on the full graph (nnz={int(rp[-1])}), 1+{want} steps, "
          f"{cores} threads ...", file=sys.stderr, flush=True)
    A = om.csr_to_torch(oadj.CSR(rp.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy(), (n, n)))
    tabs = [t.requires_grad_() for t in om.xavier_tables([(nu, D), (ni, D)], 2020)]
    mats = []
    if model_kind == "ngcf":
        for _ in range(L):
            for shp in ((D, D), (1, D), (D, D), (1, D)):
                t = torch.empty(*shp)
                torch.nn.init.xavier_uniform_(t)
                mats.append(t.requires_grad_())
    batches = [epoch[k * B:(k + 1) * B].cpu() for k in range(1 + want)]
    dt, n_timed, t_warm = _cpu_steps(om, model_kind, tabs, mats, A, L, batches, args.cpu_budget_s)
    name = "NGCF" if model_kind == "ngcf" else "LightGCN"
    return {"value": B / dt, "unit": "triplets/s", "cores": cores, "kind": "port", "ms_per_step": dt * 1e3,
            "steps_timed": n_timed, "warmup_steps": 1,
            "sample": f"CPU oracle (torch {torch.__version__} CPU ops, {cores} threads), {name} L={L} D={D} B={B} on the FULL "
                      f"workload graph ({nu} x {ni}, nnz={int(rp[-1])}): {dt:.2f} s/step over {n_timed} timed steps after 1 "
                      f"warm-up step ({t_warm:.1f} s)"
                      + ("" if n_timed == want else f"; {want} were asked for, the {args.cpu_budget_s:.0f} s budget of this leg fitted {n_timed}")
                      + "; no scaling applied"}


def probe_ceilings(dev, reps=5):
    """Measured ceilings of THIS box, taken before the timed region with two minimal kernels of the library
    (csrc/probe.hip): a stream triad over 3 x 1 GiB, a copy over 2 x 1 GiB and a read-only pass over 1 GiB (HBM streaming),
    and random whole-row gathers of 256-byte rows -- the
    access shape of the D = 64 neighbour gather -- from tables of 256 MB (fits the Infinity Cache), 512 MB (the C2 table)
    and 4 GB (HBM).  GB/s of the bytes each kernel is asked to move (12 B per triad element; row + 4-byte index per gather)."""
    from tagrec_amd import _lib
    lib = _lib.load()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        return min(ev[i].elapsed_time(ev[i + 1]) for i in range(reps)) * 1e-3

    out = {}
    n = 1 << 28
    a, b, c = (torch.empty(n, device=dev) for _ in range(3))
    b.fill_(1.0); c.fill_(2.0)
    # best of the plain and the non-temporal flavour of each (which one wins depends on what else the caches hold)
    tri = min(timed(lambda nt=nt: _lib.check(lib.tagrec_probe_triad_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), 0.5, n, nt, _lib.stream_ptr()),
                                              "triad")) for nt in (0, 1))
    cpy = min(timed(lambda nt=nt: _lib.check(lib.tagrec_probe_triad_f32(_lib.ptr(a), _lib.ptr(b), None, 0.0, n, nt, _lib.stream_ptr()),
                                              "copy")) for nt in (0, 1))
    out["stream_triad_3x1GiB"] = 12.0 * n / tri / 1e9
    out["stream_copy_2x1GiB"] = 8.0 * n / cpy / 1e9
    sink0 = torch.empty(lib.tagrec_probe_gather_out_floats(), device=dev)
    rd = timed(lambda: _lib.check(lib.tagrec_probe_read_f32(_lib.ptr(b), n, _lib.ptr(sink0), _lib.stream_ptr()), "read"))
    out["stream_read_1GiB"] = 4.0 * n / rd / 1e9
    del a, b, c, sink0
    n_idx = 1 << 26
    sink = torch.empty(lib.tagrec_probe_gather_out_floats(), device=dev)
    table = torch.empty((4 << 30) // 256, 64, device=dev).normal_()
    for label, mb in (("gather_256B_rows_from_256MB", 256), ("gather_256B_rows_from_512MB", 512), ("gather_256B_rows_from_4GB", 4096)):
        rows = (mb << 20) // 256
        idx = torch.randint(0, rows, (n_idx,), device=dev, dtype=torch.int32)
        t = timed(lambda: _lib.check(lib.tagrec_probe_gather_rows_f32(_lib.ptr(table), rows, 64, _lib.ptr(idx), n_idx, _lib.ptr(sink),
                                                                      _lib.stream_ptr()), "gather"))
        out[label] = n_idx * 260.0 / t / 1e9
        del idx
    del table, sink
    torch.cuda.empty_cache()
    return {k: round(v, 1) for k, v in out.items()}


def bench_tgcn(args, ceilings=None):
    """C4: TGCN 3-layer dim 128 on a synthetic tripartite graph (1M users, 1M items, 2M tags, 100M
    assignments, k=25 neighbours per relation).  A step = one BPR mini-batch of `epoch_training` phase 0.
    Returns the JSON object of the line."""
    import tagrec_amd as T
    from tagrec_amd import tgcn as TG
    from tagrec_amd.tgcn_step import layer_params as TS_layer_params, _dense_views as TS_dense_views
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
    if not args.no_fused_adam:            # the node tables' update in the epilogue of the last product that lands on them
        opt.fuse_into(model)
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
    import gc
    gc.collect()
    gc.freeze()                   # (see main(): a full collection inside the timed region is a ~30 ms host stall)
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
                "traffic_source": "none: rocprofv3 averages a kernel over all its launches, and this step launches the kernel "
                                  "on row subsets of varying size; per-kernel means of the step are in profiles/r02_pmc_c4_tgcn.csv",
                "algorithmic_bytes_per_launch": alg, "mean_launch_ms": m, "launches_timed": len(fwd),
                "measured_on": "one full forward pass over all rows, after the timed steps",
                "all_rows_kernels_ms": {kk: sum(v) / len(v) for kk, v in full_ms.items() if kk != "attn_fwd"},
                "step_kernels_ms_per_step": {kk: sum(v) / K for kk, v in ms.items()}}
    n_nodes = nu + ni + nt
    # the other bounding roofline of C4: the fusion product of the fused dense kernel on the MFMA pipe
    fuse_all = full_ms.get("fuse_fwd", [])
    roof_mfma = None
    if fuse_all:
        tf = L * 2 * n_nodes * (32 * D + 48) * D / 1e12 / (sum(fuse_all) * 1e-3)
        roof_mfma = {"bound": "mfma", "kernel": "tgcn_fuse_fwd kernel (type attention + convolutions + fusion layer), all rows, "
                                                "flops of the fusion product 2 n (32 D + 48) D only",
                     "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                     "traffic": None, "ms_all_rows_per_layer": sum(fuse_all) / L}
    if roof_mfma is not None:
        # the clock the kernel actually runs at: the 157 TFLOP/s peak is quoted at 2.4 GHz, the chip lowers its clock under
        # the fp32 matrix load.  A one-wave sampler (tagrec_probe_clock) on a second stream, launched first, spins through
        # one forward launch on 1 M random rows.
        try:
            from tagrec_amd import _lib
            g_ = torch.Generator(device=dev).manual_seed(0)
            rnd = lambda *sh: torch.randn(*sh, device=dev, generator=g_) * 0.1
            tt3 = [rnd(1_000_000, D) for _ in range(3)]
            lp = TS_dense_views([p_.detach() for p_ in TS_layer_params(model.layer["0"])[12:]])
            res = torch.zeros(2, dtype=torch.int64, device=dev)
            side = torch.cuda.Stream()
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                _lib.check(_lib.load().tagrec_probe_clock(6000, _lib.ptr(res), _lib.c_void_p(side.cuda_stream)), "probe_clock")
            with torch.no_grad():
                TG._FusedDense.apply(*tt3, *lp, 0)
            torch.cuda.synchronize()
            cyc, ticks = res.tolist()
            mhz = cyc / ticks * 100.0
            roof_mfma["shader_clock_mhz_under_kernel"] = round(mhz)
            roof_mfma["frac_at_measured_clock"] = tf / (MFMA_F32_PEAK_TFLOPS * mhz / 2400.0)
            del tt3
        except Exception as e:                                   # a diagnostic: never fails the bench line
            roof_mfma["shader_clock_mhz_under_kernel"] = f"not measured: {e}"
    if roof is not None and ceilings:
        roof["measured_ceilings"] = ceilings
    out = {"metric": f"BPR triplets/sec, TGCN {L}-layer dim{D}, tripartite {nu}/{ni}/{nt} nodes, k={k}",
           "value": K * B / dt, "unit": "triplets/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"C4 TGCN L={L} D={D} users={nu} items={ni} tags={nt} assignments={int(ds.uit_data.shape[0])} "
                                  f"k={k} train_batch={B} adam lr=0.01 logsigmoid", "train_batch": B, "parallelism": "single"},
           "roofline": roof, "roofline_mfma": roof_mfma,
           "cpu_baseline": "not run (the reference cannot materialise this size: 4 M x (32 D + 48) floats "
                           "of convolution output per layer)",
           "extra": {"build_s": round(t_build, 1), "last_loss": [float(x.detach()) for x in last],
                     "transtag_step_ms": t_tt * 1e3,
                     "attention_ms_per_step": sum(sum(v) for kk, v in ms.items() if kk.startswith("attn")) / K,
                     "pruned_forward": bool(model.prune_forward),
                     "fused_adam": bool(not args.no_fused_adam),
                     "fused_dense_ms_per_step": sum(sum(ms.get(kk, [])) for kk in ("fuse_fwd", "fuse_bwd", "fuse_wf")) / K,
                     "note": "the step is one hand-derived autograd node (tgcn_step.py): fused MFMA dense block (csrc/tgcn_fuse.hip: fwd, "
                             "bwd-data, bwd-Wf; the three node types of a layer in one launch), neighbour attention (csrc/tgcn.hip) with the "
                             "backward pulled over on-the-spot inverted tables (csrc/spmm.hip attn_pull_*), projections on csrc/proj.hip; "
                             "no library GEMM in the step; the CPU reference cannot materialise this size"}}
    del model, opt, prod, ds
    torch.cuda.empty_cache()
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start N fresh rank processes (this parent has made no
    GPU call and makes none), hand rank 0's JSON line through, return the worst exit code."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()              # counts devices without initialising the GPU
    if n_dev < args.gpus and not args.share_gpu:
        print(json.dumps({"error": f"bench.py --gpus {args.gpus}: this machine exposes {n_dev} GPU(s); one rank per GPU is "
                                   "required (RCCL).  For a functional rehearsal on fewer GPUs add --share-gpu "
                                   "(ranks share cuda:0 and exchange through gloo).", "n_gpus": args.gpus}))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def collective_probe(shape, dev, world, rank, reps=5):
    """ms and achieved GB/s (bytes received per rank / time) of one exchange of a [rows, D] block per rank, by the collective
    (all_gather_into_tensor) and by the direct schedule (one grouped send / receive pair per peer), and ms of one
    all_reduce of a [3, 1536, D] buffer -- the exchange shapes of the row-sharded step, measured before the timed region.
    On the xGMI full mesh a direct exchange is bounded by ONE link per peer (~64 GB/s per direction each, all in parallel: rows
    * D * 4 B / 64 GB/s), a ring by (G - 1) hops over one link: the two GB/s figures tell which one ran."""
    import torch.distributed as dist
    rows, D = shape
    x = torch.zeros(rows, D, device=dev)
    full = torch.empty(rows * world, D, device=dev)
    small = torch.zeros(3, 1536, D, device=dev)

    def direct():
        if dist.get_backend() != "nccl":
            torch.cuda.synchronize()               # (gloo rehearsal: its point-to-point path does not look at HIP streams)
        ops = []
        for d in range(1, world):
            to, frm = (rank + d) % world, (rank - d) % world
            ops.append(dist.P2POp(dist.isend, x, to))
            ops.append(dist.P2POp(dist.irecv, full[frm * rows:(frm + 1) * rows], frm))
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    out = {}
    nbytes = rows * D * 4 * (world - 1)
    for name, fn in (("all_gather_block_collective", lambda: dist.all_gather_into_tensor(full, x)),
                     ("all_gather_block_direct", direct),
                     ("all_reduce_batch_rows", lambda: dist.all_reduce(small))):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
        out[name + "_ms"] = ms
        if name.startswith("all_gather"):
            out[name + "_GBs_in"] = nbytes / (ms * 1e-3) / 1e9
    out["all_gather_block_bytes_in"] = nbytes
    out["one_link_bound_ms_at_64GBs"] = rows * D * 4 / 64e9 * 1e3
    return out


def run_table_model(args, dev, rank, world, sharded, ceilings=None, ds=None, light=False):
    """One table-model leg (LightGCN = C2 headline, NGCF = C3, DGCF / DisenGCN on the C2 graph; single GPU or sharded):
    build, warm up, time exactly `args.steps` steps between barriers.  Returns (JSON object of the line, inputs of the CPU
    baseline leg, the dataset for a following leg on the same graph).  light: no all-rows / big-batch variants."""
    import tagrec_amd as T
    if sharded:
        import torch.distributed as dist
        from tagrec_amd import dist as TD

    nu = ni = max(int(1_000_000 * args.scale), 2000)
    ne = max(int(50_000_000 * args.scale), 40000)
    D, L, B = args.dim, args.layers, args.batch
    cfg = T.get_config(args.model, use_tag=False, dim_latent=D, dim_layer_list=[D] * L, device=dev, train_batch=B,
                       all_gather=args.all_gather)
    parallel = args.parallel
    if parallel == "auto":
        # Byte budget of DESIGN.md section 6: the row partition (the reference's folds) sends 4 table shards per step over
        # every xGMI link -- 1024 / 512 / 256 MB per link at 2 / 4 / 8 ranks for the 512 MB C2 table, i.e. ~16 / 8 / 4 ms at
        # 64 GB/s -- against 6.6 / 3.5 / 1.9 ms of per-rank compute.  The column partition exchanges three small all-reduces
        # per step and its per-rank step was measured on one rank's slice: 7.5 / 6.9 / 6.5 ms (32 / 16 / 8 columns; below
        # 32 columns a gathered row is shorter than the 128-byte line).  Projected step, row vs column: 2 ranks 20 vs 7.6 ms,
        # 4 ranks 10.4 vs 7.0 ms, 8 ranks 5.4 vs 6.6 ms: columns up to 4 ranks, rows from 8.
        # At D / ranks >= 32 (C5: 256 / 8) a column slice still gathers whole 128-byte rows and exchanges nothing but three
        # small all-reduces: one rank's slice of the C5 shape runs at 79 ms against 436 ms for the whole problem on one
        # GPU (5.5x at 8 ranks), where the row partition would move 4 x 2.56 GB per link and step (~160 ms).
        col_ok = D % world == 0 and D // world >= 8
        parallel = "feature" if (col_ok and (D // world >= 32 or world <= 4)) else "row"
    parallelism_note = None
    if sharded and args.parallel == "auto" and parallel != "row" and args.model == "lightgcn":
        # BASELINE.json's multi-GPU configuration names the ROW partition ("embedding tables row-sharded ... RCCL
        # all-reduce"); `auto` deviates from it here, so say so and put both projections side by side
        tab_mb = 2 * nu * D * 4 / 1e6
        link_ms = 4 * (tab_mb / world) / 64.0             # four shard exchanges per step, one shard per xGMI link, 64 GB/s
        parallelism_note = (
            f"DEVIATION from the north star's row partition: --parallel auto picked the COLUMN (feature) partition at "
            f"{world} ranks (D / ranks = {D // world}).  Row partition (the reference's split_adj_k folds): 4 table-shard "
            f"exchanges per step = {4 * tab_mb / world:.0f} MB per xGMI link = ~{link_ms:.1f} ms at 64 GB/s per direction "
            f"before compute (projection, DESIGN.md section 6: C2 ~20 / 10.4 / 5.4 ms per step at 2 / 4 / 8 ranks, C5 ~160 ms at 8); "
            f"column partition: three all-reduces of a few KB per step, per-rank slice measured on one GPU "
            f"(C2: 7.4 / 7.0 / 6.7 ms at 2 / 4 / 8 ranks, C5 at 8 ranks: 79 ms vs 436 ms on one GPU).  Run --parallel row, "
            f"or --parallel both for the two partitions in one job.")
    Dl = D // world if (sharded and parallel == "feature") else D
    if args.model not in ("lightgcn", "ngcf") and sharded:
        sys.exit(f"bench.py: the sharded path covers LightGCN (C2/C5) and NGCF (C3); run --model {args.model} on one GPU")
    if args.model == "ngcf" and sharded:
        parallel, Dl = "row", D                 # W1 / W2 mix the columns: NGCF shards by rows only
    routed = args.model in ("dgcf", "disengcn")

    t0 = time.perf_counter()
    if ds is None:
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
        if args.model == "ngcf":
            model = TD.ShardedNGCF(ds, cfg, rp, col, val, n, n_chunks=args.chunks or None)
            timed_graph = model.graph_chunks[0]
        elif parallel == "feature":
            model = TD.FeatureShardedLightGCN(ds, cfg, rp, col, val, n)
            timed_graph = model.graph
        else:
            model = TD.ShardedLightGCN(ds, cfg, rp, col, val, n, n_chunks=args.chunks or None)
            timed_graph = model.graph
        del rp, col, val
    opt = T.Adam(model.parameters(), lr=cfg["lr"])
    if not args.no_fused_adam:
        opt.fuse_into(model)          # the table's Adam update runs in the epilogue of the last backward product (models with the hook)
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
    # Python's cyclic collector walks every tracked object of the process when a full collection triggers -- with the
    # synthetic dataset's structures alive that is a ~30 ms host stall in the middle of a timed step (seen as one 40 ms
    # window in a rocprofv3 trace of 7.5 ms steps).  Collect now and park what exists in the permanent generation.
    import gc
    gc.collect()
    gc.freeze()
    row_sharded = sharded and parallel == "row"
    probe = collective_probe((model.part.rc, D), dev, world, rank) if (row_sharded and world > 1) else None
    # per-kernel HIP events on the launch stream; a row shard is walked in row blocks, one handle (and event list) each
    timed_graphs = model.graph_chunks if row_sharded else [timed_graph]
    for g_ in timed_graphs:
        g_.timing = {}
    if sharded:
        model.timing, model.comm_bytes = {}, 0
    dt, last = timed(batches[W:])
    per_block = [g_.timing_ms() for g_ in timed_graphs]
    kernel_ms = {k: [sum(v) for v in zip(*[pb.get(k, []) for pb in per_block])] for k in per_block[0]}
    for g_ in timed_graphs:
        g_.timing = None
    loss_val = [float(x) for x in last]
    comm = None
    if sharded:
        waits = model.timing_ms()
        model.timing = None
        comm = {"world_size_seen_by_torch_distributed": dist.get_world_size(), "backend": dist.get_backend(),
                "all_gather": getattr(model, "all_gather_mode", None),
                "row_blocks_per_shard": model.part.n_chunks if row_sharded else None,
                "rows_per_rank": model.part.per if row_sharded else n,
                "columns_per_rank": D if row_sharded else Dl,
                "bytes_exchanged_per_rank_per_step": model.comm_bytes / K,
                "compute_stream_wait_ms_per_step": {k: sum(v) / K for k, v in waits.items()},
                "collectives_per_step": {k: len(v) / K for k, v in waits.items()},
                "probe": probe}

    extra = {"graph_build_s": round(t_build, 2), "epoch_sampling_s": round(t_sample, 3),
             "epoch_triplets": int(epoch.shape[0]), "last_loss": loss_val,
             "edge_layers_per_s": nnz * L * 2 * K / dt}
    if comm is not None:
        extra["collectives"] = comm
    if args.model == "lightgcn" and not sharded and not light and getattr(model, "restrict_forward", False):
        # the same step with every forward layer computed on ALL rows (the timed step above computes the top two
        # layers only on the rows the batch's loss depends on -- same loss and gradients)
        ws_mb = round(model.step_ws.nbytes() / 2 ** 20, 1) if getattr(model, "step_ws", None) is not None else None
        if ws_mb is not None:
            model.step_ws.clear()          # the all-rows step does not use the workspace and needs its memory at the C5 shape
            torch.cuda.empty_cache()
        model.restrict_forward = False
        run_steps(batches[:W])
        dtf, _ = timed(batches[W:])
        model.restrict_forward = True
        extra["step_workspace_MB"] = ws_mb
        extra["ms_per_step_all_rows_forward"] = dtf / K * 1e3
        extra["triplets_per_s_all_rows_forward"] = K * B / dtf
    if args.big_batch and epoch.shape[0] >= args.big_batch * 3 and not sharded and not light:
        BB = args.big_batch
        bb = [epoch[k * BB:(k + 1) * BB] for k in range(3)]
        run_steps(bb[:1])
        dtb, _ = timed(bb[1:])
        extra[f"triplets_per_s_at_B{BB}"] = 2 * BB / dtb
        extra[f"ms_per_step_at_B{BB}"] = dtb / 2 * 1e3

    if args.model in ("lightgcn", "ngcf") and not sharded and not light and not routed and not args.no_graph_replay:
        # the SAME step replayed as one captured HIP graph (train.GraphedStep): the optimizer's step counter and factors live
        # in device memory (Adam(capturable=True)), the table's update stays inside the last backward product
        try:
            opt_g = T.Adam(model.parameters(), lr=cfg["lr"], capturable=True)
            if not args.no_fused_adam:
                opt_g.fuse_into(model)
            for b in batches[:2]:                      # eager: optimizer state and every lazily sized buffer exist before capture
                lossx = model.loss(b)
                opt_g.zero_grad()
                sum(lossx).backward()
                opt_g.step()
            gstep = T.GraphedStep(model.loss, opt_g, batches[0])
            for b in batches[:W]:
                gstep(b)
            barrier()
            t = time.perf_counter()
            for b in batches[W:]:
                gstep(b)
            barrier()
            dtg = time.perf_counter() - t
            extra["hip_graph_replay"] = {"ms_per_step": dtg / K * 1e3, "eager_ms_per_step": dt / K * 1e3,
                                         "fused_adam": bool(not args.no_fused_adam),
                                         "step_workspace_MB": round(model.step_ws.nbytes() / 2 ** 20, 1)
                                         if getattr(model, "step_ws", None) is not None else None}
            del gstep
            if not args.no_fused_adam:
                opt.fuse_into(model)                   # hand the table back to the eager optimizer
        except Exception as exc:                       # capture is an extra: the headline above stands without it
            extra["hip_graph_replay"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            torch.cuda.synchronize()

    # roofline of the dominant kernel: fused forward layer (local rows of this rank)
    dom = "spmm_norm_acc" if args.model == "lightgcn" else "spmm"
    # LightGCN's training step forms the layer mean on the batch rows only, so its forward layers run the NORM_ACC
    # epilogue WITHOUT the accumulator (acc = NULL): per row the product, its norm (4 B) and nothing else
    epi_row_bytes = 4 if args.model == "lightgcn" else 0
    KF = cfg.get("factor_k", 1)
    if routed:
        dom = "route_spmm"
    if sharded and parallel == "feature":
        dom, epi_row_bytes = "spmm_ss", 4                  # Y = A X on D/N columns + one float of row sum-of-squares
    fwd = kernel_ms.get(dom, [])
    n_local_rows = model.per if row_sharded else timed_graph.shape[0]     # (a shard is walked in row blocks: whole shard)
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
    traffic, traffic_source = None, None
    epi_name = "SS" if dom == "spmm_ss" else ("NORM_ACC" if args.model == "lightgcn" else "NONE")
    epi_id = {"NONE": 0, "NORM_ACC": 1, "SS": 4}[epi_name]
    prof_kname = f"spmm_rows_kernel<{Dl // 4}, {epi_id}, false>"
    if world == 1 and args.scale == 1.0 and not routed:
        traffic, traffic_source = committed_traffic(args.model, D, prof_kname)
    if fwd:
        ms = sum(fwd) / len(fwd)
        ach = alg / (ms * 1e-3) / 1e9
        kname = (f"route_spmm_kernel<{D // 4}, {KF}>" if routed else
                 f"spmm_rows_kernel<{Dl // 4}, {epi_name}{', acc = NULL' if epi_name == 'NORM_ACC' else ''}>")
        # compulsory bytes: every array of the launch touched exactly once (CSR, gathered table, outputs, accumulator
        # read-modify-write) -- what a perfect cache would leave for HBM
        comp = local_nnz * 8 + (n_local_rows + 1) * 8 + timed_graph.shape[1] * Dl * 4 + n_local_rows * Dl * 4
        comp += {"NORM_ACC": n_local_rows * 4, "SS": n_local_rows * 4, "NONE": 0}[epi_name]
        roof = {"bound": "hbm", "kernel": kname + " (+ long-row finish)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg, "compulsory_bytes": comp,
                "mean_launch_ms": ms, "launches_timed": len(fwd),
                "other_kernels_ms": {k: sum(v) / len(v) for k, v in kernel_ms.items() if k != dom}}
        if traffic:
            # rate of the bytes that did cross the L2 <-> fabric boundary (HBM + Infinity Cache), from the PMC passes
            roof["frac_counter"] = traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof["refetch_factor"] = traffic / comp
        if ceilings:
            # ceilings MEASURED on this box before the timed region (probe_ceilings): what a stripped gather of the same
            # row shape reaches from a table of the workload's size, and the HBM streaming rate
            roof["measured_ceilings"] = ceilings
            tab_mb = timed_graph.shape[1] * Dl * 4 / 2 ** 20
            gkey = ("gather_256B_rows_from_256MB" if tab_mb <= 256 else
                    "gather_256B_rows_from_512MB" if tab_mb <= 1024 else "gather_256B_rows_from_4GB")
            if Dl == 64 and ceilings.get(gkey):
                roof["gather_ceiling"] = gkey
                roof["frac_of_gather_ceiling"] = ach / ceilings[gkey]
            if ceilings.get("stream_read_1GiB"):
                roof["frac_of_stream_read"] = ach / ceilings["stream_read_1GiB"]
        if roof["frac"] > 1.0:
            roof["note"] = ("frac > 1: `achieved` bills every gathered row to HBM (SURVEY.md 8d, no cache credit) while part "
                            "of them is served by L2 / Infinity Cache, so the 8 TB/s specification is not this kernel's "
                            "ceiling; frac_of_gather_ceiling prices it against the measured random-row gather rate of this "
                            "box, frac_counter is the counted fabric-side rate, refetch_factor = traffic / compulsory_bytes")
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
        cfg_tag = {"lightgcn": "C2", "ngcf": "C3"}.get(args.model, "C2-graph")
        if args.scale != 1.0 or (args.model in ("lightgcn", "ngcf") and D != 64):
            cfg_tag = ("C5-shape (10M x 10M x 500M edges, dim 256) on ONE GPU" if (args.model == "lightgcn" and args.scale == 10.0
                       and D == 256 and world == 1) else f"{cfg_tag}-shaped (scale {args.scale:g}, dim {D})")
        size = "1M users x 1M items x 50M edges" if args.scale == 1.0 else f"{nu} users x {ni} items x {ne} edges"
        out = {"metric": f"BPR triplets/sec, {mname} {L}-layer dim{D}, {size}",
               "value": K * B / dt, "unit": "triplets/s", "n_gpus": world, "steps": K, "warmup": W,
               "ms_per_step": dt / K * 1e3, "higher_is_better": True,
               "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{cfg_tag} {mname} L={L} D={D} users={nu} items={ni} "
                                      f"edges={ne} nnz={nnz} train_batch={B} adam lr=0.01 {cfg['norm_type']} {cfg['mul_loss_func']}",
                          "train_batch": B, "parallelism": f"{parallel}-shard x{world}" if sharded else "single",
                          **({"parallelism_note": parallelism_note} if parallelism_note else {}),
                          "step": "loss -> backward -> Adam on one batch; same loss and gradients as the all-rows step: rows "
                                  "of the top two forward layers that the batch's loss does not read are not computed, "
                                  "backward products do not fetch operand rows that are exactly zero "
                                  "(extra.ms_per_step_all_rows_forward = every forward layer on all rows)"
                                  + ("; the table's Adam update (torch's arithmetic, bit-identical) runs in the epilogue of the "
                                     "last backward product instead of a separate launch (--no-fused-adam separates them)"
                                     if (not args.no_fused_adam and getattr(model, "_fused_opt", None) is not None) else "")},
               "roofline": roof, "extra": extra}
        cpu_inputs = (rp, col, val, n, nu, ni, epoch) if (world == 1 and args.model in ("lightgcn", "ngcf") and not sharded) else None
        del model, opt, prod
        torch.cuda.empty_cache()
        return out, cpu_inputs, ds
    return None, None, ds


def _leg_args(args, **kw):
    import copy
    a = copy.copy(args)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _compact(line):
    """What extra.configs keeps of a leg's line: the figures and the rooflines, not the prose."""
    keep = {k: line[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype") if k in line}
    keep["workload"] = line["config"]["workload"]
    for rk in ("roofline", "roofline_mfma"):
        r = line.get(rk)
        if r:
            keep[rk] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "mean_launch_ms", "launches_timed",
                                          "frac_of_gather_ceiling", "frac_of_stream_read", "frac_counter", "traffic",
                                          "step_kernels_ms_per_step", "ms_all_rows_per_layer", "shader_clock_mhz_under_kernel",
                                          "frac_at_measured_clock") if k in r}
    return keep


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and args.model != "tgcn":
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.model == "tgcn":
        world, rank, local = 1, 0, 0
    args.gpus = world
    if args.share_gpu:
        local = 0
    assert torch.cuda.is_available(), "bench.py needs a GPU (tagrec_amd has no CPU path)"
    if local >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} wants cuda:{local} but only {torch.cuda.device_count()} GPU(s) are visible "
                 "(one rank per GPU; --share-gpu for a rehearsal)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sharded = (world > 1 or args.force_shard) and args.model != "tgcn"
    if sharded:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.share_gpu and world > 1:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    # measured ceilings of this box (rank 0's GPU; every rank runs them so that the ranks stay in step)
    ceilings = None if args.no_probes else probe_ceilings(dev)
    if args.model == "tgcn":
        print(json.dumps(bench_tgcn(args, ceilings)))
        return
    if args.parallel == "both" and sharded and args.model == "lightgcn":
        # both partitions of the SAME problem in one job: the line is `auto`'s choice, the other one rides along
        o_auto, cpu_inputs, ds = run_table_model(_leg_args(args, parallel="auto"), dev, rank, world, sharded, ceilings)
        picked = "feature" if (rank != 0 or "feature" in o_auto["config"]["parallelism"]) else "row"
        if world > 1:
            t_ = torch.tensor([1 if picked == "feature" else 0], device=dev)
            dist.broadcast(t_, 0)
            picked = "feature" if int(t_) else "row"
        other = "row" if picked == "feature" else "feature"
        o_other, _, ds = run_table_model(_leg_args(args, parallel=other), dev, rank, world, sharded, ceilings, ds=ds, light=True)
        out = o_auto
        if rank == 0:
            out["extra"]["partitions"] = {picked: {"ms_per_step": o_auto["ms_per_step"], "value": o_auto["value"]},
                                          other: {**_compact(o_other), "collectives": o_other["extra"].get("collectives")}}
    else:
        if args.parallel == "both":
            args.parallel = "auto"
        out, cpu_inputs, ds = run_table_model(args, dev, rank, world, sharded, ceilings)
    if rank == 0:
        headline = (args.model == "lightgcn" and not sharded and args.scale == 1.0 and args.dim == 64 and args.layers == 3)
        if headline and not args.no_extra_configs:
            # the other single-GPU configurations of BASELINE.json, short legs in the same process so that the driver's
            # clock covers them: C3 = NGCF on the same graph, C4 = TGCN on the tripartite graph
            legs = {}
            print("[bench] extra.configs: C3 NGCF leg ...", file=sys.stderr, flush=True)
            c3, _, _ = run_table_model(_leg_args(args, model="ngcf", steps=10, warmup=3, big_batch=0), dev, 0, 1, False, ceilings,
                                       ds=ds, light=True)
            legs["C3"] = _compact(c3)
            del ds
            torch.cuda.empty_cache()
            print("[bench] extra.configs: C4 TGCN leg ...", file=sys.stderr, flush=True)
            legs["C4"] = _compact(bench_tgcn(_leg_args(args, model="tgcn", steps=8, warmup=3, dim=128), ceilings))
            out["extra"]["configs"] = legs
        ds = None
        torch.cuda.empty_cache()
        if not args.no_cpu and cpu_inputs is not None:
            out["cpu_baseline"] = cpu_baseline(args, args.model, *cpu_inputs)
        else:
            out["cpu_baseline"] = ("not run (--no-cpu)" if args.no_cpu else
                                   "not run (timed on rank 0 of the 1-GPU configuration only)" if world > 1 or sharded else
                                   "not run (no CPU oracle leg for this model in bench.py)")
        print(json.dumps(out))
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
