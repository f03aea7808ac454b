"""CPU-only: the C-ABI library loads and exports exactly what include/tagrec.h declares;
host-side logic (adjacency build, producer protocol, config defaults) against the golden vectors."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, blocks_from_fixture

import tagrec_amd as T
from tagrec_amd import graph as G
from tagrec_amd.synth import Coo


def _header_functions():
    text = open(os.path.join(ROOT, "include", "tagrec.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tagrec_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = T._lib.load()
    declared = _header_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tagrec.h but not exported"
    assert sorted(T._lib.exported_symbols()) == declared, "ctypes table and header disagree"
    assert lib.tagrec_abi_version() == 2


def test_missing_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ds = T.synth.make_cf_dataset(20, 15, 80, seed=0)
    with pytest.raises(T.TagrecError):
        T.LightGCN(ds, config=T.get_config("lightgcn", use_tag=False, device="cpu"))
    with pytest.raises(T.TagrecError):
        T.Graph(torch.zeros(3, dtype=torch.int64), torch.zeros(0, dtype=torch.int32), torch.zeros(0), (2, 2))


def _coo(block):
    r, c, v, shape = block
    return Coo(np.asarray(r), np.asarray(c), np.asarray(v), shape)


@pytest.mark.parametrize("use_tag", [0, 1])
@pytest.mark.parametrize("norm", ["bi_norm", "ngcf", "si_norm", "si_norm_self", "plain"])
def test_host_adjacency_bit_exact_vs_reference(golden, use_tag, norm):
    fx = golden("adj_toy")
    ui, ut, it = blocks_from_fixture(fx, use_tag)
    if use_tag:
        rowptr, col, val, n = G.block_adjacency_host(_coo(ui), _coo(ut), _coo(it))
    else:
        rowptr, col, val, n = G.block_adjacency_host(_coo(ui))
    rowptr, col, val = G.normalise_host(rowptr, col, val, n, norm)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    assert np.array_equal(np.stack([rows, col.astype(np.int64)]), fx[f"{norm}_{use_tag}_idx"])
    assert np.array_equal(val, fx[f"{norm}_{use_tag}_val"])


def test_fold_bounds_match_split_sp_mat(golden):
    fx = golden("adj_toy")
    n = int(fx["n_user"] + fx["n_item"] + fx["n_tag"])
    bounds = G.fold_bounds(n, 3)
    assert [hi - lo for lo, hi in bounds] == [int(fx[f"fold3_{k}_shape"][0]) for k in range(3)]
    assert G.fold_bounds(10, 1) == [(0, 10)]


def test_mini_batch_protocol(golden):
    fx = golden("producer")
    for key in (k for k in fx if k.startswith("mb_")):
        _, n, B = key.split("_")
        prod = T.Fixed_training_data([np.arange(int(n))[:, None].repeat(3, 1)], int(B), "cpu")
        prod.reset()
        got = [(int(b[0, 0]), int(b[-1, 0]) + 1) for b in prod.mini_batch()]
        assert got == [tuple(r) for r in fx[key].tolist()], key


def test_config_defaults_match_reference():
    cfg = T.get_config("lightgcn")
    assert (cfg["train_batch"], cfg["test_batch"], cfg["lr"], cfg["reg"], cfg["dim_latent"]) == (512, 512, 0.01, 0.0, 64)
    assert cfg["dim_layer_list"] == [64, 32, 16] and cfg["topks"] == [10, 20] and cfg["seed"] == 2020
    assert cfg["mul_loss_func"] == "softplus" and cfg["norm_type"] == "bi_norm"
    n = T.get_config("ngcf")
    assert (n["mul_loss_func"], n["norm_type"], n["agg_type"]) == ("logsigmoid", "ngcf", "bi_agg")
    t = T.get_config("tgcn")
    assert (t["neighbor_k"], t["dim_atten"], t["num_bit_conv"], t["num_vec_conv"], t["margin"]) == (25, 32, 32, 8, 1)
    d, k = T.get_config("dgcf"), T.get_config("kgat")
    assert (d["factor_k"], d["iterate_k"], d["norm_type"], d["mul_loss_func"]) == (4, 2, "plain", "softplus")
    assert (k["dim_relation"], k["transe_batch"], k["agg_type"]) == (64, 1024, "bi_agg")
    with pytest.raises(KeyError):
        T.get_config("disenhan")


def test_text_loader_matches_reference_loader(golden, tmp_path):
    """tagrec_amd.data.TGCN_load on the same train/test/user_item_tag files as the reference's TGCN_load
    (fixture loader_toy.npz holds the file bytes and what the reference parsed)."""
    fx = golden("loader_toy")
    d = tmp_path / "toyset"
    d.mkdir()
    for n in ("train.txt", "test.txt", "user_item_tag.txt"):
        (d / n).write_bytes(fx["files." + n].tobytes())
    ld = T.data.TGCN_load(str(tmp_path), "toyset")
    assert {k: int(v) for k, v in ld.num.items()} == {k[4:]: int(fx[k]) for k in fx if k.startswith("num.")}
    for split in ("train", "test"):
        e = ld.edge_index[split]
        assert np.array_equal(e[np.lexsort((e[:, 1], e[:, 0]))], fx["edges." + split])
        assert sum(len(v) for v in ld.user_items[split].values()) == len(e)
    assert np.array_equal(ld.uit_data, fx["uit_data"])
    for nm in ("ui_adj", "ut_adj", "it_adj"):
        c = getattr(ld, nm)
        assert tuple(c.shape) == tuple(fx[nm + ".shape"])
        key = c.row.astype(np.int64) * c.shape[1] + c.col
        uk, cnt = np.unique(key, return_counts=True)          # duplicates are summed when the adjacency is built
        assert np.array_equal(uk, fx[nm + ".row"].astype(np.int64) * c.shape[1] + fx[nm + ".col"])
        assert np.array_equal(cnt.astype(np.float32), fx[nm + ".data"])
    # round trip through our own writer
    T.data.write_dataset(ld, str(tmp_path), "again")
    ld2 = T.data.TGCN_load(str(tmp_path), "again")
    assert ld2.num == ld.num and np.array_equal(ld2.uit_data, ld.uit_data)


def test_create_edge_matches_reference(golden):
    """`Dataset.create_edge` against what the reference's `TGCN_load.create_edge` produced for the same dataset
    (fixture kgat_toy_wired holds those arrays verbatim)."""
    fx = golden("kgat_toy_wired")
    toy = T.synth.make_cf_dataset(40, 30, 300, seed=1, n_tag=12, n_assign=200)
    edges = toy.create_edge()
    assert sorted(edges) == list(range(6))
    for k in range(6):
        assert np.array_equal(edges[k], fx[f"edges.{k}"])


def test_early_stop_protocol(tmp_path):
    import types
    cfg = T.get_config("lightgcn", patient_epoch=1)
    es = T.Early_stop(types.SimpleNamespace(out_dir=str(tmp_path)), cfg)
    model = torch.nn.Linear(2, 2)
    assert es(model, {"ndcg": [0.5, 0.6]}, 0) is False and es.best_epoch == 0       # key ndcg -> NDCG@10 = first entry
    assert (tmp_path / "model.pth.tar").exists()
    assert es(model, {"ndcg": [0.4, 0.9]}, 5) is False and es.count_step == 1       # worse @10 although better @20
    assert es(model, {"ndcg": [0.45, 0.9]}, 10) is True                             # count 2 > patient 1
    assert es.best_result == {"ndcg": [0.5, 0.6]}


def test_step_trace_tool_cuts_at_optimizer_bursts(tmp_path):
    """tools/step_trace.py on a synthetic rocprofv3 kernel trace: the window between two bursts of `adam_kernel`
    launches is one step; per-kernel totals, busy time and idle gaps of that window."""
    import subprocess
    import sys
    d = tmp_path / "run" / "1"
    d.mkdir(parents=True)
    rows, t = [], 0
    for step in range(3):
        for name, dur in (("spmm_rows_kernel", 3_000_000), ("bpr_fwd_kernel", 10_000), ("spmm_rows_kernel", 2_000_000)):
            rows.append((t, t + dur, name)); t += dur + 5_000
        for _ in range(2):
            rows.append((t, t + 50_000, "tagrec::adam_kernel(float4*)")); t += 60_000
        t += 3_000_000                                   # host gap before the next step (> 2 ms: ends the burst)
    with open(d / "1_kernel_trace.csv", "w") as f:
        f.write("Kind,Agent_Id,Kernel_Name,Start_Timestamp,End_Timestamp\n")
        for s, e, n in rows:
            f.write(f'KERNEL_DISPATCH,0,"{n}",{s},{e}\n')
    out = tmp_path / "step.csv"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_trace.py"), str(tmp_path / "run"), str(out)],
                       capture_output=True, text=True, check=True)
    assert "3 optimizer bursts" in r.stdout and "step window 8.14 ms" in r.stdout and ", 5 launches" in r.stdout
    table = {row[0]: row for row in __import__("csv").reader(open(out))}
    assert float(table["spmm_rows_kernel"][1]) == 5.0 and int(table["spmm_rows_kernel"][2]) == 2
    assert int(table["tagrec::adam_kernel(float4*)"][2]) == 2


def _check_device_adjacency(fx, use_tag, norm, device):
    """graph.block_adjacency_device + normalise_device (the builder of every synthetic-graph headline number, and the
    on-device ingest of [E,2] / [A,3] tensors) against the reference's own `creat_adj` output: indices bit for bit,
    values within 4 ulp.  Why not bit for bit: the reference evaluates d = rowsum ** -0.5 with numpy's fp32 `power`,
    whose SIMD implementation is itself 1-2 ulp off the correctly rounded value for a quarter of the small integers, and
    every bi_norm entry multiplies two such factors; the device path rounds the fp64 power once.  (The HOST path,
    `normalise_host`, calls numpy as the reference does and is bit-exact: test_host_adjacency_bit_exact_vs_reference.)"""
    ui, ut, it = blocks_from_fixture(fx, use_tag)
    dev = torch.device(device)
    t = lambda b: (torch.from_numpy(np.asarray(b[0])).to(dev), torch.from_numpy(np.asarray(b[1])).to(dev), None, b[3])
    rowptr, col, val, n = G.block_adjacency_device(t(ui), t(ut) if use_tag else None, t(it) if use_tag else None)
    rowptr, col, val = G.normalise_device(rowptr, col, val, n, norm)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), rowptr[1:] - rowptr[:-1]).cpu().numpy()
    want_idx, want_val = fx[f"{norm}_{use_tag}_idx"], fx[f"{norm}_{use_tag}_val"]
    assert np.array_equal(np.stack([rows, col.cpu().numpy().astype(np.int64)]), want_idx)
    got = val.cpu().numpy()
    assert got.dtype == np.float32 and np.all(np.abs(got - want_val) <= 4 * np.spacing(np.abs(want_val)))
    if not use_tag:                       # the two-block wrapper the benchmarks call
        rp2, c2, v2, n2 = G.bipartite_norm_device(t(ui)[0], t(ui)[1], ui[3][0], ui[3][1], norm)
        assert n2 == n and torch.equal(rp2, rowptr) and torch.equal(c2, col) and torch.equal(v2, val)


@pytest.mark.parametrize("use_tag", [0, 1])
@pytest.mark.parametrize("norm", ["bi_norm", "ngcf", "si_norm", "si_norm_self", "plain"])
def test_device_adjacency_builder_on_cpu_tensors_vs_reference(golden, use_tag, norm):
    _check_device_adjacency(golden("adj_toy"), use_tag, norm, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("use_tag", [0, 1])
@pytest.mark.parametrize("norm", ["bi_norm", "ngcf", "si_norm", "si_norm_self", "plain"])
def test_device_adjacency_builder_on_gpu_vs_reference(golden, use_tag, norm):
    _check_device_adjacency(golden("adj_toy"), use_tag, norm, "cuda:0")


def test_bench_refuses_more_ranks_than_gpus_without_touching_the_gpu():
    """`python bench.py --gpus N` starts its own ranks; with fewer GPUs than ranks it must say so and exit non-zero
    (the parent only counts devices)."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and "one rank per GPU" in line["error"]
