"""GPU: the sharded LightGCN models with the REAL HIP kernels on two ranks.  A one-GPU box cannot host two RCCL ranks
(one communicator rank per device), so both processes use cuda:0 and exchange through gloo; what is exercised is the
kernel chain on column slices / row shards together with real inter-process reductions.  RCCL itself is exercised at
world size 1 in test_gpu_lightgcn.py and by `bench.py --gpus N` on a multi-GPU node."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from conftest import ROOT, load_golden, blocks_from_fixture


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, kind):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tagrec_amd as T
    from tagrec_amd import dist as TD
    from oracle import adj as oadj
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        fx = load_golden("lightgcn_toy")
        csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, 1)), "bi_norm")
        cfg = T.get_config("lightgcn", use_tag=True, dim_layer_list=[64, 64], reg=float(fx["reg"]), device=dev)
        ds = T.synth.Dataset()
        ds.num = {"user": int(fx["n_user"]), "item": int(fx["n_item"]), "tag": int(fx["n_tag"])}
        args = (ds, cfg, torch.from_numpy(csr.rowptr).to(dev), torch.from_numpy(csr.col).to(dev), torch.from_numpy(csr.val).to(dev),
                csr.shape[0])
        full = torch.cat([torch.from_numpy(fx[f"init.embed.{t}"]) for t in range(3)]).to(dev)
        if kind == "feature":
            m = TD.FeatureShardedLightGCN(*args)
            lo = rank * m.dim_local
            with torch.no_grad():
                m.table.copy_(full[:, lo:lo + m.dim_local])
        else:
            m = TD.ShardedLightGCN(*args)
            with torch.no_grad():
                m.table.zero_()
                hi = min(m.hi, full.shape[0])
                m.table[:hi - m.lo] = full[m.lo:hi]
        opt = T.Adam(m.parameters(), lr=0.01)
        losses = []
        for b in fx["batches"][:3]:
            lossx = m.loss(torch.from_numpy(b).to(dev))
            losses.append([float(x) for x in lossx])
            opt.zero_grad()
            sum(lossx).backward()
            opt.step()
        table = m.gathered_table()
        if rank == 0:
            np.savez(os.path.join(out_dir, f"{kind}.npz"), losses=np.array(losses), table=table.cpu().numpy()[:full.shape[0]])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["feature", "row"])
def test_two_ranks_real_kernels(tmp_path, golden, kind):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), kind), nprocs=2, join=True)
    got = np.load(tmp_path / f"{kind}.npz")
    fx = golden("lightgcn_toy")
    np.testing.assert_allclose(got["losses"][0], fx["loss_parts"], rtol=1e-5)
    np.testing.assert_allclose(got["losses"].sum(1), fx["step3.losses"], rtol=2e-5)
    want = np.concatenate([fx[f"step3.embed.{t}"] for t in range(3)])
    assert np.abs(got["table"] - want).max() <= 2e-4
