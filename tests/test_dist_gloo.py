"""World-size-2 (and 3: uneven shards) run of the row-sharded LightGCN host logic on CPU ranks over gloo.

The local kernels are replaced by a CPU test double built from torch ops (`CpuOps` below) -- this
checks the partitioning, the all-gather / all-reduce plumbing, the batch-row exchange and the
gradient scatter of tagrec_amd/dist.py against the single-process oracle.  The HIP kernels
themselves are checked in the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, blocks_from_fixture


class CpuOps:
    """Test double for tagrec_amd.dist.HipOps (same call signatures, torch CPU arithmetic)."""

    def make_graph(self, rowptr, col, val, shape):
        return torch.sparse_csr_tensor(rowptr, col.long(), val, size=shape)

    @staticmethod
    def _nb(x_raw, inv, dz):
        z = x_raw * inv[:, None]
        dot = (z * dz).sum(1, keepdim=True)
        dot = torch.where(inv[:, None] >= 1e12, torch.zeros_like(dot), dot)
        return inv[:, None] * (dz - z * dot)

    # -- row-sharded tables (same contracts as HipOps) ---------------------------------------------
    row_sparse_backward = True

    def row_block(self, rowptr, col, val, lo, hi, n_cols):
        a, b = int(rowptr[lo]), int(rowptr[hi])
        return torch.sparse_csr_tensor(rowptr[lo:hi + 1] - a, col[a:b].long(), val[a:b], size=(hi - lo, n_cols))

    def mark_cols(self, g, rows, flags):
        flags[(g.to_dense() != 0)[rows].any(0)] = 1

    def spmm_listed(self, g, rows, x, out):
        w = g.to_dense()[rows]
        used = (w != 0).any(0)                      # rows of x no listed row references may be unwritten
        out.copy_(w[:, used] @ x[used])

    @staticmethod
    def _keep(row_mask, n):
        return torch.ones(n, dtype=torch.bool) if row_mask is None else row_mask.bool()

    @staticmethod
    def _flagged(g_in, in_flags, in_count):
        # with a count the kernels MAY ignore the flags (they do once these cover 4/5 of the rows): the caller must keep
        # rows flagged zero at zero -- reading everything here checks that it does.  Without a count the flags are always
        # consulted and rows flagged zero may hold anything.
        if in_flags is None or in_count is not None:
            return g_in
        return torch.where(in_flags.bool()[:, None], g_in, torch.zeros_like(g_in))

    def layer_fwd(self, g, xf, y, inv, acc, s, row_mask=None):
        full = g @ xf
        den = full.norm(dim=1).clamp_min(1e-12)
        k = self._keep(row_mask, y.shape[0])
        y[k] = full[k]
        inv[k] = (1.0 / den)[k]
        if acc is not None:
            acc[k] += (s * full / den[:, None])[k]

    def _nb_rows(self, x_raw, inv, dz, s, dz_flags):
        if dz_flags is None:
            return self._nb(x_raw, inv, s * dz)
        f = dz_flags.bool()
        out = torch.zeros_like(dz)
        out[f] = self._nb(x_raw[f], inv[f], s * dz[f])
        return out

    def layer_bwd(self, g, g_in, in_flags, in_count, x_raw, inv, dz, s, out, out_flags, row_mask=None, dz_flags=None):
        res = g @ self._flagged(g_in, in_flags, in_count) + self._nb_rows(x_raw, inv, dz, s, dz_flags)
        k = self._keep(row_mask, out.shape[0])
        out[k] = res[k]
        if out_flags is not None:
            out_flags[k] = (res[k] != 0).any(1).to(torch.uint8)

    def last_hop(self, g, g_in, in_flags, in_count, b, s, out, b_flags=None, row_mask=None):
        bb = b if b_flags is None else torch.where(b_flags.bool()[:, None], b, torch.zeros_like(b))
        k = self._keep(row_mask, out.shape[0])
        out[k] = (g @ self._flagged(g_in, in_flags, in_count) + s * bb)[k]

    def last_hop_adam(self, g, g_in, in_flags, in_count, b, s, b_flags, p, m, v, lr, betas, eps, step):
        bb = b if b_flags is None else torch.where(b_flags.bool()[:, None], b, torch.zeros_like(b))
        gr = g @ self._flagged(g_in, in_flags, in_count) + s * bb
        b1, b2 = betas
        m.copy_(m + (1.0 - b1) * (gr - m))                 # torch.optim.Adam, in its operation order
        v.copy_(v * b2 + ((1.0 - b2) * gr) * gr)
        step_size, bc2_sqrt = lr / (1.0 - b1 ** step), (1.0 - b2 ** step) ** 0.5
        p.copy_(p - step_size * (m / (v.sqrt() / bc2_sqrt + eps)))

    def rownorm_fwd(self, x):
        den = x.norm(dim=1).clamp_min(1e-12)
        return x / den[:, None], 1.0 / den

    def rownorm_bwd(self, x_raw, inv, dz, s, out):
        out.copy_(self._nb(x_raw, inv, s * dz))

    def rownorm_bwd_flags(self, x_raw, inv, dz, s, out):
        out.copy_(self._nb(x_raw, inv, s * dz))
        return (out != 0).any(1).to(torch.uint8)

    def bpr_fwd(self, U, I, Ur, Ir, trip, kind):
        u, p, n = U[trip[:, 0]], I[trip[:, 1]], I[trip[:, 2]]
        x = (u * n).sum(1) - (u * p).sum(1)
        loss = torch.nn.functional.softplus(x).mean() if kind == 0 else -torch.nn.functional.logsigmoid(-x).mean()
        reg = 0.5 * (Ur[trip[:, 0]].pow(2).sum() + Ir[trip[:, 1]].pow(2).sum() + Ir[trip[:, 2]].pow(2).sum()) / trip.shape[0]
        return torch.stack([loss, reg]), torch.sigmoid(x)

    def bpr_bwd(self, U, I, Ur, Ir, trip, coef, g, dU, dI, dUr, dIr):
        B = trip.shape[0]
        if dU is not None:
            c = (g[0] * coef / B)[:, None]
            u, p, n = U[trip[:, 0]], I[trip[:, 1]], I[trip[:, 2]]
            dU.index_add_(0, trip[:, 0], c * (n - p))
            dI.index_add_(0, trip[:, 1], -c * u)
            dI.index_add_(0, trip[:, 2], c * u)
        if dUr is not None:
            cr = g[1] / B
            dUr.index_add_(0, trip[:, 0], cr * Ur[trip[:, 0]])
            dIr.index_add_(0, trip[:, 1], cr * Ir[trip[:, 1]])
            dIr.index_add_(0, trip[:, 2], cr * Ir[trip[:, 2]])


    # -- restricted forward (the toy batches touch too many rows for it to switch on; the double keeps the contract)
    restrict_forward = True

    def mark_rows(self, g, rows, flags):
        dense = g.to_dense() != 0
        flags[rows] = 1
        flags[dense[rows].any(0)] = 1
        return flags

    def spmm_ss_rows(self, g, x, y, ss, mask):
        keep = mask.bool()
        full = g @ x
        y[keep] = full[keep]
        ss[keep] = (full[keep] ** 2).sum(1)

    # -- NGCF dense block (same contract as tagrec_amd.ngcf.dense_forward / dense_backward)
    def ngcf_dense_fwd(self, nei, x, w1p, w2p, xp, inv, z_slot, ldz, row_mask=None):
        lr = torch.nn.functional.leaky_relu
        k = self._keep(row_mask, x.shape[0])              # rows outside the mask: neither read nor written
        v = lr((nei[k] + x[k]) @ w1p, 0.2) + lr((nei[k] * x[k]) @ w2p, 0.2)
        den = v.norm(dim=1).clamp_min(1e-12)
        xp[k] = v
        inv[k] = 1.0 / den
        if z_slot is not None:
            z_slot[k, :v.shape[1]] = v / den[:, None]

    def ngcf_dense_bwd(self, dxp, nei, x, w1p, w2p, norm, row_mask=None, dz_flags=None):
        xp, inv, dz, _ = norm
        k = self._keep(row_mask, x.shape[0])
        nan = float("nan")
        d_nei, d_xd = torch.full_like(x, nan), torch.full_like(x, nan)      # rows outside the mask stay unwritten
        dzk = dz[k, :xp.shape[1]]
        if dz_flags is not None:                          # rows flagged zero may hold anything: not read
            dzk = torch.where(dz_flags.bool()[k][:, None], dzk, torch.zeros_like(dzk))
            xpk = torch.where(dz_flags.bool()[k][:, None], xp[k], torch.ones_like(xp[k]))
            invk = torch.where(dz_flags.bool()[k], inv[k], torch.ones_like(inv[k]))
        else:
            xpk, invk = xp[k], inv[k]
        gx = self._nb(xpk, invk, dzk)
        if dxp is not None:
            gx = gx + dxp[k]
        a1, a2 = nei[k] + x[k], nei[k] * x[k]
        slope = lambda p: torch.where(p > 0, torch.ones_like(p), torch.full_like(p, 0.2))
        dp1, dp2 = gx * slope(a1 @ w1p), gx * slope(a2 @ w2p)
        da1, da2 = dp1 @ w1p.t(), dp2 @ w2p.t()
        d_nei[k] = da1 + da2 * x[k]
        d_xd[k] = da1 + da2 * nei[k]
        return d_nei, d_xd, a1.t() @ dp1, a2.t() @ dp2

    # -- column-sharded tables
    def spmm_axpy(self, g, g_in, b, s, out):
        out.copy_(g @ g_in + s * b)

    def spmm_plain(self, g, x, y, row_mask=None):
        k = self._keep(row_mask, y.shape[0])
        y[k] = (g @ x)[k]

    def spmm_flags(self, g, g_in, in_flags, in_count, out, out_flags, row_mask=None):
        res = g @ self._flagged(g_in, in_flags, in_count)
        k = self._keep(row_mask, out.shape[0])
        out[k] = res[k]
        if out_flags is not None:
            out_flags[k] = (res[k] != 0).any(1).to(torch.uint8)

    def spmm_ss(self, g, x, y, ss):
        y.copy_(g @ x)
        ss.copy_((y * y).sum(1))

    def spmm_normbwd_dot(self, g, g_in, x_raw, inv, dz, dot, s, out):
        d = torch.where(inv >= 1e12, torch.zeros_like(dot), dot)
        out.copy_(g @ g_in + inv[:, None] * (s * dz - x_raw * (inv * d)[:, None]))

    def row_scale_acc(self, y, inv, s, acc):
        acc.add_(s * y * inv[:, None])

    def row_dot(self, x, inv, dz, s, out):
        out.copy_(inv * s * (x * dz).sum(1))

    def rownorm_bwd_dot(self, x, inv, dz, dot, s, out):
        d = torch.where(inv >= 1e12, torch.zeros_like(dot), dot)
        out.copy_(inv[:, None] * (s * dz - x * (inv * d)[:, None]))

    def bpr_dots(self, U, I, Ur, Ir, trip):
        u, p, n = U[trip[:, 0]], I[trip[:, 1]], I[trip[:, 2]]
        reg = 0.5 * (Ur[trip[:, 0]].pow(2).sum(1) + Ir[trip[:, 1]].pow(2).sum(1) + Ir[trip[:, 2]].pow(2).sum(1))
        return torch.stack([(u * p).sum(1), (u * n).sum(1), reg], dim=1).contiguous()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _feature_worker(rank, world, port, out_dir, n_layer, restrict):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tagrec_amd as T
    from tagrec_amd import dist as TD
    from oracle import adj as oadj
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        fx = load_golden("lightgcn_toy")
        csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, 1)), "bi_norm")
        cfg = T.get_config("lightgcn", use_tag=True, dim_layer_list=[64] * n_layer, reg=float(fx["reg"]), device="cpu")
        ds = T.synth.Dataset()
        ds.num = {"user": int(fx["n_user"]), "item": int(fx["n_item"]), "tag": int(fx["n_tag"])}
        m = TD.FeatureShardedLightGCN(ds, cfg, torch.from_numpy(csr.rowptr), torch.from_numpy(csr.col),
                                      torch.from_numpy(csr.val), csr.shape[0], ops=CpuOps())
        m.restrict_forward = restrict
        m.restrict_min_ratio = 0                  # the toy batches touch most rows: force the restricted step when asked
        full = torch.cat([torch.from_numpy(fx[f"init.embed.{t}"]) for t in range(3)])
        lo = rank * m.dim_local
        with torch.no_grad():
            m.table.copy_(full[:, lo:lo + m.dim_local])
        opt = torch.optim.Adam(m.parameters(), lr=0.01)
        losses = []
        for b in fx["batches"][:3]:
            lossx = m.loss(torch.from_numpy(b))
            losses.append([float(x) for x in lossx])
            opt.zero_grad()
            sum(lossx).backward()
            if len(losses) == 1:
                parts = [torch.empty_like(m.table.grad) for _ in range(world)]
                dist.all_gather(parts, m.table.grad.contiguous())
                grad0 = torch.cat(parts, dim=1)
            opt.step()
        table = m.gathered_table()
        if rank == 0:
            np.savez(os.path.join(out_dir, "feat.npz"), losses=np.array(losses), grad0=grad0.numpy(), table=table.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_layer,restrict", [(2, 2, False), (4, 2, False), (2, 2, True), (4, 2, True), (2, 3, True),
                                                     (4, 1, True), (8, 3, True)])
def test_feature_sharded_lightgcn_matches_single_process(tmp_path, golden, world, n_layer, restrict):
    """Column-sharded tables: per-column products; all-rows step = all-reduced row norms / row dots / triplet scores over
    [L, N]; restricted step = the same quantities on the 3 B batch rows only."""
    port = _free_port()
    mp.spawn(_feature_worker, args=(world, port, str(tmp_path), n_layer, restrict), nprocs=world, join=True)
    got = np.load(tmp_path / "feat.npz")
    fx = golden("lightgcn_toy")
    if n_layer == 2:
        np.testing.assert_allclose(got["losses"][0], fx["loss_parts"], rtol=1e-5)
        np.testing.assert_allclose(got["losses"].sum(1), fx["step3.losses"], rtol=1e-5)
        want_g = np.concatenate([fx[f"grad.embed.{t}"] for t in range(3)])
        np.testing.assert_allclose(got["grad0"], want_g, rtol=1e-3, atol=1e-8)
        want_t = np.concatenate([fx[f"step3.embed.{t}"] for t in range(3)])
        assert np.abs(got["table"] - want_t).max() <= 2e-4
    losses, grad0, table, _ = _single_process(fx, n_layer)
    np.testing.assert_allclose(got["losses"], losses, rtol=1e-5)
    np.testing.assert_allclose(got["grad0"], grad0, rtol=1e-3, atol=1e-8)
    assert np.abs(got["table"] - table).max() <= 2e-4


def _row_model(rank, world, fx, n_layer, n_chunks, restrict, norm="bi_norm", all_gather="collective"):
    import tagrec_amd as T
    from tagrec_amd import dist as TD
    from oracle import adj as oadj
    csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, 1)), norm)
    cfg = T.get_config("lightgcn", use_tag=True, dim_layer_list=[64] * n_layer, reg=float(fx["reg"]), device="cpu",
                       norm_type=norm, all_gather=all_gather)
    ds = T.synth.Dataset()
    ds.num = {"user": int(fx["n_user"]), "item": int(fx["n_item"]), "tag": int(fx["n_tag"])}
    torch.manual_seed(2020)
    m = TD.ShardedLightGCN(ds, cfg, torch.from_numpy(csr.rowptr), torch.from_numpy(csr.col),
                           torch.from_numpy(csr.val), csr.shape[0], ops=CpuOps(), n_chunks=n_chunks)
    m.restrict_forward = restrict
    m.restrict_min_ratio = 0                  # the toy batches touch most rows: force the restricted step when asked
    # load the fixture's init so the comparison does not depend on the RNG
    full = torch.cat([torch.from_numpy(fx[f"init.embed.{t}"]) for t in range(3)])
    with torch.no_grad():
        m.table.zero_()
        real_hi = min(m.hi, full.shape[0])
        if real_hi > m.lo:
            m.table[:real_hi - m.lo] = full[m.lo:real_hi]
    return m, csr, full


def _worker(rank, world, port, out_dir, n_layer, n_chunks, restrict, all_gather="collective", tag="row"):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        fx = load_golden("lightgcn_toy")
        m, csr, full = _row_model(rank, world, fx, n_layer, n_chunks, restrict, all_gather=all_gather)
        opt = torch.optim.Adam(m.parameters(), lr=0.01)
        losses = []
        for b in fx["batches"][:3]:
            lossx = m.loss(torch.from_numpy(b))
            losses.append([float(x) for x in lossx])
            opt.zero_grad()
            sum(lossx).backward()
            if len(losses) == 1:
                grad0 = m.all_gather(m.table.grad)[:full.shape[0]].clone()
            opt.step()
        table = m.gathered_table()
        u_out, i_out = m.forward()
        if rank == 0:
            np.savez(os.path.join(out_dir, f"{tag}.npz"), losses=np.array(losses), grad0=grad0.numpy(),
                     table=table.numpy(), u_out=u_out.numpy(), i_out=i_out.numpy())
    finally:
        dist.destroy_process_group()


def _single_process(fx, n_layer):
    """The same three steps through the CPU oracle (one process, whole table)."""
    from oracle import adj as oadj, models as om
    csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, 1)), "bi_norm")
    A = om.csr_to_torch(csr)
    tabs = [torch.from_numpy(fx[f"init.embed.{t}"]).clone().requires_grad_() for t in range(3)]
    opt = torch.optim.Adam(tabs, lr=0.01)
    losses, grad0 = [], None
    for b in fx["batches"][:3]:
        lossx = om.lightgcn_loss(tabs, A, n_layer, torch.from_numpy(b), float(fx["reg"]), "softplus")
        losses.append([float(x) for x in lossx])
        opt.zero_grad()
        sum(lossx).backward()
        if grad0 is None:
            grad0 = torch.cat([t.grad for t in tabs]).numpy().copy()
        opt.step()
    with torch.no_grad():
        nums = [t.shape[0] for t in tabs]
        outs = torch.split(om.lightgcn_propagate(torch.cat(tabs), A, n_layer), nums, dim=0)
    return np.array(losses), grad0, torch.cat(tabs).detach().numpy(), outs


# (world, layers, row blocks per shard, restricted step): uneven shards (3), every depth the restricted step
# distinguishes (1: push only, 2: masked + push, 3: full + masked + push), pipelined blocks, and the all-rows step
@pytest.mark.parametrize("world,n_layer,n_chunks,restrict", [
    (2, 2, 1, False), (3, 2, 2, False), (2, 2, 1, True), (3, 2, 3, True),
    (2, 3, 2, True), (2, 1, 1, True), (2, 3, 1, False), (4, 3, 2, True)])
def test_sharded_lightgcn_matches_single_process(tmp_path, golden, world, n_layer, n_chunks, restrict):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), n_layer, n_chunks, restrict), nprocs=world, join=True)
    got = np.load(tmp_path / "row.npz")
    fx = golden("lightgcn_toy")
    if n_layer == 2:                                      # the depth the reference fixture was captured at
        np.testing.assert_allclose(got["losses"][0], fx["loss_parts"], rtol=1e-5)
        np.testing.assert_allclose(got["losses"].sum(1), fx["step3.losses"], rtol=1e-5)
        want_g = np.concatenate([fx[f"grad.embed.{t}"] for t in range(3)])
        np.testing.assert_allclose(got["grad0"], want_g, rtol=1e-3, atol=1e-8)
        want_t = np.concatenate([fx[f"step3.embed.{t}"] for t in range(3)])
        assert np.abs(got["table"] - want_t).max() <= 2e-4
    losses, grad0, table, outs = _single_process(fx, n_layer)
    np.testing.assert_allclose(got["losses"], losses, rtol=1e-5)
    np.testing.assert_allclose(got["grad0"], grad0, rtol=1e-3, atol=1e-8)
    assert np.abs(got["table"] - table).max() <= 2e-4
    np.testing.assert_allclose(got["u_out"], outs[0].numpy(), rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(got["i_out"], outs[1].numpy(), rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("world,n_layer,n_chunks,restrict", [(2, 3, 2, True), (3, 2, 3, False), (4, 3, 1, True)])
def test_direct_all_gather_is_bit_identical_to_the_collective(tmp_path, world, n_layer, n_chunks, restrict):
    """config["all_gather"] = "direct": every block goes to every peer by one send / receive pair per peer posted together
    (dist.batch_isend_irecv; over RCCL one grouped ncclSend / ncclRecv launch = each GPU pushes its block over the xGMI link
    it shares with each peer) instead of all_gather_into_tensor.  Same destination layout, a pure copy: losses,
    gradient, tables after three Adam steps and the propagated outputs are BIT-identical to the collective's, for even and
    uneven shards, one and several row blocks, restricted and all-rows steps."""
    res = {}
    for mode in ("collective", "direct"):
        port = _free_port()
        mp.spawn(_worker, args=(world, port, str(tmp_path), n_layer, n_chunks, restrict, mode, mode), nprocs=world, join=True)
        res[mode] = dict(np.load(tmp_path / f"{mode}.npz"))
    for k in res["collective"]:
        assert np.array_equal(res["collective"][k], res["direct"][k]), k


def _asym_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tagrec_amd._lib import TagrecError
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        fx = load_golden("lightgcn_toy")
        try:
            _row_model(rank, world, fx, 2, 1, True, norm="ngcf")
            msg = "no error"
        except TagrecError as e:
            msg = str(e)
        if rank == 0:
            with open(os.path.join(out_dir, "msg.txt"), "w") as f:
                f.write(msg)
    finally:
        dist.destroy_process_group()


def test_sharded_lightgcn_rejects_asymmetric_normalisation(tmp_path):
    """D^-1 A + I is not symmetric: the row-sharded backward (pull over A[rows_g, :]) would be silently wrong."""
    port = _free_port()
    mp.spawn(_asym_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert "not symmetric" in open(tmp_path / "msg.txt").read()


def _ngcf_worker(rank, world, port, out_dir, n_chunks, restricted=False):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tagrec_amd as T
    from tagrec_amd import dist as TD
    from oracle import adj as oadj
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        fx = load_golden("ngcf_toy")
        csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, 1)), "ngcf")
        cfg = T.get_config("ngcf", use_tag=True, dim_layer_list=[int(v) for v in fx["layers"]], dim_latent=int(fx["D"]),
                           reg=float(fx["reg"]), device="cpu")
        ds = T.synth.Dataset()
        ds.num = {"user": int(fx["n_user"]), "item": int(fx["n_item"]), "tag": int(fx["n_tag"])}
        torch.manual_seed(2020)
        m = TD.ShardedNGCF(ds, cfg, torch.from_numpy(csr.rowptr), torch.from_numpy(csr.col), torch.from_numpy(csr.val),
                           csr.shape[0], ops=CpuOps(), n_chunks=n_chunks)
        m.restrict_min_ratio = 0 if restricted else 10 ** 9      # the toy batch touches most rows: force either path
        full = torch.cat([torch.from_numpy(fx[f"init.embed.{t}"]) for t in range(3)])
        with torch.no_grad():
            m.table.zero_()
            hi = min(m.hi, full.shape[0])
            if hi > m.lo:
                m.table[:hi - m.lo] = full[m.lo:hi]
            for k, p in m.mat.items():
                p.copy_(torch.from_numpy(fx["init.mat." + k]))
        opt = torch.optim.Adam(m.parameters(), lr=0.01)
        losses, grads = [], {}
        for b in fx["batches"][:3]:
            lossx = m.loss(torch.from_numpy(b))
            losses.append([float(x) for x in lossx])
            opt.zero_grad()
            sum(lossx).backward()
            if len(losses) == 1:
                grads = {"table": m.all_gather(m.table.grad)[:full.shape[0]].clone().numpy()}
                grads.update({"mat." + k: p.grad.clone().numpy() for k, p in m.mat.items()})
            opt.step()
        table = m.gathered_table()
        if rank == 0:
            np.savez(os.path.join(out_dir, "ngcf.npz"), losses=np.array(losses), table=table.numpy(),
                     **{"g." + k: v for k, v in grads.items()}, **{"p." + k: p.detach().numpy() for k, p in m.mat.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_chunks,restricted", [(2, 1, False), (3, 2, False), (2, 2, True), (3, 1, True)])
def test_row_sharded_ngcf_matches_reference_fixture(tmp_path, golden, world, n_chunks, restricted):
    """Row-sharded NGCF (non-symmetric D^-1 A + I: the backward multiplies by the rank's rows of A^T; W / b replicated,
    their gradients all-reduced) against the reference's own run: loss parts, every gradient, parameters after 3 steps.
    restricted: the compact restricted step (masked layer below the top, push-form top layer on the batch rows, flagged
    gradient exchange); the double leaves rows outside a mask as NaN, so a read of a row nobody computed fails."""
    port = _free_port()
    mp.spawn(_ngcf_worker, args=(world, port, str(tmp_path), n_chunks, restricted), nprocs=world, join=True)
    got = np.load(tmp_path / "ngcf.npz")
    fx = golden("ngcf_toy")
    np.testing.assert_allclose(got["losses"][0], fx["loss_parts"], rtol=1e-5)
    np.testing.assert_allclose(got["losses"].sum(1), fx["step3.losses"], rtol=1e-5)
    want_g = np.concatenate([fx[f"grad.embed.{t}"] for t in range(3)])
    scale = np.abs(want_g).max()
    np.testing.assert_allclose(got["g.table"], want_g, rtol=2e-3, atol=1e-6 * scale)
    for k in [f for f in fx if f.startswith("grad.mat.")]:
        np.testing.assert_allclose(got["g." + k[5:]], fx[k], rtol=2e-3, atol=1e-6 * max(scale, np.abs(fx[k]).max()), err_msg=k)
    want_t = np.concatenate([fx[f"step3.embed.{t}"] for t in range(3)])
    assert np.abs(got["table"] - want_t).max() <= 2e-4
    for k in [f for f in fx if f.startswith("step3.mat.")]:
        assert np.abs(got["p." + k[10:]] - fx[k]).max() <= 2e-4, k


def test_row_partition_layout():
    from tagrec_amd import dist as TD
    p = TD.RowPartition(82, 3, 2)                 # 82 nodes, 3 ranks, 2 blocks per shard: Rc = 14, R = 28, padded 84
    assert (p.rc, p.per, p.n_pad) == (14, 28, 84)
    ids = torch.arange(84)
    g = p.gathered(ids)
    assert sorted(g.tolist()) == list(range(84))  # a permutation
    # block c of rank h sits at [c*G*Rc + h*Rc, ...): what all_gather_into_tensor of that block writes
    for h in range(3):
        for c in range(2):
            src = ids[h * 28 + c * 14: h * 28 + (c + 1) * 14]
            assert g[src].tolist() == list(range(c * 42 + h * 14, c * 42 + (h + 1) * 14))
    assert TD.RowPartition(82, 3, 1).gathered(ids[:82]).tolist() == list(range(82))
    rp = torch.tensor([0, 2, 2, 5, 6])
    col = torch.tensor([1, 3, 0, 1, 2, 0], dtype=torch.int32)
    val = torch.arange(6, dtype=torch.float32)
    trp, tc, tv = TD.transpose_csr(rp, col, val, 4)
    dense = torch.sparse_csr_tensor(rp, col.long(), val, size=(4, 4)).to_dense()
    assert torch.equal(torch.sparse_csr_tensor(trp, tc.long(), tv, size=(4, 4)).to_dense(), dense.t())


def test_shard_helpers():
    from tagrec_amd import dist as TD
    assert TD.shard_rows(82, 2) == (41, 82) and TD.shard_rows(82, 3) == (28, 84) and TD.shard_rows(8, 8) == (1, 8)
    rp = torch.tensor([0, 2, 2, 5, 6])
    col = torch.arange(6, dtype=torch.int32)
    val = torch.arange(6, dtype=torch.float32)
    a, c, v = TD.local_csr(rp, col, val, 3, 6, 3)          # last shard: one real row + two padding rows
    assert a.tolist() == [0, 1, 1, 1] and c.tolist() == [5] and v.tolist() == [5.0]
    a, c, v = TD.local_csr(rp, col, val, 0, 3, 3)
    assert a.tolist() == [0, 2, 2, 5] and c.tolist() == [0, 1, 2, 3, 4]


def _fused_adam_worker(rank, world, port, out_dir, kind):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tagrec_amd as T
    from tagrec_amd import dist as TD
    from oracle import adj as oadj
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        fx = load_golden("lightgcn_toy")
        csr = oadj.normalise(oadj.block_adjacency(*blocks_from_fixture(fx, 1)), "bi_norm")
        cfg = T.get_config("lightgcn", use_tag=True, dim_layer_list=[64, 64, 64], reg=0.0, device="cpu")
        ds = T.synth.Dataset()
        ds.num = {"user": int(fx["n_user"]), "item": int(fx["n_item"]), "tag": int(fx["n_tag"])}
        full = torch.cat([torch.from_numpy(fx[f"init.embed.{t}"]) for t in range(3)])
        tables = []
        for fuse in (False, True):
            args = (ds, cfg, torch.from_numpy(csr.rowptr), torch.from_numpy(csr.col), torch.from_numpy(csr.val), csr.shape[0])
            if kind == "feature":
                m = TD.FeatureShardedLightGCN(*args, ops=CpuOps())
                with torch.no_grad():
                    m.table.copy_(full[:, rank * m.dim_local:(rank + 1) * m.dim_local])
            else:
                m = TD.ShardedLightGCN(*args, ops=CpuOps(), n_chunks=2)
                with torch.no_grad():
                    m.table.zero_()
                    hi = min(m.hi, full.shape[0])
                    if hi > m.lo:
                        m.table[:hi - m.lo] = full[m.lo:hi]
            m.restrict_min_ratio = 0
            if fuse:
                opt = T.Adam(m.parameters(), lr=0.01).fuse_into(m)       # every parameter is fused: no HIP call on the CPU
            else:
                opt = torch.optim.Adam(m.parameters(), lr=0.01)
            for i, b in enumerate(fx["batches"][:3]):
                lossx = m.loss(torch.from_numpy(b))
                opt.zero_grad()
                sum(lossx).backward()
                assert (m.table.grad is None) == fuse
                if kind == "row":
                    # the fused last hop starts the NEXT step's all-gather of X^0 behind each updated row block; the next
                    # loss() consumes it; a hand edit of the table between two steps must be followed by invalidate_prefetch()
                    assert (m._x0_prefetched is not None) == fuse
                opt.step()
                if i == 1:
                    with torch.no_grad():
                        m.table.mul_(1.0 + 1e-3)
                    if kind == "row":
                        m.invalidate_prefetch()
                        assert m._x0_prefetched is None
            tables.append(m.gathered_table())
        if rank == 0:
            np.savez(os.path.join(out_dir, "fused.npz"), a=tables[0].numpy(), b=tables[1].numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("row", 2), ("row", 3), ("feature", 2)])
def test_sharded_step_with_adam_in_the_last_hop(tmp_path, kind, world):
    """`Adam.fuse_into(sharded model)`: the restricted step (reg == 0) applies the shard's update inside its last hop (per
    row block / on the column slice) -- same parameters after three steps as torch.optim.Adam on the gradient."""
    port = _free_port()
    mp.spawn(_fused_adam_worker, args=(world, port, str(tmp_path), kind), nprocs=world, join=True)
    got = np.load(tmp_path / "fused.npz")
    np.testing.assert_allclose(got["b"], got["a"], rtol=0, atol=2e-6)
