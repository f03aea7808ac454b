import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tagrec_amd as T
from tagrec_amd import tgcn as TG, _lib
DEV = torch.device("cuda:0")
D, Dout, n = 128, 128, 257
torch.manual_seed(D * 3 + Dout)
A, C, V = 32, 32, 8
ts = [torch.randn(n, D) * 0.5 for _ in range(3)]
prm = [torch.randn(D, A) * 0.2, torch.randn(1, A) * 0.1, torch.randn(1, A), torch.randn(C, 1, 3, 1) * 0.5,
       torch.randn(V, 1, 1, D) * 0.2, torch.randn(V, 1, 2, D) * 0.2, torch.randn(V, 1, 3, D) * 0.2,
       torch.randn(C * D + 6 * V, Dout) * 0.05, torch.randn(1, Dout) * 0.1]
up = torch.randn(n, Dout)
# fp64 reference with intermediates
U, q, p, wb, w1, w2, w3, Wf, bf = [x.double() for x in prm]
st = torch.stack([t.double() for t in ts], 1)
S = (st @ U + q).requires_grad_()
s = torch.relu(S) @ p.t()
b = torch.softmax(s, dim=1)
e3 = (b * st).requires_grad_()
e3r = e3
w = wb[:, 0, :, 0]
bitpre = torch.einsum("cj,njd->ncd", w, e3r)
bit = torch.relu(bitpre).reshape(n, -1)
flat = e3r.reshape(n, 3 * D)
v1 = torch.relu(torch.einsum("nhd,cd->nch", e3r, w1[:, 0, 0, :])).reshape(n, -1)
v2 = torch.relu(torch.stack([flat[:, :2 * D] @ w2.reshape(-1, 2 * D).t(), flat[:, D:] @ w2.reshape(-1, 2 * D).t()], dim=2)).reshape(n, -1)
v3 = torch.relu(flat @ w3.reshape(-1, 3 * D).t())
y = torch.cat([bit, v1, v2, v3], 1)
outpre = y @ Wf + bf
out = torch.relu(outpre)
(out * up.double()).sum().backward()
de3_ref = e3.grad
print("node 208: min |outpre|", float(outpre[208].abs().min()), " min|bitpre|", float(bitpre[208].abs().min()))
# kernel forward + backward raw
g = [t.to(DEV).contiguous() for t in ts]
P = [x.to(DEV).contiguous() for x in prm]
Ug, qg, pg, wbg, w1g, w2g, w3g, Wfg, bfg = P[0], P[1].reshape(-1), P[2].reshape(-1), P[3].reshape(C, 3).contiguous(), P[4].reshape(V, -1).contiguous(), P[5].reshape(V, -1).contiguous(), P[6].reshape(V, -1).contiguous(), P[7], P[8].reshape(-1)
lib = _lib.load()
outk = torch.empty(n, Dout, device=DEV); bw = torch.empty(n, 3, device=DEV)
_lib.check(lib.tagrec_tgcn_fuse_fwd_f32(_lib.ptr(g[0]), _lib.ptr(g[1]), _lib.ptr(g[2]), n, D, Dout, A, C, V, _lib.ptr(Ug), _lib.ptr(qg), _lib.ptr(pg), _lib.ptr(wbg), _lib.ptr(w1g), _lib.ptr(w2g), _lib.ptr(w3g), _lib.ptr(Wfg), _lib.ptr(bfg), _lib.ptr(bw), _lib.ptr(outk), _lib.stream_ptr()))
print("fwd out max diff", float((outk.cpu().double() - out.detach()).abs().max()), "mask mismatches", int(((outk.cpu() > 0) != (out.detach() > 0)).sum()))
dts = [torch.empty(n, D, device=DEV) for _ in range(3)]
yvec = torch.empty(n, 48, device=DEV); dfeat = torch.empty(n, 48, device=DEV); dS = torch.empty(n, 96, device=DEV)
small = torch.empty(3 * C + 2 * A, device=DEV)
wsn = lib.tagrec_tgcn_fuse_bwd_workspace(Dout); ws = torch.empty(wsn, device=DEV)
dout = up.to(DEV).contiguous()
_lib.check(lib.tagrec_tgcn_fuse_bwd_f32(_lib.ptr(g[0]), _lib.ptr(g[1]), _lib.ptr(g[2]), n, D, Dout, A, C, V, _lib.ptr(Ug), _lib.ptr(qg), _lib.ptr(pg), _lib.ptr(wbg), _lib.ptr(w1g), _lib.ptr(w2g), _lib.ptr(w3g), _lib.ptr(Wfg), _lib.ptr(outk), _lib.ptr(dout), _lib.ptr(dts[0]), _lib.ptr(dts[1]), _lib.ptr(dts[2]), _lib.ptr(yvec), _lib.ptr(dfeat), _lib.ptr(dS), _lib.ptr(small), _lib.ptr(ws), wsn, _lib.stream_ptr()))
torch.cuda.synchronize()
dS_ref = S.grad.reshape(n, 96)
err = (dS.cpu().double() - dS_ref).abs()
print("dS max err", float(err.max()), "row208", float(err[208].max()), "rows bad", torch.nonzero(err.max(1).values > 1e-4).flatten().tolist())
print("dS kernel row208[:8]", dS[208, :8].cpu().numpy(), "ref", dS_ref[208, :8].numpy())
print("bw kernel", bw[208].cpu().numpy(), "ref", b[208, :, 0].detach().numpy())
# reconstruct de3 from the kernel: dt_j = bw_j de3_j + U dS_j  =>  de3_j = (dt_j - dS_j U^T) / bw_j
for j in range(3):
    de3_k = (dts[j].cpu().double() - dS.cpu().double()[:, j * A:(j + 1) * A] @ U.t()) / bw[:, j:j + 1].cpu().double()
    e = (de3_k - de3_ref[:, j]).abs()
    print("de3 j", j, "max err", float(e.max()), "row208", float(e[208].max()), "bad rows", torch.nonzero(e.max(1).values > 1e-3).flatten().tolist()[:10],
          "bad cols row208", torch.nonzero(e[208] > 1e-3).flatten().tolist()[:20])
dwb_ref = None
