import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tagrec_amd as T
from tagrec_amd import tgcn as TG
DEV = torch.device("cuda:0")
D, Dout, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(D * 3 + Dout)
A, C, V = 32, 32, 8
ts = [torch.randn(n, D) * 0.5 for _ in range(3)]
prm = [torch.randn(D, A) * 0.2, torch.randn(1, A) * 0.1, torch.randn(1, A), torch.randn(C, 1, 3, 1) * 0.5,
       torch.randn(V, 1, 1, D) * 0.2, torch.randn(V, 1, 2, D) * 0.2, torch.randn(V, 1, 3, D) * 0.2,
       torch.randn(C * D + 6 * V, Dout) * 0.05, torch.randn(1, Dout) * 0.1]
up = torch.randn(n, Dout)
rt = [t.double().requires_grad_() for t in ts]
rp = [x.double().requires_grad_() for x in prm]
want = TG._dense_block(torch.stack(rt, dim=1), *rp)
(want * up.double()).sum().backward()
gt = [t.to(DEV).requires_grad_() for t in ts]
gp = [x.to(DEV).requires_grad_() for x in prm]
U, q, p, wb, w1, w2, w3, Wf, bf = gp
got = TG._FusedDense.apply(gt[0], gt[1], gt[2], U, q.reshape(-1), p.reshape(-1), wb.reshape(C, 3), w1.reshape(V, -1),
                           w2.reshape(V, -1), w3.reshape(V, -1), Wf, bf.reshape(-1), 50)
(got * up.to(DEV)).sum().backward()
names = ["t0", "t1", "t2", "U", "q", "p", "wb", "w1", "w2", "w3", "Wf", "bf"]
for name, a, b in zip(names, gt + gp, rt + rp):
    g, w = a.grad.cpu().numpy().reshape(b.shape), b.grad.float().numpy()
    err = np.abs(g - w)
    bad = err > 2e-3 * np.abs(w) + 2e-6 * np.abs(w).max()
    print(name, g.shape, "max err", err.max(), "bad", int(bad.sum()))
    if bad.sum() and g.ndim == 2:
        rows, cols = np.nonzero(bad)
        print("   rows", sorted(set(rows.tolist()))[:20], "n_rows", len(set(rows.tolist())), "cols", sorted(set(cols.tolist()))[:40])
# inspect the suspicious node: near-zero pre-activations?
with torch.no_grad():
    st = torch.stack([t.double() for t in ts], 1)
    S = st @ prm[0].double() + prm[1].double()
    for node in (208, 0, 100):
        print("node", node, "min|S|", float(S[node].abs().min()), "argmin", int(S[node].abs().argmin()))
f32 = [t.to(DEV).requires_grad_() for t in ts]
p32 = [x.to(DEV).requires_grad_() for x in prm]
o32 = TG._dense_block(torch.stack(f32, dim=1), *p32)
with torch.no_grad():
    print("fwd diff fused vs torch32 row208", float((got.detach()[208] - o32.detach()[208]).abs().max()))
(o32 * up.to(DEV)).sum().backward()
for k in range(3):
    d = (f32[k].grad - gt[k].grad).abs()
    print("t%d torch32 vs fused: max" % k, float(d.max()), "row208", float(d[208].max()), "torch32 vs f64 row208",
          float((f32[k].grad[208].cpu() - rt[k].grad[208].float()).abs().max()))
