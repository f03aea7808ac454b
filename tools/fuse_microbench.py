"""Time the three fused TGCN dense kernels (csrc/tgcn_fuse.hip) on random data: forward, backward-data, backward-weights.
Usage: python tools/fuse_microbench.py [n_nodes] [D]     (needs a GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tagrec_amd as T
from tagrec_amd import tgcn as TG

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.1)
A, C, V = 32, 32, 8
t = [rnd(n, D).requires_grad_() for _ in range(3)]
prm = [rnd(D, A), rnd(A), rnd(A), rnd(C, 3), rnd(V, D), rnd(V, 2 * D), rnd(V, 3 * D), rnd(C * D + 6 * V, D), rnd(D)]
prm = [p.requires_grad_() for p in prm]
TG.timing = {}
for it in range(4):
    out = TG._FusedDense.apply(*t, *prm, 0)
    out.backward(torch.ones_like(out))
torch.cuda.synchronize()
flop = 2 * n * (C * D + 6 * V) * D
for k, v in TG.timing.items():
    ms = [a.elapsed_time(b) for a, b in v][1:]
    m = sum(ms) / len(ms)
    print(f"{k}: {m:.2f} ms  ({flop / m / 1e9:.1f} TFLOP/s of the fusion product)")

# the shader clock while each kernel runs: a one-wave sampler on a second stream, launched first (tagrec_probe_clock)
from tagrec_amd import _lib
lib = _lib.load()
side = torch.cuda.Stream()
for name in ("idle", "fuse_fwd", "fuse_bwd+wf"):
    res = torch.zeros(2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        _lib.check(lib.tagrec_probe_clock(6000 if name != "fuse_bwd+wf" else 20000, _lib.ptr(res), _lib.c_void_p(side.cuda_stream)), "probe_clock")
    if name == "fuse_fwd":
        with torch.no_grad():
            TG._FusedDense.apply(*t, *prm, 0)
    elif name != "idle":
        out = TG._FusedDense.apply(*t, *prm, 0)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            _lib.check(lib.tagrec_probe_clock(20000, _lib.ptr(res), _lib.c_void_p(side.cuda_stream)), "probe_clock")
        out.backward(torch.ones_like(out))
    torch.cuda.synchronize()
    c, tk = res.tolist()
    print(f"shader clock during {name}: {c / tk * 100:.0f} MHz  ({tk / 100:.0f} us sampled)")
