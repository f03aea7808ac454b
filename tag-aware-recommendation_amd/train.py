"""The training step surface of /root/reference/training/basic_train.py, plus the
fused Adam that replaces `torch.optim.Adam` (com.py:14,25,69).

    epoch_training(training_data, loss_func, opt) -> list[float]      basic_train.py:10-30
    GraphedStep / epoch_training(..., graphs={})                       the same step replayed as one HIP graph
    Basic_train(train_data, loss_func, opt, test, args).run(model)    basic_train.py:50-85
    Early_stop                                                         training/early_stop.py:8-41
    Adam(params, lr)   zero_grad()/step()                              torch.optim.Adam defaults
"""
import os
import time

import numpy as np
import torch

from . import _lib
from .config import CFG as _GLOBAL_CFG


class Adam:
    """torch.optim.Adam (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) as one fused
    HIP pass per parameter tensor.  Same zero_grad()/step() protocol, so it drops into
    `epoch_training` where the reference passes a torch optimizer."""

    MULTI_MAX_NUMEL = 1 << 22         # tensors up to this size share one launch (tagrec_adam_multi_f32); 0: one launch each

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        """capturable=True keeps every parameter's step counter in device memory and advances it inside the
        update (tagrec_adam_graph_f32), so `step()` can be captured in a HIP graph and replayed (`GraphedStep`)."""
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("Adam: empty parameter list")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.state = {}
        self.step_count = 0
        self.capturable = bool(capturable)
        self._fused_done = set()          # parameters whose update of the current step was applied inside backward()
        for p in self.params:             # the newest optimizer built over a parameter owns it: a fusion set up by an
            p._tagrec_owner = self        # earlier one is revoked (`fused_optimizer`), the model hands out gradients again

    def fuse_into(self, model):
        """Let `model` apply this optimizer's update of its embedding table inside the last kernel of its backward pass
        (the gradient row is consumed where it is formed: no gradient tensor, no separate Adam launch over the table).
        LightGCN / NGCF: `model.table`, in the epilogue of the last backward product; TGCN: the three node tables
        (`model.fused_tables()`), in the epilogue of the bottom layer's dQ W_2^T product of the BPR phase -- the TransTag
        phase and every other path hand out gradients as usual.
        The zero_grad() / backward() / step() protocol of basic_train.py:19-25 is unchanged: step() then only counts
        the step for that parameter.  One backward() per step().  With capturable=True (models that declare
        `fused_capturable`: LightGCN, NGCF) the step counter and the step-dependent factors live in device memory and
        are advanced inside the fused launch, so the whole step can be captured as a HIP graph (`GraphedStep`).  Models
        without the hook ignore the call.

        CONTRACT: with the fusion on, a training-mode backward() of `model.loss` CHANGES the table (and exp_avg /
        exp_avg_sq) -- also one that is not followed by step() (gradient inspection, clipping); a second such
        backward() before step() raises.  Use an un-fused optimizer for anything but the plain loop."""
        if self.capturable and not getattr(model, "fused_capturable", False):
            raise _lib.TagrecError("Adam.fuse_into: this model's fused update keeps its step counter on the host (capturable=False)")
        if hasattr(model, "set_fused_optimizer"):
            for table in _fused_tables(model):
                if not any(q is table for q in self.params):
                    raise _lib.TagrecError("Adam.fuse_into: the model's table is not one of this optimizer's parameters")
            for table in _fused_tables(model):
                table._tagrec_owner = self
            model.set_fused_optimizer(self)
        return self

    def fused_state(self, p):
        """(m, v, step number of the update about to be applied) for a parameter updated inside backward().  The caller
        launches the update and then calls `fused_commit(p)`: a launch that fails leaves no sticky mark."""
        if id(p) in self._fused_done:
            raise _lib.TagrecError("Adam: a second backward() before step() with a fused update")
        st = self.state.get(id(p))
        if st is None:
            st = self.state[id(p)] = {"m": torch.zeros_like(p.data), "v": torch.zeros_like(p.data), "t": 0}
            if self.capturable:
                st["t_dev"] = torch.zeros(1, dtype=torch.int64, device=p.device)
                st["coef"] = torch.zeros(2, dtype=torch.float32, device=p.device)
        return st["m"], st["v"], st["t"] + 1

    def fused_dev(self, p):
        """(device step counter, device factors) of a capturable optimizer's parameter, else None."""
        st = self.state.get(id(p))
        return (st["t_dev"], st["coef"]) if (self.capturable and st is not None) else None

    def fused_commit(self, p):
        """The fused update of `p` has been launched: step() only counts the step for it."""
        self._fused_done.add(id(p))

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self):
        lib = _lib.load()
        self.step_count += 1
        small = {}                                        # step count -> [(p, g, m, v)] of the tensors updated in one launch
        for p in self.params:
            if id(p) in self._fused_done:             # updated inside backward(): only the step counter moves here
                self._fused_done.discard(id(p))
                self.state[id(p)]["t"] += 1
                if p.grad is None:
                    continue
                raise _lib.TagrecError("Adam: parameter with a fused update also received a gradient tensor")
            if p.grad is None:
                continue
            _lib.require_gpu_tensor(p.data, torch.float32, "Adam parameter")
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            st = self.state.get(id(p))
            if st is None:
                st = self.state[id(p)] = {"m": torch.zeros_like(p.data), "v": torch.zeros_like(p.data), "t": 0}
                if self.capturable:
                    st["t_dev"] = torch.zeros(1, dtype=torch.int64, device=p.device)
                    st["coef"] = torch.zeros(2, dtype=torch.float32, device=p.device)
            st["t"] += 1
            if not self.capturable and p.numel() <= self.MULTI_MAX_NUMEL:
                small.setdefault(st["t"], []).append((p.data, g, st["m"], st["v"]))       # one launch for all of them, below
                continue
            if self.capturable:
                _lib.check(lib.tagrec_adam_graph_f32(_lib.ptr(p.data), _lib.ptr(g), _lib.ptr(st["m"]), _lib.ptr(st["v"]),
                                                     p.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                                     _lib.ptr(st["t_dev"]), _lib.ptr(st["coef"]), _lib.stream_ptr()),
                           "adam_graph")
            else:
                _lib.check(lib.tagrec_adam_f32(_lib.ptr(p.data), _lib.ptr(g), _lib.ptr(st["m"]), _lib.ptr(st["v"]),
                                               p.numel(), self.lr, self.betas[0], self.betas[1], self.eps, st["t"],
                                               _lib.stream_ptr()), "adam")
        # the small tensors (a TGCN layer has 21): launches of up to 64 tensors each instead of a burst of tiny launches, during
        # which the GPU caught up with the host at the end of every step
        for t, group in small.items():
            k = len(group)
            if k < 3:                                     # a model with one or two tensors: the plain launch is cheaper to set up
                for pd, g_, m_, v_ in group:
                    _lib.check(lib.tagrec_adam_f32(_lib.ptr(pd), _lib.ptr(g_), _lib.ptr(m_), _lib.ptr(v_), pd.numel(), self.lr,
                                                   self.betas[0], self.betas[1], self.eps, t, _lib.stream_ptr()), "adam")
                continue
            arr = [(_lib.ctypes.c_void_p * k)(*[x[i].data_ptr() for x in group]) for i in range(4)]
            n = (_lib.ctypes.c_int64 * k)(*[x[0].numel() for x in group])
            _lib.check(lib.tagrec_adam_multi_f32(k, arr[0], arr[1], arr[2], arr[3], n, self.lr, self.betas[0], self.betas[1], self.eps,
                                                 t, _lib.stream_ptr()), "adam_multi")

    def prepare_state(self):
        """Allocate every parameter's state now (graph capture must not meet a first-use allocation whose zero fill
        would be replayed)."""
        for p in self.params:
            if id(p) not in self.state:
                st = self.state[id(p)] = {"m": torch.zeros_like(p.data), "v": torch.zeros_like(p.data), "t": 0}
                if self.capturable:
                    st["t_dev"] = torch.zeros(1, dtype=torch.int64, device=p.device)
                    st["coef"] = torch.zeros(2, dtype=torch.float32, device=p.device)


def _fused_tables(model):
    """The parameters a model can update inside its own backward pass: `model.fused_tables()` (TGCN: the three node
    tables) or the single `model.table`."""
    if hasattr(model, "fused_tables"):
        return list(model.fused_tables())
    table = getattr(model, "table", None)
    return [] if table is None else [table]


def fused_optimizer(model):
    """The optimizer whose update of `model.table` runs inside the model's backward pass (`Adam.fuse_into`), or None:
    the fusion is live only while that optimizer is still the newest one built over the table -- a second optimizer
    created for the same model (new run, new lr) revokes it instead of leaving the old one updating behind its back."""
    opt = getattr(model, "_fused_opt", None)
    if opt is None:
        return None
    tables = _fused_tables(model)
    if not tables or any(getattr(t, "_tagrec_owner", None) is not opt for t in tables):
        model._fused_opt = None
        return None
    return opt


def _step(loss_func, opt, data):
    lossx = loss_func(data)
    parts = torch.stack([x.detach() for x in lossx])
    loss = sum(lossx)
    if isinstance(opt, list):
        [op.zero_grad() for op in opt]
        loss.backward()
        [op.step() for op in opt]
    else:
        opt.zero_grad()
        loss.backward()
        opt.step()
    return parts


class GraphedStep:
    """One training step -- loss parts, backward, optimizer -- captured once as a HIP graph and replayed per batch.
    At the reference's dataset sizes (HetRec / ml-100k: ~1e5 edges) a step is a few dozen to a few hundred small
    launches and the GPU idles between them; replaying the captured graph removes the per-launch host cost.
    The optimizer must advance its step counter on the device (`Adam(capturable=True)`), the batch shape is fixed,
    and the step must not synchronise with the host (the fused model paths do not)."""

    def __init__(self, loss_func, opt, batch):
        opts = opt if isinstance(opt, list) else [opt]
        for op in opts:
            if not getattr(op, "capturable", False):
                raise _lib.TagrecError("GraphedStep: the optimizer must be Adam(capturable=True)")
            op.prepare_state()
        self.static_batch = batch.clone()
        self.graph = torch.cuda.CUDAGraph()
        self._opts = opts
        before = [op.step_count for op in opts]
        [op.zero_grad() for op in opts]
        with torch.cuda.graph(self.graph):
            self.static_parts = _step(loss_func, opt, self.static_batch)
        for op, b in zip(opts, before):          # capturing executed nothing: the host-side counters follow the replays
            op.step_count = b

    def __call__(self, batch):
        self.static_batch.copy_(batch)
        self.graph.replay()
        for op in self._opts:
            op.step_count += 1
        return self.static_parts.clone()


def _graph_key(loss_func, data):
    owner = getattr(loss_func, "__self__", None)
    return (id(owner), getattr(loss_func, "__name__", id(loss_func)), tuple(data.shape), data.dtype)


def epoch_training(training_data, loss_func, opt, verbose=True, graphs=None):
    """One pass over `training_data.mini_batch()` (basic_train.py:10-30): per batch the loss parts,
    their sum, zero_grad / backward / step.  Returns the per-batch totals as floats.
    The reference synchronises twice per step to read the losses (:16,27); here the parts stay on
    the device and are read back once at the end of the epoch -- same numbers, one sync.

    graphs: a dict kept by the caller across epochs.  When given, the first two batches of every (loss function,
    batch shape) run as usual -- they also grow every lazily sized buffer -- the third is captured as a HIP graph
    (`GraphedStep`) and from then on batches of that shape are graph replays; other shapes (the merged tail
    batch) and batches that are not plain tensors run as usual.  If a step cannot be captured it keeps running
    eagerly and the reason is recorded under graphs["errors"]."""
    parts_dev = []
    training_data.reset()
    for data in training_data.mini_batch():
        if graphs is not None and isinstance(data, torch.Tensor) and data.is_cuda:
            key = _graph_key(loss_func, data)
            entry = graphs.get(key, 0)
            if isinstance(entry, GraphedStep):
                parts_dev.append(entry(data))
                continue
            if entry == 2:
                try:
                    graphs[key] = GraphedStep(loss_func, opt, data)
                    parts_dev.append(graphs[key](data))
                    continue
                except Exception as exc:                 # not capturable: stay eager for this key
                    graphs[key] = -1
                    graphs.setdefault("errors", []).append(f"{key}: {type(exc).__name__}: {exc}")
                    torch.cuda.synchronize()
            elif entry >= 0:
                graphs[key] = entry + 1
        parts_dev.append(_step(loss_func, opt, data))
    if not parts_dev:
        return []
    parts = torch.stack(parts_dev).cpu().numpy()
    if verbose:
        print(f"[avg_loss of each part]:{list(parts.sum(0))}")
    # float32 sum of the parts, as `sum(lossx)` computes it on the device
    return [float(np.float32(sum(np.float32(v) for v in row))) for row in parts]


class Early_stop:
    """training/early_stop.py:8-41: track the best `key` metric (first entry of a list metric),
    save `state_dict` on improvement, stop after `patient_epoch` non-improving evaluations."""

    def __init__(self, args=None, config=None):
        cfg = config if config is not None else _GLOBAL_CFG
        self.best_value = None
        self.count_step = 0
        self.best_result = None
        self.best_epoch = 0
        self.patient_step = cfg["patient_epoch"]
        out_dir = getattr(args, "out_dir", None)
        self.save_path = f"{out_dir}/model.pth.tar" if out_dir else None
        self.key = cfg["early_stop_key"]
        self.higher = self.key in ("precision", "recall", "ndcg")

    def __call__(self, model, cur_results, epoch):
        cur = cur_results[self.key]
        cur = cur[0] if isinstance(cur, (list, tuple, np.ndarray)) else cur
        better = self.best_value is None or (cur > self.best_value if self.higher else cur < self.best_value)
        if better:
            self.best_value, self.count_step = cur, 0
            if self.save_path:
                os.makedirs(os.path.dirname(self.save_path), exist_ok=True)
                torch.save(model.state_dict(), self.save_path)
            self.best_result, self.best_epoch = cur_results, epoch
        else:
            self.count_step += 1
        return self.count_step > self.patient_step


class Basic_train:
    """basic_train.py:50-85: epochs x phases (phase i = train_data[i], loss_func[i], opt[i]),
    evaluation every `test_interval` epochs, early stop."""

    def __init__(self, train_data, loss_func, opt, test, args=None, config=None):
        self.cfg = config if config is not None else _GLOBAL_CFG
        self.train_sphase = len(train_data)
        self.train_data, self.loss_func, self.opt, self.test, self.args = train_data, loss_func, opt, test, args
        self.early_stop = Early_stop(args, self.cfg)
        # config["hip_graph"]: replay each phase's step as a captured HIP graph (needs Adam(capturable=True))
        self.graphs = {} if self.cfg.get("hip_graph", False) else None

    def run(self, model, verbose=True):
        history = []
        for ep in range(self.cfg["epochs"]):
            model.train()
            for i in range(self.train_sphase):
                start = time.time()
                loss_list = epoch_training(self.train_data[i], self.loss_func[i], self.opt[i], verbose=verbose, graphs=self.graphs)
                if verbose:
                    print(f"[Epoch:{ep}][Time:{(time.time() - start) / 60:.2}]:"
                          f"avg_loss_{i} :{sum(loss_list) / len(loss_list):.5}")
                history.append((ep, i, loss_list))
            if self.test is not None and ep % self.cfg["test_interval"] == 0:
                results = self.test.run(model)
                if verbose:
                    print(f"[Epoch {ep}] results: {results}")
                if self.early_stop(model, results, ep):
                    if verbose:
                        print(f"early stop trigger at epoch {ep}")
                    break
        return history
