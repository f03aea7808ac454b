"""Shared plumbing of the table-based models (LightGCN, NGCF): one contiguous [N, D] parameter whose
row slices are the reference's per-type tables, the `embed.k` state-dict layout, and the cached
`predict_rating` (/root/reference/model/lightgcn.py:37-47,84-89; model/ngcf.py:39-44,107-112)."""
import torch
import torch.nn as nn

from . import _lib


def xavier_tables(num_list, dim, device):
    """xavier_uniform_ per table, in order, drawn from torch's CPU generator so a seeded run
    reproduces the reference's initial values (lightgcn.py:37-47)."""
    parts = []
    for n in num_list:
        t = torch.empty(n, dim)
        nn.init.xavier_uniform_(t)
        parts.append(t)
    return torch.cat(parts, dim=0).to(device)


class StepWorkspace:
    """The [N, D]-sized buffers of a model's restricted training step, owned by the model and reused from step to step: no
    allocator traffic inside the step (at the C5 shape the caching allocator held 298 GB reserved against a 180 GB peak),
    and fixed addresses for a captured HIP graph.  One step at a time: `acquire(token)` hands the buffers to a forward pass;
    they are free again when its backward pass has run (`release`) or its autograd context has been dropped (the token
    died).  A forward pass that finds them taken -- two losses alive at once -- allocates its own buffers as before."""

    def __init__(self):
        self._buf = {}
        self._owner = None

    def acquire(self, token):
        if self._owner is not None and self._owner() is not None:
            return False
        import weakref
        self._owner = weakref.ref(token)
        return True

    def release(self, token):
        if self._owner is not None and self._owner() is token:
            self._owner = None

    def get(self, name, shape, dtype, device):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device:
            t = self._buf[name] = torch.empty(shape, dtype=dtype, device=device)
        return t

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self._buf.values())

    def clear(self):
        """Give the buffers back to the allocator (they are re-created by the next restricted step): before a phase that
        needs the memory for something else -- an all-rows pass at the C5 shape holds six more [N, D] tensors.  Refused
        while a forward pass still owns them."""
        if self._owner is not None and self._owner() is not None:
            return False
        self._buf = {}
        return True


class _Token:
    pass


def step_buffer(ws, name, shape, dtype, device):
    """A workspace buffer when a workspace is in use, otherwise a fresh torch.empty."""
    return torch.empty(shape, dtype=dtype, device=device) if ws is None else ws.get(name, shape, dtype, device)


class TableModel(nn.Module):
    def _init_table(self, data, use_tag, dim, device):
        if torch.device(device).type != "cuda":
            raise _lib.TagrecError(f"{type(self).__name__}: tagrec_amd needs a GPU device (no CPU path)")
        _lib.load()
        self.num_list = [data.num["user"], data.num["item"]] + ([data.num["tag"]] if use_tag else [])
        self.table = nn.Parameter(xavier_tables(self.num_list, dim, device))
        self._offsets = [0]
        for n in self.num_list:
            self._offsets.append(self._offsets[-1] + n)
        self._eval_cache = None
        self._register_state_dict_hook(_split_table_hook)
        self._register_load_state_dict_pre_hook(_merge_table_hook, with_module=True)

    @property
    def embed(self):
        return [self.table[a:b] for a, b in zip(self._offsets[:-1], self._offsets[1:])]

    def get_ego_embed(self):
        return self.embed

    get_ego_emb = get_ego_embed          # ngcf.py:92 spells it without the final 'ed'

    def _split(self, out):
        return tuple(out[a:b] for a, b in zip(self._offsets[:-1], self._offsets[1:]))

    def train(self, mode=True):
        self._eval_cache = None          # parameters may change once training resumes
        return super().train(mode)

    def predict_rating(self, users):
        """sigmoid(U_b I^T).  The reference re-runs forward() for every 512-user batch
        (lightgcn.py:85).  In eval mode the propagated tables are computed once and reused until
        `train()` is called again (same values, fewer propagations); in training mode every call
        propagates, as the reference does."""
        if self.training or self._eval_cache is None:
            with torch.no_grad():
                all_users, all_items = self.forward()[:2]
            if not self.training:
                self._eval_cache = (all_users, all_items)
        else:
            all_users, all_items = self._eval_cache
        users = users.to(self.table.device)
        return torch.sigmoid(torch.matmul(all_users[users], all_items.t()))


def _split_table_hook(module, state_dict, prefix, local_metadata):
    table = state_dict.pop(prefix + "table")
    for k, (a, b) in enumerate(zip(module._offsets[:-1], module._offsets[1:])):
        state_dict[f"{prefix}embed.{k}"] = table[a:b]
    # keep the reference's key order: embed.* first
    for key in [k for k in state_dict if k.startswith(prefix) and not k.startswith(prefix + "embed.")]:
        state_dict.move_to_end(key)
    return state_dict


def _merge_table_hook(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
    keys = [f"{prefix}embed.{k}" for k in range(len(module.num_list))]
    module._eval_cache = None
    if all(k in state_dict for k in keys):
        state_dict[prefix + "table"] = torch.cat([state_dict.pop(k) for k in keys], dim=0)
