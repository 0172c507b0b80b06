"""``Adam`` with torch.optim.Adam's constructor, state layout and update (the optimiser of the reference's train.py:99-100), whose step
is ONE launch of the multi-tensor HIP kernel ``mi_adam_step`` (csrc/optim.hip) for all parameters on the GPU.

State per parameter: ``step`` (a host float tensor like torch's default), ``exp_avg``, ``exp_avg_sq`` -- a ``state_dict()`` of either
optimiser loads into the other.  Parameters that are not fp32 / not on the GPU / not contiguous take torch's own update."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L

CHUNK = 4096           # elements per launched workgroup


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0., amsgrad: bool = False,
                 maximize: bool = False, capturable: bool = False):
        if amsgrad or maximize or capturable:
            raise NotImplementedError("minimagen_amd.optim.Adam implements torch.optim.Adam's default update only (no amsgrad / maximize / capturable)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}
        self._count = {}          # parameter -> step count as a Python int (mirrored into the state's host ``step`` tensor at every step)

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            self._count[p] = 0
        elif p not in self._count:                      # state that came in through load_state_dict
            self._count[p] = int(float(st["step"]))
        return st

    def state_dict(self):
        for p, n in self._count.items():
            if p in self.state:
                self.state[p]["step"].fill_(float(n))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        for g in state_dict.get("param_groups", ()):
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable"):
                raise NotImplementedError("minimagen_amd.optim.Adam: a state dict with amsgrad / maximize / capturable set would be silently ignored")
        super().load_state_dict(state_dict)
        self._count = {}
        self._tables = {}

    def _table(self, gi, ps):
        """device-resident tensor / chunk tables of one parameter group, rebuilt only when a pointer changed (zero_grad(set_to_none=True)
        re-allocates the gradients: the caching allocator usually hands the same blocks back)"""
        rows = [(p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel()) for p in ps]
        key = tuple(rows)
        tb = self._tables.get(gi)
        if tb is not None and tb[0] == key:
            return tb
        dev = ps[0].device
        tens = np.zeros((len(rows), 5), dtype=np.int64)
        tens[:] = rows
        ct, co = [], []
        for k, r in enumerate(rows):
            n = -(-r[4] // CHUNK)
            ct += [k] * n
            co += list(range(n))
        # through pinned memory, asynchronously: zero_grad(set_to_none=True) re-allocates the gradients, so this table is rebuilt on most steps, and a
        # pageable host -> device copy waits for everything queued before it -- one full host / GPU synchronisation per training step
        up = (lambda a: torch.from_numpy(a).pin_memory().to(dev, non_blocking=True)) if dev.type == "cuda" else (lambda a: torch.from_numpy(a).to(dev))
        tb = (key, up(tens.view(np.uint8).reshape(-1)), up(np.asarray(ct, dtype=np.int32)), up(np.asarray(co, dtype=np.int32)), len(ct))
        self._tables[gi] = tb
        return tb

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            fast, slow = [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                ok = p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous() \
                    and (p.is_cuda or L.backend() == "hipemu") and p.device == p.grad.device
                (fast if ok else slow).append(p)
            for p in fast + slow:
                st = self._state_of(p)
                self._count[p] += 1
                st["step"].fill_(float(self._count[p]))          # a host scalar: readers of optimizer.state[p]['step'] see the live count
            # parameters of one group share the step count in every ordinary use; groups whose counts differ are split by count
            by_step = {}
            for p in fast:
                by_step.setdefault(float(self._count[p]), []).append(p)
            # (table slots are keyed by group and position among the distinct counts, not by the count itself: nothing accumulates per step)
            for slot, (t, ps) in enumerate(by_step.items()):
                _, tens, ct, co, nchunks = self._table((gi, slot), ps)
                a = L.MiAdamParams()
                a.tensors, a.chunk_tensor, a.chunk_off, a.nchunks, a.chunk = tens.data_ptr(), ct.data_ptr(), co.data_ptr(), nchunks, CHUNK
                a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = group["lr"], b1, b2, group["eps"], group["weight_decay"]
                a.bias_correction1, a.bias_correction2 = 1.0 - b1 ** t, 1.0 - b2 ** t
                a.one_minus_beta1, a.one_minus_beta2 = 1.0 - b1, 1.0 - b2
                L.check(lib.mi_adam_step(C.byref(a), L.current_stream()), "mi_adam_step")
                for p in ps:                                # the kernel wrote through raw pointers: tell autograd / every version-keyed cache
                    torch.autograd.graph.increment_version(p)
            for p in slow:                                  # torch's single-tensor update, same formulas
                st = self.state[p]
                g = p.grad if group["weight_decay"] == 0 else p.grad.add(p, alpha=group["weight_decay"])
                t = float(self._count[p])
                st["exp_avg"].lerp_(g, 1 - b1)
                st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (st["exp_avg_sq"].sqrt() / (1 - b2 ** t) ** 0.5).add_(group["eps"])
                p.addcdiv_(st["exp_avg"], denom, value=-group["lr"] / (1 - b1 ** t))
        return loss
