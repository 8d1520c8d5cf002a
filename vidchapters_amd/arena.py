"""Flat parameter arena: one fp32 master buffer, one fp32 gradient buffer and one bf16 shadow buffer
for ALL parameters of the model, laid out in backward-completion order.

Why: (1) the fused clip+Adam kernel, the grad-norm reduction and the DP all-reduce buckets work on
contiguous ranges instead of ~260 separate tensors; (2) q|k|v (and cross k|v) weights sit next to each other
so one GEMM computes the fused projection; (3) wgrad kernels accumulate straight into the gradient arena,
``param.grad`` being a view of it (the reference's ``optimizer.step()`` / ``clip_grad_norm_`` still work).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
from torch import nn

ALIGN = 64  # elements; keeps every view 256-byte aligned in fp32 and 128-byte aligned in bf16


class ParamArena:
    def __init__(self, named: List[Tuple[str, nn.Parameter]], device: torch.device, tail_pad: int = 0):
        """``tail_pad``: zero elements appended after the LAST parameter in all three buffers (the tied embedding is last: the LM-head
        GEMMs then read it as a [vocab rounded up to 64][d_model] matrix whose extra rows are zero and stay zero -- their gradient
        is never written and Adam maps (0 weight, 0 gradient) to 0)."""
        self.device = device
        self.names: List[str] = []
        self.offsets: Dict[str, int] = {}
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        self.params: Dict[str, nn.Parameter] = {}
        off = 0
        seen = {}
        for name, p in named:
            if id(p) in seen:          # tied parameter registered twice
                continue
            seen[id(p)] = name
            self.names.append(name)
            self.offsets[name] = off
            self.shapes[name] = tuple(p.shape)
            self.params[name] = p
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off + (tail_pad + ALIGN - 1) // ALIGN * ALIGN
        self.master = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(self.numel, dtype=torch.bfloat16, device=device)
        with torch.no_grad():
            for name in self.names:
                p = self.params[name]
                view = self.view(self.master, name)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = self.view(self.grad, name)
        self._ptrs = {n: self.params[n].data_ptr() for n in self.names}
        self._seen_version = -1

    # ------------------------------------------------------------------ views
    def view(self, buf: torch.Tensor, name: str, shape=None) -> torch.Tensor:
        o = self.offsets[name]
        shp = self.shapes[name] if shape is None else shape
        n = 1
        for s in shp:
            n *= s
        return buf[o:o + n].view(*shp)

    def w(self, name: str, shape=None) -> torch.Tensor:
        """bf16 shadow view (GEMM operand)."""
        return self.view(self.shadow, name, shape)

    def f(self, name: str, shape=None) -> torch.Tensor:
        """fp32 master view (norm weights, biases, bias tables)."""
        return self.view(self.master, name, shape)

    def g(self, name: str, shape=None) -> torch.Tensor:
        """fp32 gradient view."""
        return self.view(self.grad, name, shape)

    def adjacent(self, *names: str) -> bool:
        for a, b in zip(names, names[1:]):
            n = 1
            for s in self.shapes[a]:
                n *= s
            if self.offsets[a] + n != self.offsets[b]:
                return False
        return True

    # ------------------------------------------------------------------ consistency with the nn.Module
    def intact(self) -> bool:
        """False if someone rebound parameter storage (``model.to()``, ``.half()``...)."""
        for n in self.names:
            if self.params[n].data_ptr() != self._ptrs[n]:
                return False
        return True

    def attach_grads(self) -> bool:
        """Re-attach ``param.grad`` views dropped by ``optimizer.zero_grad(set_to_none=True)``.
        Returns True if any was missing (the caller then zeroes the gradient arena)."""
        missing = False
        for n in self.names:
            p = self.params[n]
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * self.offsets[n]:
                p.grad = self.view(self.grad, n)
                missing = True
        return missing

    def refresh_shadow(self, force: bool = False) -> None:
        """bf16 shadow <- fp32 master when any parameter was modified in place by torch code
        (external optimizer, load_state_dict, dvc.py's div_)."""
        from . import lib as L
        # ``p.data = view`` keeps each Parameter's own version counter, so in-place updates made through the
        # Parameters (optimizer.step, div_, load_state_dict) show up there, not on the arena tensor
        v = self.master._version + sum(self.params[n]._version for n in self.names)
        if force or v != self._seen_version:
            guard = getattr(self, "stale_guard", None)
            if guard is not None and not force:      # sharded optimizer: never re-cast stale foreign stripes over the gathered shadow
                guard("refresh of the bf16 shadow weights from the fp32 masters (a Parameter was modified in place, or mark_dirty())")
            L.cast_bf16(self.master, self.shadow, self.numel)
            self._seen_version = v
