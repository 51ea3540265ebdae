"""``torch.optim.Adam`` / ``AdamW`` behind a resident training step, as ONE launch.

The reference builds its optimizer from ``OPTIM_DICT`` (config/config.py:24-28; train/train.py:82,
train/train_clustering.py:30-33) and calls ``optimizer.step()`` after every backward.  The resident steps
(``step.ResidentTrainStep`` / ``step.ScnTrainStep``) leave all parameter gradients in ONE flat buffer; ``FlatAdam``
applies torch's single-tensor Adam / AdamW update to the module's own parameter tensors from that buffer with one
launch (``hscn_adam_step``, csrc/optim.hip) where torch's capturable fused optimizer takes two (8.6 us of kernel
time behind a 21 us stage-A step).  Same formulas, operation for operation; the state lives in flat buffers
(``exp_avg``, ``exp_avg_sq``, a float ``step`` counter and the learning rate, all on the device: the launch is
capturable and a scheduler may rewrite ``lr`` between launches through ``set_lr``).
"""
from __future__ import annotations

import ctypes
from typing import Sequence, Tuple

import torch
from torch import Tensor

from . import _hip


class _AdamC(ctypes.Structure):        # include/hscn.h: hscn_adam
    _fields_ = [("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p), ("step", ctypes.c_void_p),
                ("beta_pows", ctypes.c_void_p), ("lr", ctypes.c_void_p), ("beta1", ctypes.c_double),
                ("beta2", ctypes.c_double), ("eps", ctypes.c_double), ("weight_decay", ctypes.c_double),
                ("decoupled", ctypes.c_int)]


class FlatAdam:
    """``param_grads``: ``[(parameter, view of its gradient inside flat_grads)]`` in flat order -- what
    ``ResidentTrainStep.param_grads`` / ``ScnTrainStep.param_grads`` hold.  ``decoupled=True`` is ``AdamW``.
    ``amsgrad`` / ``maximize`` are not offered (the reference never sets them)."""

    MAX_PARAMS = 32

    def __init__(self, param_grads: Sequence[Tuple[Tensor, Tensor]], flat_grads: Tensor, lr: float = 1e-3,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 decoupled: bool = False):
        if not param_grads or len(param_grads) > self.MAX_PARAMS:
            raise ValueError(f"FlatAdam takes 1..{self.MAX_PARAMS} parameter tensors")
        if flat_grads.dtype != torch.float32 or not flat_grads.is_contiguous():
            raise ValueError("the flat gradient buffer must be contiguous float32")
        dev = flat_grads.device
        base, off, offs = flat_grads.data_ptr(), 0, [0]
        for p, g in param_grads:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise ValueError("parameters must be contiguous float32 tensors on the gradient buffer's device")
            if g.data_ptr() != base + 4 * off or g.numel() != p.numel():
                raise ValueError("param_grads must tile the front of the flat gradient buffer in order")
            off += p.numel()
            offs.append(off)
        self.params = [p for p, _ in param_grads]
        self.P = off
        self.grads = flat_grads
        self.betas, self.eps, self.weight_decay, self.decoupled = (float(betas[0]), float(betas[1])), float(eps), \
            float(weight_decay), bool(decoupled)
        self._ptr_list = [p.data_ptr() for p in self.params]
        self._ptrs = (ctypes.c_void_p * len(self.params))(*self._ptr_list)      # host tables: kernel arguments
        self._off = (ctypes.c_int32 * len(offs))(*offs)
        self.exp_avg = torch.zeros(self.P, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.P, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self._beta_pows = torch.ones(2, dtype=torch.float64, device=dev)      # beta1^t, beta2^t (running products)
        self._lr = torch.tensor([float(lr)], dtype=torch.float64, device=dev)
        self.lr = float(lr)

    @property
    def c(self) -> _AdamC:
        """The state as ``hscn_adam`` (for a step that applies the update in its own launch:
        ``ScnTrainStep.run(opt=...)``)."""
        if not hasattr(self, "_c"):
            self._c = _AdamC(_hip.ptr(self.exp_avg), _hip.ptr(self.exp_avg_sq), _hip.ptr(self.step_count),
                             _hip.ptr(self._beta_pows), _hip.ptr(self._lr), self.betas[0], self.betas[1], self.eps,
                             self.weight_decay, int(self.decoupled))
        return self._c

    def set_lr(self, lr: float) -> None:
        """A scheduler's new learning rate (one tiny copy; the captured launch reads the device value)."""
        self.lr = float(lr)
        self._lr.fill_(self.lr)

    def step(self) -> None:
        """One optimizer step on the gradients the flat buffer holds NOW.  Asynchronous, capturable."""
        _hip.call("hscn_adam_step", self._ptrs, self._off, len(self.params), _hip.ptr(self.grads),
                  _hip.ptr(self.exp_avg), _hip.ptr(self.exp_avg_sq), self.P, _hip.ptr(self.step_count),
                  _hip.ptr(self._beta_pows), _hip.ptr(self._lr), self.betas[0], self.betas[1], self.eps, self.weight_decay, int(self.decoupled),
                  _hip.stream())

    def reset_state(self) -> None:
        """Back to the state of a freshly built optimizer (moments and step counter zero), in place."""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_count.zero_()
        self._beta_pows.fill_(1.0)

    def step_from_autograd(self) -> None:
        """A step on the gradients an EAGER backward left in ``p.grad`` (an epoch's ragged last batch runs through
        autograd): copied into the flat buffer first (parameters without a gradient contribute zeros, as torch's
        optimizers skip them only when ALL their history is empty -- here they have none either: the resident
        steps never produce a gradient for them and they are not in ``params``)."""
        off = 0
        with torch.no_grad():
            for p in self.params:
                dst = self.grads[off: off + p.numel()].view_as(p)
                if p.grad is None:
                    dst.zero_()
                elif p.grad.data_ptr() != dst.data_ptr():
                    dst.copy_(p.grad)
                off += p.numel()
        self.step()

    def check(self) -> None:
        """The parameter tensors are still the ones the pointer table was built from (``module.to()`` / a loaded
        checkpoint that re-allocates them would leave the launch updating dead memory)."""
        if self._ptr_list != [p.data_ptr() for p in self.params]:
            raise RuntimeError("a parameter tensor was re-allocated after FlatAdam was built")

    def zero_grad(self, set_to_none: bool = True) -> None:
        """For an EAGER backward (the resident steps overwrite the flat buffer and need none of this): drop the
        parameters' ``.grad`` (autograd then allocates fresh ones; ``step_from_autograd`` collects them) or zero them."""
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @classmethod
    def from_config(cls, optim_type: str, param_grads, flat_grads, lr: float, weight_decay: float):
        """The reference's ``OPTIM_DICT[optim_type](params, lr=..., weight_decay=...)`` for the two members this
        class covers ("adam", "adamW"); None for the others (the caller keeps the torch optimizer)."""
        if optim_type == "adam":
            return cls(param_grads, flat_grads, lr=lr, weight_decay=weight_decay, decoupled=False)
        if optim_type == "adamW":
            return cls(param_grads, flat_grads, lr=lr, weight_decay=weight_decay, decoupled=True)
        return None
