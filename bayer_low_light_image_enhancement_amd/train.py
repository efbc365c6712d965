"""Training step on the HIP path (SURVEY.md section 8 f3 / BASELINE configs[4]).

Mirrors the reference's loop body (train.py:127-147; RawFomer_WFB_FFAB/train.py:124 for the L1 loss):

    pred = model(inp); loss = criterion(pred, gt); optimizer.zero_grad(); loss.backward(); optimizer.step()

with torch.autograd replaced by the library's explicit adjoint schedule (``rf_train_step``) and ``nn.DataParallel``
(train.py:108-111) by one process per GPU: every rank runs its own images, the flat gradient buffer is all-reduced in
fixed-size buckets (RCCL with the ``nccl`` backend, ``gloo`` in the CPU tests), then ``rf_adam_step`` updates the flat
parameter buffer.  PyTorch provides device memory, streams and the collective; no torch op computes anything.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .model import RawFormer

LOSS_L1, LOSS_CHARBONNIER = 0, 1


def bucket_bounds(n: int, bucket_floats: int):
    """Contiguous [lo, hi) slices of a flat buffer of ``n`` floats, ``bucket_floats`` each (last one shorter)."""
    return [(lo, min(n, lo + bucket_floats)) for lo in range(0, n, bucket_floats)]


def allreduce_flat(flat: torch.Tensor, group=None, bucket_floats: int = 1 << 20):
    """Sum ``flat`` over the ranks in buckets (async, waited at the end).  9.9 MB of RawFormer-S gradients = 3 buckets of
    4 MiB: large enough for xGMI links, small enough that the first bucket is on the wire while the others queue."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    works = [dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True) for lo, hi in bucket_bounds(flat.numel(), bucket_floats)]
    for w in works:
        w.wait()


class OverlappedReducer:
    """All-reduce of the flat gradient buffer OVERLAPPED with the backward pass (the reference reduces inside backward through
    ``nn.DataParallel``, train.py:108-111).  ``rf_train_step`` announces ranges of the flat buffer whose gradients are final, from
    the end of the buffer towards its start (``rf_set_grad_ready``); ``ready`` coalesces them into buckets of at least
    ``bucket_floats`` and starts each bucket's asynchronous all-reduce at once -- with the ``nccl`` backend (RCCL) the collective
    is ordered behind the kernels already enqueued on the current stream and runs beside the rest of the backward pass; ``finish``
    waits for all of them.  Works on any tensor / backend (the CPU tests drive it with ``gloo``)."""

    def __init__(self, flat: torch.Tensor, group=None, bucket_floats: int = 1 << 20):
        self.flat, self.group, self.bucket_floats = flat, group, int(bucket_floats)
        self.works, self.buckets = [], []
        self.hi = self.lo = flat.numel()

    def begin(self) -> None:
        self.works, self.buckets = [], []
        self.hi = self.lo = self.flat.numel()

    def ready(self, offset: int, count: int) -> None:
        import torch.distributed as dist
        if offset + count != self.lo:
            raise RuntimeError(f"gradient ranges must arrive contiguously from the end: got [{offset}, {offset + count}), expected to end at {self.lo}")
        self.lo = offset
        if self.hi - self.lo >= self.bucket_floats or self.lo == 0:
            if self.hi > self.lo:
                self.buckets.append((self.lo, self.hi))
                self.works.append(dist.all_reduce(self.flat[self.lo:self.hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.hi = self.lo

    def finish(self) -> None:
        if self.lo != 0:
            raise RuntimeError(f"gradient ranges stopped at float {self.lo}: the step did not announce the whole buffer")
        for w in self.works:
            w.wait()
        self.works = []


class Trainer:
    """Owns the flat parameter / gradient / moment buffers of a ``RawFormer`` and runs training steps on them."""

    def __init__(self, model: RawFormer, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 decoupled: bool = False, loss: str = "l1", charbonnier_eps: float = 1e-3, group=None, overlap_allreduce: bool = True,
                 bucket_floats: int = 1 << 20):
        if model.variant not in ("plain", "flca"):
            raise RuntimeError("Trainer: the adjoint schedule exists for variants 'plain' and 'flca'")
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Trainer needs the model on a ROCm device: there is no CPU path in this package")
        self.model, self.group = model, group
        self.lr, self.betas, self.eps, self.wd, self.decoupled = float(lr), betas, float(eps), float(weight_decay), bool(decoupled)
        self.loss_mode = {"l1": LOSS_L1, "charbonnier": LOSS_CHARBONNIER}[loss]
        self.loss_eps = float(charbonnier_eps)
        lib = _lib.load()
        self.state = model._state_for(dev)
        n = C.c_size_t()
        _lib.check(lib.rf_flat_param_floats(self.state.handle, C.byref(n)), "rf_flat_param_floats")
        self.n = n.value
        self.flat = torch.zeros(self.n, dtype=torch.float32, device=dev)
        params = dict(model.named_parameters())
        off = C.c_size_t()
        self.slices = {}
        with torch.no_grad():
            for i, k in enumerate(model._param_names):
                _lib.check(lib.rf_flat_offset(self.state.handle, i, C.byref(off)), "rf_flat_offset")
                p = params[k]
                view = self.flat[off.value: off.value + p.numel()].view(p.shape)
                view.copy_(p)
                p.data = view                       # the module's parameters ARE the flat buffer from now on
                self.slices[k] = (off.value, p.numel())
        self.grads = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.step_no = 0
        self.workspace: Optional[torch.Tensor] = None
        self.loss_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        # gradient all-reduce started range by range from inside rf_train_step (several ranks only)
        self.overlap = bool(overlap_allreduce)
        self.overlap_always = overlap_allreduce == "always"      # also with a world of one (tests of the RCCL stream ordering)
        self.reducer = OverlappedReducer(self.grads, group, bucket_floats)
        self._reduced = False
        self._ready_cb = _lib.GRAD_READY_FN(lambda user, off, cnt, stream: self.reducer.ready(int(off), int(cnt)))   # kept alive with the Trainer
        model.invalidate_packed()

    def grad_of(self, key: str) -> torch.Tensor:
        off, n = self.slices[key]
        return self.grads[off: off + n].view(dict(self.model.named_parameters())[key].shape)

    def forward_backward(self, x: torch.Tensor, gt: torch.Tensor, want_pred: bool = False):
        """Loss (device scalar) and, in ``self.grads``, this rank's gradient of the mean loss over ITS images."""
        lib = _lib.load()
        x, gt = x.detach().float().contiguous(), gt.detach().float().contiguous()
        b, c, h, w = x.shape
        if c != 1 or h % 16 or w % 64:
            raise RuntimeError(f"training step: mosaic [B,1,H,W] with H % 16 == 0 and W % 64 == 0, got {tuple(x.shape)}")
        if tuple(gt.shape) != (b, self.model.out_channels, h, w):
            raise RuntimeError(f"ground truth must be {(b, self.model.out_channels, h, w)}, got {tuple(gt.shape)}")
        H, W = h // 2, w // 2
        with torch.cuda.device(x.device):
            self.model._sync_params(self.state, x.device, pack=False)   # the (flat-buffer) pointers, once: rf_train_step reads raw weights
            sz = C.c_size_t()
            _lib.check(lib.rf_train_workspace_bytes(self.state.handle, b, H, W, C.byref(sz)), "rf_train_workspace_bytes")
            if self.workspace is None or self.workspace.numel() < sz.value:
                self.workspace = None
                self.workspace = torch.empty(sz.value, dtype=torch.uint8, device=x.device)
            pred = torch.empty_like(gt) if want_pred else None
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            import torch.distributed as dist
            overlapped = self.overlap and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.overlap_always)
            _lib.check(lib.rf_set_grad_ready(self.state.handle, self._ready_cb if overlapped else _lib.GRAD_READY_FN(), None), "rf_set_grad_ready")
            if overlapped:
                self.reducer.begin()
            _lib.check(lib.rf_train_step(self.state.handle, C.c_void_p(x.data_ptr()), C.c_void_p(gt.data_ptr()), C.c_void_p(self.grads.data_ptr()),
                                         C.c_void_p(self.loss_dev.data_ptr()), C.c_void_p(pred.data_ptr() if want_pred else None),
                                         C.c_void_p(self.workspace.data_ptr()), self.workspace.numel(), b, H, W, self.loss_mode, self.loss_eps,
                                         stream), "rf_train_step")
            self._reduced = overlapped          # the buckets are on the wire (or done); optimizer_step waits for them
        return (self.loss_dev, pred) if want_pred else self.loss_dev

    def optimizer_step(self):
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if self._reduced:
            self.reducer.finish()
            self._reduced = False
        else:
            allreduce_flat(self.grads, self.group)
        self.step_no += 1
        lib = _lib.load()
        dev = self.flat.device
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.rf_adam_step(C.c_void_p(self.flat.data_ptr()), C.c_void_p(self.grads.data_ptr()), C.c_void_p(self.m.data_ptr()),
                                        C.c_void_p(self.v.data_ptr()), self.n, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                        int(self.decoupled), self.step_no, 1.0 / world, stream), "rf_adam_step")
        self.model.invalidate_packed()          # the weights moved: packed copies are stale

    def step(self, x: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        loss = self.forward_backward(x, gt)
        self.optimizer_step()
        return loss
