"""Drop-in ``RawFormer(nn.Module)`` whose ``forward`` runs on hand-written gfx950 kernels.

Interface kept from the reference:

* constructor ``RawFormer(inp_channels=1, out_channels=3, dim=48, num_heads=[8,8,8,8],
  ffn_expansion_factor=2)`` -- RawFomer_WFB_FFAB/model.py:448,
  FrequencyawareLumaChromaAttentionRAWFormer.py:296 (root ``model.py:111`` spells the same
  arguments ``in_ch, out_ch, dim, heads, ffn_exp``; those keywords are accepted too);
* ``state_dict()`` / ``load_state_dict(strict=True)`` with the reference's key names (the
  ``module.`` prefix left by ``nn.DataParallel`` is stripped, test.py:90);
* ``forward(x[B,1,2H,2W]) -> [B,3,2H,2W]`` float32, input not modified (test.py:116).

``variant='flca'`` is the wiring of FrequencyawareLumaChromaAttentionRAWFormer.py:284-370;
``variant='truecolor'`` is ``TrueColorRawFormer`` (BayerTORGBColorMultiLvl.py:387-462: learned Bayer
front end, colour-aware pyramid FLCA, ``exp(log_temperature)`` attention, colour-correction head;
``flca_levels`` as in its constructor); ``variant='plain'`` replaces the FLCA
branch by the 3x3 conv branch of RawFomer_WFB_FFAB/model.py:393-412 (``branch_lrelu=True``)
or model.py:94-108 (``False``) and, with ``clamp_io=True``, adds the I/O clamps of
RawFomer_WFB_FFAB/model.py:475,508.

The module owns ordinary ``nn.Parameter`` objects; the HIP library borrows their device
pointers.  There is no PyTorch fallback: on a CPU tensor ``forward`` raises.
"""
from __future__ import annotations

import ctypes as C
import math
import threading
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib

_ROOT_STAGE_ALIASES = (
    ("transformer.norm1.norm.", "Transformer.norm1.body."),
    ("transformer.norm2.norm.", "Transformer.norm2.body."),
    ("transformer.attn.scale", "Transformer.attn.temperature"),
    ("transformer.attn.qkv.0.", "Transformer.attn.qkv."),
    ("transformer.attn.qkv.1.", "Transformer.attn.qkv_dwconv."),
    ("transformer.attn.proj.", "Transformer.attn.project_out."),
    ("transformer.ffn.net.0.", "Transformer.ffn.pointwise1."),
    ("transformer.ffn.net.1.", "Transformer.ffn.depthwise."),
    ("transformer.ffn.net.3.", "Transformer.ffn.pointwise2."),
    ("reduce.", "channel_reduce."),
    ("out.0.", "Conv_out."),
    ("conv.", "conv."),
)


def canonical_key(key: str) -> str:
    """Map a checkpoint key to this module's naming: strips ``module.`` (test.py:90) and
    translates the root ``model.py`` layout to the
    RawFomer_WFB_FFAB / FrequencyawareLumaChroma layout."""
    if key.startswith("module."):
        key = key[len("module."):]
    head, _, rest = key.partition(".")
    stage = None
    if head == "embed":
        return "embedding." + rest
    if head == "output" and rest.startswith("0."):
        return "conv_out." + rest[2:]
    if head == "encoder":
        i, _, rest = rest.partition(".")
        stage = f"conv_tran{int(i) + 1}."
    elif head == "bottleneck":
        stage = "conv_tran4."
    elif head == "decoder":
        i, _, rest = rest.partition(".")
        stage = f"conv_tran{int(i) + 5}."
    elif head == "downsamples":
        i, _, rest = rest.partition(".")
        return f"down{int(i) + 1}.body.0." + rest[len("net.0."):] if rest.startswith("net.0.") else key
    elif head == "upsamples":
        i, _, rest = rest.partition(".")
        return f"up{int(i) + 1}." + rest
    if stage is None:
        return key
    for a, b in _ROOT_STAGE_ALIASES:
        if rest.startswith(a):
            return stage + b + rest[len(a):]
    return stage + rest


class _Node(nn.Module):
    """Name-only container so parameters get the reference's dotted state_dict keys."""

    def forward(self, *a, **k):  # pragma: no cover - never called
        raise RuntimeError("container module; call RawFormer.forward")


class _DeviceState:
    def __init__(self):
        self.handle = C.c_void_p()
        self.signature = None
        self.ptrs = None          # data_ptr of every parameter as registered with the handle
        self.packed: Optional[torch.Tensor] = None
        self.workspace: Optional[torch.Tensor] = None


class RawFormer(nn.Module):
    def __init__(self, inp_channels: int = 1, out_channels: int = 3, dim: int = 48,
                 num_heads: Sequence[int] = (8, 8, 8, 8), ffn_expansion_factor: int = 2, *,
                 variant: str = "flca", branch_lrelu: bool = True, clamp_io: bool = False, flca_levels: int = 2,
                 in_ch: Optional[int] = None, out_ch: Optional[int] = None,
                 heads: Optional[Sequence[int]] = None, ffn_exp: Optional[int] = None):
        super().__init__()
        inp_channels = in_ch if in_ch is not None else inp_channels
        out_channels = out_ch if out_ch is not None else out_channels
        num_heads = list(heads if heads is not None else num_heads)
        ffn_expansion_factor = ffn_exp if ffn_exp is not None else ffn_expansion_factor
        if variant not in ("flca", "plain", "truecolor"):
            raise ValueError(f"variant must be 'flca', 'plain' or 'truecolor', got {variant!r}")
        if len(num_heads) != 4:
            raise ValueError("num_heads must have 4 entries")
        self.dim, self.inp_channels, self.out_channels = int(dim), int(inp_channels), int(out_channels)
        self.num_heads, self.ffn_expansion_factor = [int(h) for h in num_heads], int(ffn_expansion_factor)
        self.variant, self.branch_lrelu, self.clamp_io = variant, bool(branch_lrelu), bool(clamp_io)
        self.flca_levels = int(flca_levels)

        cfg = self._config()
        lib = _lib.load()
        probe = C.c_void_p()
        _lib.check(lib.rf_create(C.byref(cfg), C.byref(probe)), "rf_create")
        try:
            self._param_names: List[str] = []
            name, shape, ndim = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
            for i in range(lib.rf_param_count(probe)):
                _lib.check(lib.rf_param_info(probe, i, C.byref(name), C.byref(shape), C.byref(ndim)), "rf_param_info")
                key = name.value.decode()
                self._param_names.append(key)
                self._register(key, nn.Parameter(torch.empty(tuple(shape[: ndim.value]), dtype=torch.float32)))
        finally:
            lib.rf_destroy(probe)
        if variant in ("flca", "truecolor"):
            # fixed buffers the reference keeps in its state_dict (values are constants in the kernels)
            if variant == "flca":
                for k, v in (("r_w", 0.299), ("g_w", 0.587), ("b_w", 0.114)):
                    self._register_buffer("luma_chroma." + k, torch.tensor(v, dtype=torch.float32))
            else:
                self._register_buffer("bayer_processor.y_weights", torch.tensor([0.2126, 0.7152, 0.0722], dtype=torch.float32))
            hv = torch.tensor([1.0, 1.0]) / math.sqrt(2.0)
            gv = torch.tensor([1.0, -1.0]) / math.sqrt(2.0)
            filt = torch.stack([torch.outer(hv, hv), torch.outer(hv, gv), torch.outer(gv, hv), torch.outer(gv, gv)]).unsqueeze(1)
            for i in range(1, 8):
                self._register_buffer(f"conv_tran{i}.FLCA.dwt.filt", filt.clone())
        self.reset_parameters()
        self._rt: Dict[int, _DeviceState] = {}
        self._rt_lock = threading.Lock()
        self._rt_owner = id(self)   # shallow copies (module.__dict__ copies) share _rt; only the owner frees the handles

    # ------------------------------------------------------------------ construction helpers
    def _config(self) -> _lib.RfConfig:
        return _lib.RfConfig(self.dim, (C.c_int32 * 4)(*self.num_heads), self.inp_channels, self.out_channels,
                             self.ffn_expansion_factor,
                             {"flca": _lib.RF_VARIANT_FLCA, "plain": _lib.RF_VARIANT_PLAIN, "truecolor": _lib.RF_VARIANT_TRUECOLOR}[self.variant],
                             int(self.branch_lrelu), int(self.clamp_io), self.flca_levels)

    def _node(self, path: List[str]) -> nn.Module:
        mod: nn.Module = self
        for part in path:
            if part not in mod._modules:
                mod.add_module(part, _Node())
            mod = mod._modules[part]
        return mod

    def _register(self, key: str, p: nn.Parameter) -> None:
        *path, leaf = key.split(".")
        self._node(path).register_parameter(leaf, p)

    def _register_buffer(self, key: str, t: torch.Tensor) -> None:
        *path, leaf = key.split(".")
        self._node(path).register_buffer(leaf, t)

    @torch.no_grad()
    def reset_parameters(self) -> None:
        """PyTorch's default initialisers (what the reference's ``__init__`` leaves behind)."""
        params = dict(self.named_parameters())
        for key, p in params.items():
            leaf = key.rsplit(".", 1)[-1]
            if leaf in ("alpha", "beta", "gamma", "temperature"):
                p.fill_(1.0)
            elif leaf == "log_temperature":                    # TrueColorRawFormer's own initial values
                p.zero_()                                      # (BayerTORGBColorMultiLvl.py:78-83, 144, 331)
            elif leaf == "wb_gains":
                p.copy_(torch.tensor([1.8, 1.0, 1.0, 1.6]))
            elif leaf == "color_matrix":
                p.copy_(torch.eye(3, 4))
            elif leaf == "gamma_param":
                p.fill_(2.2)
            elif p.dim() == 1:
                if leaf == "weight":
                    p.fill_(1.0)                      # LayerNorm scale
                elif "norm" in key:
                    p.zero_()                          # LayerNorm shift
                else:                                  # conv bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                    w = params[key[: -len("bias")] + "weight"]
                    fan_in = w[0].numel()              # also what torch uses for ConvTranspose2d [Cin,Cout,2,2]
                    bound = 1.0 / math.sqrt(max(fan_in, 1))
                    p.uniform_(-bound, bound)
            else:
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))

    # ------------------------------------------------------------------ state_dict compatibility
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        if prefix == "":
            for k in list(state_dict.keys()):
                ck = canonical_key(k)
                if ck != k:
                    state_dict[ck] = state_dict.pop(k)
            # temperature may arrive as [1,heads,1,1] (root model.py `scale`)
            for k, v in list(state_dict.items()):
                if k.endswith("attn.temperature") and v.dim() == 4:
                    state_dict[k] = v.reshape(v.shape[1], 1, 1)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    # ------------------------------------------------------------------ runtime
    def _state_for(self, device: torch.device) -> _DeviceState:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        with self._rt_lock:
            st = self._rt.get(idx)
            if st is None:
                st = _DeviceState()
                cfg = self._config()
                _lib.check(_lib.load().rf_create(C.byref(cfg), C.byref(st.handle)), "rf_create")
                self._rt[idx] = st
        return st

    def invalidate_packed(self) -> None:
        """Force the next ``forward`` to re-register and repack the weights.

        The repack check compares ``(data_ptr, _version)`` of every parameter.  ``_version`` does not move for writes
        through ``p.data`` (``p.data.copy_()`` / ``p.data.add_()``: EMA weight swaps, manual initialisation), so after such
        a write call this method -- ``load_state_dict``, ``.to()``, optimiser steps and ``torch.no_grad()`` in-place
        updates are detected without it."""
        for st in self._rt.values():
            st.signature = None

    # The runtime state (library handles, packed weights, workspace, lock) is per process and per device: it is dropped
    # by copy.deepcopy / pickle / torch.save(model) and rebuilt lazily by the first forward of the copy.
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_rt"] = {}
        state["_rt_lock"] = None
        state["_rt_owner"] = None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._rt = {}
        self._rt_lock = threading.Lock()
        self._rt_owner = id(self)

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__setstate__({k: copy.deepcopy(v, memo) for k, v in self.__getstate__().items()})
        return new

    def _sync_params(self, st: _DeviceState, device: torch.device, pack: bool = True) -> None:
        """Register the parameter pointers with the handle and (``pack=True``, the inference forward) repack the weights when
        any of them changed.  ``pack=False`` is the training step's call: ``rf_train_step`` reads the raw weights, so it only
        needs the pointers -- registered once as long as they do not move -- and leaves the packed copies stale (the next
        inference forward repacks: ``signature`` stays ``None``)."""
        params = dict(self.named_parameters())
        if len(params) < len(self._param_names):
            # nn.DataParallel replicas carry their weights as plain tensors, not Parameters (train.py:108-111 is the
            # reference's only multi-device construct); this package shards across PROCESSES instead (tiling.py)
            raise RuntimeError("RawFormer (HIP) cannot run as an nn.DataParallel replica: use one process per GPU "
                               "(bayer_low_light_image_enhancement_amd.tiling / torch.distributed)")
        sig = tuple((p.data_ptr(), p._version) for p in (params[k] for k in self._param_names))
        ptrs = tuple(s[0] for s in sig)
        if sig == st.signature or (not pack and ptrs == st.ptrs):
            return
        lib = _lib.load()
        for k in self._param_names:
            p = params[k]
            if p.device != device:
                raise RuntimeError(f"parameter {k} is on {p.device} but the input is on {device}; call model.to(device)")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError(f"parameter {k} must be contiguous float32")
            shape = (C.c_int64 * max(p.dim(), 1))(*p.shape)
            _lib.check(lib.rf_set_param(st.handle, k.encode(), C.c_void_p(p.data_ptr()), shape, p.dim()), "rf_set_param")
        st.ptrs = ptrs
        if not pack:
            st.signature = None
            return
        sz = C.c_size_t()
        _lib.check(lib.rf_packed_bytes(st.handle, C.byref(sz)), "rf_packed_bytes")
        if st.packed is None or st.packed.numel() < sz.value or st.packed.device != device:
            st.packed = torch.empty(sz.value, dtype=torch.uint8, device=device)
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        _lib.check(lib.rf_pack_params(st.handle, C.c_void_p(st.packed.data_ptr()), sz.value, stream), "rf_pack_params")
        st.signature = sig

    def _run(self, x: torch.Tensor, packed_input: bool) -> torch.Tensor:
        if x.device.type != "cuda":
            raise RuntimeError("RawFormer (HIP) needs a ROCm device tensor: there is no CPU path in this package "
                               f"(input is on {x.device})")
        if x.dim() != 4:
            raise RuntimeError(f"expected a 4-D input [B,C,H,W], got {tuple(x.shape)}")
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("RawFormer.forward (HIP) is the inference path and builds no autograd graph: call model.eval() or wrap "
                               "in torch.no_grad(); to train use bayer_low_light_image_enhancement_amd.train.Trainer(model).step(x, gt) "
                               "(rf_train_step: forward, loss, explicit backward, Adam / AdamW)")
        x = x.detach()
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        b, c, h, w = x.shape
        if packed_input:
            if c != 4 * self.inp_channels:
                raise RuntimeError(f"expected {4 * self.inp_channels} packed channels, got {c}")
            H, W = h, w
        else:
            if c != self.inp_channels:
                raise RuntimeError(f"Given a mosaic with {c} channels, expected {self.inp_channels}")
            if h % 2 or w % 2:
                raise RuntimeError(f"mosaic size {h}x{w} must be even")
            H, W = h // 2, w // 2
        if H % 8 or W % 8:
            raise RuntimeError(f"mosaic size must be divisible by 16 (three 2x down-samplings after the Bayer pack); "
                               f"got {2 * H}x{2 * W}")
        lib = _lib.load()
        with torch.cuda.device(x.device):
            st = self._state_for(x.device)
            self._sync_params(st, x.device)
            sz = C.c_size_t()
            _lib.check(lib.rf_workspace_bytes(st.handle, b, H, W, C.byref(sz)), "rf_workspace_bytes")
            if st.workspace is None or st.workspace.numel() < sz.value:
                st.workspace = None
                st.workspace = torch.empty(sz.value, dtype=torch.uint8, device=x.device)
            out = torch.empty((b, self.out_channels, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            _lib.check(lib.rf_forward(st.handle, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                      C.c_void_p(st.workspace.data_ptr()), st.workspace.numel(), b, H, W,
                                      int(packed_input), stream), "rf_forward")
        return out

    def forward_window(self, x: torch.Tensor, y_lo: int, y_hi: int, total_rows: int, group=None) -> torch.Tensor:
        """Forward of one rank's WINDOW of a spatially sharded frame (``tiling.forward_full_frame_exact``; C ABI
        ``rf_set_shard``).  ``x``: the window of the mosaic ``[B,1,2Hl,2Wl]``, the same shape on every rank of ``group``;
        ``[y_lo, y_hi)``: this rank's interior rows inside the window and ``total_rows`` the frame's height, in PACKED rows
        (mosaic rows / 2, multiples of 8).  The channel attention's Gram statistics and FLCA's squeeze-excite pooling are
        taken over interior rows and all-reduced over ``group`` (RCCL with the ``nccl`` backend), so interior rows of the
        result equal the whole-frame forward up to summation order when the window reaches ``tiling.HALO_ROWS`` rows beyond
        the interior (or the frame border).  Rows outside the interior are meaningless."""
        import torch.distributed as dist

        if self.variant == "truecolor":
            raise RuntimeError("forward_window: variants 'flca' and 'plain' only")
        lib = _lib.load()
        failure: List[BaseException] = []

        def allreduce(_user, buf, n, op, _stream):
            try:
                ws = self._state_for(x.device).workspace
                off = buf - ws.data_ptr()
                if off < 0 or off % 4 or off + 4 * n > ws.numel():
                    raise RuntimeError("all-reduce buffer outside the workspace")
                view = ws[off: off + 4 * n].view(torch.float32)
                dist.all_reduce(view, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=group)      # ordered on torch's current stream = `stream`
            except BaseException as e:  # noqa: BLE001 - a ctypes callback cannot raise; re-raised below
                failure.append(e)

        cb = _lib.ALLREDUCE_FN(allreduce)
        with torch.cuda.device(x.device):
            st = self._state_for(x.device)
            # the workspace must exist (and not move) before the callback sees pointers into it
            sz = C.c_size_t()
            _lib.check(lib.rf_workspace_bytes(st.handle, x.shape[0], x.shape[2] // 2, x.shape[3] // 2, C.byref(sz)), "rf_workspace_bytes")
            if st.workspace is None or st.workspace.numel() < sz.value:
                st.workspace = None
                st.workspace = torch.empty(sz.value, dtype=torch.uint8, device=x.device)
            _lib.check(lib.rf_set_shard(st.handle, int(y_lo), int(y_hi), int(total_rows), cb, None), "rf_set_shard")
            try:
                out = self._run(x, packed_input=False)
            finally:
                lib.rf_set_shard(st.handle, 0, 0, 0, _lib.ALLREDUCE_FN(0), None)
        if failure:
            raise failure[0]
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Mosaic ``[B, inp_channels, 2H, 2W]`` -> ``[B, out_channels, 2H, 2W]`` (test.py:116)."""
        return self._run(x, packed_input=False)

    def forward_packed(self, x4: torch.Tensor) -> torch.Tensor:
        """Packed RGGB ``[B, 4, H, W]`` (the ``downshuffle`` already applied) -> ``[B, 3, 2H, 2W]``."""
        return self._run(x4, packed_input=True)

    def forward_stage(self, stage: int, x: torch.Tensor, packed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One ``conv_tran<stage>`` (``Conv_Transformer``) exactly as ``forward`` schedules it.  ``x``: the stage input
        ``[B, dim*2^l, H>>l, W>>l]``; ``packed``: the packed frame ``[B,4,H,W]`` the FLCA guidance is derived from
        (``variant='flca'``; for ``'plain'`` pass ``None`` and H, W are taken from ``x``)."""
        if x.device.type != "cuda":
            raise RuntimeError("RawFormer (HIP) needs a ROCm device tensor: there is no CPU path in this package")
        lvl = stage - 1 if stage <= 4 else 7 - stage
        x = x.detach().float().contiguous()
        b, c, hh, ww = x.shape
        if c != self.dim << lvl:
            raise RuntimeError(f"stage {stage} expects {self.dim << lvl} channels, got {c}")
        H, W = hh << lvl, ww << lvl
        if packed is not None:
            packed = packed.detach().float().contiguous()
            if tuple(packed.shape) != (b, 4, H, W):
                raise RuntimeError(f"packed frame must be {(b, 4, H, W)}, got {tuple(packed.shape)}")
        lib = _lib.load()
        with torch.cuda.device(x.device):
            st = self._state_for(x.device)
            self._sync_params(st, x.device)
            sz = C.c_size_t()
            _lib.check(lib.rf_workspace_bytes(st.handle, b, H, W, C.byref(sz)), "rf_workspace_bytes")
            if st.workspace is None or st.workspace.numel() < sz.value:
                st.workspace = None
                st.workspace = torch.empty(sz.value, dtype=torch.uint8, device=x.device)
            out = torch.empty_like(x)
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            _lib.check(lib.rf_forward_stage(st.handle, stage, C.c_void_p(x.data_ptr()),
                                            C.c_void_p(packed.data_ptr() if packed is not None else None),
                                            C.c_void_p(out.data_ptr()), C.c_void_p(st.workspace.data_ptr()),
                                            st.workspace.numel(), b, H, W, stream), "rf_forward_stage")
        return out

    def workspace_bytes(self, batch: int, H: int, W: int) -> int:
        probe, sz = C.c_void_p(), C.c_size_t()
        cfg = self._config()
        lib = _lib.load()
        _lib.check(lib.rf_create(C.byref(cfg), C.byref(probe)), "rf_create")
        try:
            _lib.check(lib.rf_workspace_bytes(probe, batch, H, W, C.byref(sz)), "rf_workspace_bytes")
        finally:
            lib.rf_destroy(probe)
        return sz.value

    def __del__(self):
        try:
            if getattr(self, "_rt_owner", None) != id(self):
                return
            lib = _lib.load()
            for st in getattr(self, "_rt", {}).values():
                if st.handle:
                    lib.rf_destroy(st.handle)
                    st.handle = C.c_void_p()
        except Exception:
            pass
