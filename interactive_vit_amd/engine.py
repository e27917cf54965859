"""ctypes binding of the C ABI in ``include/ivit.h`` (libivit.so: hand-written gfx950 kernels).

There is no CPU or eager-PyTorch fallback: if the shared library is missing, fails to load, or an
entry point is absent, construction raises.  torch is used for device memory and streams only -
tensors cross the ABI as raw ``data_ptr()`` addresses.
"""
from __future__ import annotations

import ctypes
import weakref
import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .vit_config import VitConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libivit.so")

# every symbol include/ivit.h declares (tests check the built library exports each one)
ABI_SYMBOLS = (
    "ivit_abi_version", "ivit_build_info", "ivit_last_error", "ivit_stage_count", "ivit_stage_shape",
    "ivit_unfold_offset", "ivit_create", "ivit_destroy", "ivit_set_weight", "ivit_weights_ready", "ivit_ln_fold",
    "ivit_forward_host", "ivit_forward_host_chained", "ivit_forward_device", "ivit_preprocess", "ivit_preprocess_host", "ivit_attention_map", "ivit_attention_map_host",
    "ivit_fp8_calibrate", "ivit_fp8_scales", "ivit_debug_unfold", "ivit_profile_enable",
    "ivit_profile_reset", "ivit_profile_class_count", "ivit_profile_class_name", "ivit_profile_read",
    "ivit_profile_kernel_count", "ivit_profile_kernel_read", "ivit_debug_layer_tap", "ivit_debug_weight_fp8", "ivit_ln_fold_calibrate",
    "ivit_forward_host_async", "ivit_host_wait", "ivit_comm_unique_id", "ivit_comm_init", "ivit_allgather_cls",
    "ivit_forward_device_packed", "ivit_shard_layout", "ivit_allgather_rows", "ivit_layer_with_attn", "ivit_layer_with_attn_host", "ivit_fused_mlp", "ivit_split_set", "ivit_ln_fold_centres",
)


class IvitConfigC(ctypes.Structure):
    _fields_ = [("image", ctypes.c_int32), ("patch", ctypes.c_int32), ("dim", ctypes.c_int32),
                ("heads", ctypes.c_int32), ("layers", ctypes.c_int32), ("mlp", ctypes.c_int32),
                ("classes", ctypes.c_int32), ("ln_eps", ctypes.c_float), ("device", ctypes.c_int32),
                ("max_batch", ctypes.c_int32), ("precision", ctypes.c_int32)]


ABI_VERSION = 10
PRECISIONS = {"bf16": 0, "fp8": 1, "f16": 2, "f16x": 3, "fp8m": 4}


_lib = None
_lib_lock = threading.Lock()


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """Loads libivit.so (once).  Raises if it is absent - the product path has no fallback."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or os.environ.get("IVIT_LIB", LIB_PATH)
        if not os.path.exists(p):
            raise RuntimeError(
                f"libivit.so not found at {p}: build it with `python -m interactive_vit_amd.build` "
                "(hipcc, gfx950).  The ViT operators have no CPU fallback.")
        lib = ctypes.CDLL(p)
        for sym in ABI_SYMBOLS:
            if not hasattr(lib, sym):
                raise RuntimeError(f"{p} does not export {sym}")
        c_i, c_p, c_i64 = ctypes.c_int, ctypes.c_void_p, ctypes.c_int64
        cfgp = ctypes.POINTER(IvitConfigC)
        lib.ivit_abi_version.restype = c_i
        lib.ivit_build_info.restype = ctypes.c_char_p
        lib.ivit_last_error.restype = ctypes.c_char_p
        lib.ivit_stage_count.argtypes = [cfgp]
        lib.ivit_stage_shape.argtypes = [cfgp, c_i, c_i, ctypes.POINTER(c_i64)]
        lib.ivit_unfold_offset.argtypes = [ctypes.c_int32] * 4
        lib.ivit_unfold_offset.restype = c_i64
        lib.ivit_create.argtypes = [cfgp, ctypes.POINTER(c_p)]
        lib.ivit_destroy.argtypes = [c_p]
        lib.ivit_destroy.restype = None
        lib.ivit_set_weight.argtypes = [c_p, ctypes.c_char_p, c_p, ctypes.POINTER(c_i64), c_i]
        lib.ivit_weights_ready.argtypes = [c_p]
        lib.ivit_ln_fold.argtypes = [c_p, c_i]
        lib.ivit_ln_fold.restype = c_i
        lib.ivit_fused_mlp.argtypes = [c_p, c_i]
        lib.ivit_fused_mlp.restype = c_i
        lib.ivit_split_set.argtypes = [c_p]
        lib.ivit_split_set.restype = c_i
        lib.ivit_forward_host.argtypes = [c_p, c_i, c_i, c_i, c_p, c_p, c_i64]
        lib.ivit_forward_host_chained.argtypes = [c_p, c_i, c_i, c_i, c_p, c_p, c_i64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
        lib.ivit_forward_host_async.argtypes = [c_p, c_i, c_i, c_i, c_p, c_p, c_i64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                                ctypes.POINTER(ctypes.c_uint64)]
        lib.ivit_host_wait.argtypes = [c_p, ctypes.c_uint64]
        lib.ivit_comm_unique_id.argtypes = [c_p]
        lib.ivit_comm_init.argtypes = [c_p, c_p, c_i, c_i]
        lib.ivit_allgather_cls.argtypes = [c_p, c_p, c_p, c_i64, c_p]
        lib.ivit_forward_device_packed.argtypes = [c_p, c_i, c_i, c_p, c_p, c_i64, c_p]
        lib.ivit_shard_layout.argtypes = [c_i64, c_i, c_i, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]
        lib.ivit_allgather_rows.argtypes = [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p]
        lib.ivit_layer_with_attn.argtypes = [c_p, c_i, c_i, c_p, c_p, c_p, c_p]
        lib.ivit_layer_with_attn_host.argtypes = [c_p, c_i, c_i, c_p, c_p, c_i64, c_p, c_i64]
        lib.ivit_preprocess_host.argtypes = [c_p, c_i, c_p, c_i, c_i, c_p, c_i64, ctypes.POINTER(ctypes.c_uint64)]
        lib.ivit_preprocess.argtypes = [c_p, c_i, c_p, c_i, c_i, c_p, c_p]
        lib.ivit_forward_device.argtypes = [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p]
        lib.ivit_attention_map.argtypes = [c_p, c_i, c_i, c_p, c_p, c_p]
        lib.ivit_attention_map_host.argtypes = [c_p, c_i, c_i, c_p, c_p, c_i64]
        lib.ivit_fp8_calibrate.argtypes = [c_p, c_i, c_p, c_p]
        lib.ivit_fp8_scales.argtypes = [c_p, ctypes.POINTER(ctypes.c_float), c_i]
        lib.ivit_debug_unfold.argtypes = [c_p, c_i, c_p, c_p, c_i, c_p]
        lib.ivit_profile_enable.argtypes = [c_p, c_i]
        lib.ivit_profile_reset.argtypes = [c_p]
        lib.ivit_profile_class_name.argtypes = [c_i]
        lib.ivit_profile_class_name.restype = ctypes.c_char_p
        lib.ivit_profile_read.argtypes = [c_p, c_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64),
                                          ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        lib.ivit_debug_layer_tap.argtypes = [c_p, c_i, c_i, c_p, c_i, c_p, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_i), c_p]
        lib.ivit_debug_weight_fp8.argtypes = [c_p, c_i, c_i, c_p, c_i64, ctypes.POINTER(ctypes.c_float), c_i, ctypes.POINTER(c_i),
                                              ctypes.POINTER(c_i), ctypes.POINTER(c_i)]
        lib.ivit_ln_fold_calibrate.argtypes = [c_p, c_i, c_p, ctypes.c_float, ctypes.POINTER(ctypes.c_float), c_p]
        lib.ivit_profile_kernel_count.argtypes = [c_p]
        lib.ivit_profile_kernel_read.argtypes = [c_p, c_i, ctypes.c_char_p, c_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64),
                                                 ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        if path is None:
            _lib = lib
        return lib


def shard_layout(total: int, world: int, rank: int):
    """(begin, rows, padded_rows) of rank's shard of `total` rows - the engine's own rule (include/ivit.h: ivit_shard_layout; host
    arithmetic, no GPU), which interactive_vit_amd.sharding.shard_range restates for the torch.distributed path."""
    lib = load_library()
    b, r, pz = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    if lib.ivit_shard_layout(int(total), int(world), int(rank), ctypes.byref(b), ctypes.byref(r), ctypes.byref(pz)) != 0:
        raise ValueError(lib.ivit_last_error().decode("utf-8", "replace"))
    return b.value, r.value, pz.value


class PendingTensor(torch.Tensor):
    """A CPU f32 tensor in page-locked memory whose bytes the engine's D2H copy may still be writing
    (include/ivit.h: ivit_forward_host_async).  Any torch operation on it - ``numpy()``, arithmetic, indexing, ``shape`` -
    first waits for that copy (once), then runs on the plain tensor; the engine itself recognises it and continues from
    the device-resident copy without touching the host bytes.  So in a chain of nodes (Context.compute hands node k's
    output to node k+1 by reference and Response reads all outputs only at the end, reference main/context.py:143-147,
    main/message.py:77-83) the copy of node k runs behind the kernels of node k+1."""

    @staticmethod
    def wrap(data: torch.Tensor, engine: "Engine", ticket: int) -> "PendingTensor":
        t = torch.Tensor._make_subclass(PendingTensor, data)
        t._ivit_engine = weakref.ref(engine)
        t._ivit_ticket = int(ticket)
        t._ivit_done = False
        t._ivit_ptr = data.data_ptr()
        t._ivit_shape = tuple(data.shape)
        return t

    def _ivit_wait(self) -> None:
        if not getattr(self, "_ivit_done", True):
            eng = self._ivit_engine()
            if eng is not None and eng._h is not None and eng._h.value:
                eng._check(eng.lib.ivit_host_wait(eng._h, ctypes.c_uint64(self._ivit_ticket)))
                eng._retire(self._ivit_ticket)
            self._ivit_done = True

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}

        def sync(a):
            if isinstance(a, PendingTensor):
                a._ivit_wait()
            elif isinstance(a, (list, tuple)):
                for b in a:
                    sync(b)
            elif isinstance(a, dict):
                for b in a.values():
                    sync(b)

        sync(args)
        sync(kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


class EngineError(Exception):
    """A non-zero return from libivit; str(e) is ivit_last_error() (-> HTTP 400 body, views.py:40-42)."""


def _config_c(cfg: VitConfig, device: int = 0, max_batch: int = 1, precision: str = "bf16") -> IvitConfigC:
    return IvitConfigC(cfg.image, cfg.patch, cfg.dim, cfg.heads, cfg.layers, cfg.mlp, cfg.classes,
                       cfg.ln_eps, device, max_batch, PRECISIONS[precision])


def stage_names(cfg: VitConfig) -> List[str]:
    """Stage index -> node suffix (the order of include/ivit.h)."""
    return (["transform", "conv_proj", "tokens"] + [f"encoder.layers.{i}" for i in range(cfg.layers)]
            + ["encoder.ln", "cls", "heads"])


def stage_shape(cfg: VitConfig, stage: int, which: int) -> Tuple[int, ...]:
    """Per-image input (which=0) / output (which=1) shape of a stage, from the library's own table."""
    lib = load_library()
    c = _config_c(cfg)
    dims = (ctypes.c_int64 * 3)()
    nd = lib.ivit_stage_shape(ctypes.byref(c), stage, which, dims)
    if nd < 0:
        raise EngineError(f"stage {stage} out of range")
    return tuple(int(dims[i]) for i in range(nd))


def unfold_offset(image: int, patch: int, n: int, k: int) -> int:
    return int(load_library().ivit_unfold_offset(image, patch, n, k))


class Engine:
    """One ViT engine instance bound to one GPU (owns bf16 weights + workspaces on that device)."""

    def __init__(self, cfg: VitConfig, state_dict: Dict[str, torch.Tensor], device: int = 0, max_batch: int = 1,
                 precision: str = "bf16"):
        self.lib = load_library()
        if self.lib.ivit_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libivit.so has ABI {self.lib.ivit_abi_version()}, this binding needs {ABI_VERSION}: rebuild")
        self.cfg = cfg
        self.ln_fold = False      # set after creation: does the engine fold LayerNorm into the next GEMM (ivit_ln_fold)
        self._pin = os.environ.get("IVIT_PINNED_OUTPUTS", "1") != "0"
        # host path: return lazily-synchronised tensors (the D2H copy of a node overlaps the next node's kernels)
        self._async = self._pin and os.environ.get("IVIT_ASYNC_OUTPUTS", "1") != "0"
        self._last_out = None     # (weakref to the last host-path output, its version counter, its residency token)
        self._held = {}           # ticket -> page-locked result buffer the copy stream may still be writing (see _hold)
        self._held_lock = threading.Lock()
        self.device = int(device)
        self.max_batch = int(max_batch)
        self.precision = precision
        self.stages = stage_names(cfg)
        self._h = ctypes.c_void_p()
        c = _config_c(cfg, self.device, self.max_batch, precision)
        self._check(self.lib.ivit_create(ctypes.byref(c), ctypes.byref(self._h)))
        self.ln_fold = bool(self.lib.ivit_ln_fold(self._h, 1))      # small batches; ln_fold_for(batch) for a given size
        try:
            for name, t in state_dict.items():
                self.set_weight(name, t)
            self._check(self.lib.ivit_weights_ready(self._h))
        except Exception:
            self.close()
            raise

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise EngineError(self.lib.ivit_last_error().decode("utf-8", "replace"))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ivit_destroy(self._h)      # synchronises the device: every copy has landed
            self._h = ctypes.c_void_p()
        if getattr(self, "_held", None):
            with self._held_lock:
                self._held.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weight(self, name: str, t: torch.Tensor) -> None:
        t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
        shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
        self._check(self.lib.ivit_set_weight(self._h, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()))

    # -- shapes --------------------------------------------------------------------------------
    def stage_index(self, suffix: str) -> int:
        return self.stages.index(suffix)

    def in_shape(self, stage: int) -> Tuple[int, ...]:
        return self._stage_shape(stage, 0)

    def out_shape(self, stage: int) -> Tuple[int, ...]:
        return self._stage_shape(stage, 1)

    def _stage_shape(self, stage: int, which: int) -> Tuple[int, ...]:
        # the library's table, asked once per stage (two library calls per node call were 10 % of an interactive request's host time)
        cache = self.__dict__.setdefault("_shape_cache", {})
        key = (stage, which)
        if key not in cache:
            cache[key] = stage_shape(self.cfg, stage, which)
        return cache[key]

    def _split_batch(self, x: torch.Tensor, stage: int) -> Tuple[int, bool]:
        want = self.in_shape(stage)
        shape = x._ivit_shape if isinstance(x, PendingTensor) else tuple(x.shape)   # (shape of a pending tensor: no wait)
        if shape == want:
            return 1, False
        if len(shape) == len(want) + 1 and shape[1:] == want:
            return int(shape[0]), True
        raise EngineError(f"{self.cfg.name}:{self.stages[stage]} expects input {list(want)} "
                          f"(optionally with a leading batch axis), got {list(x.shape)}")

    # -- page-locked result buffers of asynchronous host calls ------------------------------------
    # The D2H copy that fills such a buffer runs on the engine's own copy stream, which torch's caching host allocator knows nothing
    # about: a caller that drops its PendingTensor before anything waited on it (a later node raised and the request ended in HTTP 400,
    # or an output nobody reads) would hand the block back to the pool while the DMA is still writing it.  So the engine holds every
    # buffer until its ticket has completed: tickets complete in order (one copy stream), a wait on ticket t retires every ticket <= t,
    # and the number of held buffers is bounded by waiting for the oldest.
    _MAX_HELD = 32

    def _hold(self, ticket: int, buf: torch.Tensor) -> None:
        with self._held_lock:
            self._held[ticket] = buf
            oldest = min(self._held) if len(self._held) > self._MAX_HELD else None
        if oldest is not None and self._h is not None and self._h.value:
            self._check(self.lib.ivit_host_wait(self._h, ctypes.c_uint64(oldest)))
            self._retire(oldest)

    def _retire(self, ticket: int) -> None:
        with self._held_lock:
            for t in [t for t in self._held if t <= ticket]:
                del self._held[t]

    # -- forward -------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, begin: int, end: int, want_cls: bool = False):
        """Runs stages [begin, end).  CPU input -> CPU f32 output (host path); CUDA input -> CUDA
        f32 output on the current torch stream (device path, asynchronous)."""
        batch, batched = self._split_batch(x, begin)
        oshape = self.out_shape(end - 1)
        full = ((batch,) + oshape) if batched else oshape
        if isinstance(x, PendingTensor) or x.device.type == "cpu":
            # node chains: when `x` is the very tensor the previous host call returned (Context.compute
            # hands outputs on by reference) and nobody wrote to it since, its device-resident copy is
            # consumed instead of uploading it again (include/ivit.h: ivit_forward_host_chained)
            if isinstance(x, PendingTensor) and x._ivit_engine() is not self:
                x._ivit_wait()            # another engine's copy stream fills it: nothing of ours is ordered behind that
            with torch._C.DisableTorchFunctionSubclass():      # no torch call below may wait for a pending copy
                token = 0
                last = self._last_out
                if last is not None and last[0]() is x and x._version == last[1]:
                    token = last[2]
                xin = x.detach().to(torch.float32).contiguous()   # a PendingTensor is f32 and contiguous: a view, no bytes read
                in_ptr = xin.data_ptr()
            # page-locked result buffer (torch caches the blocks): the D2H copy runs at DMA speed
            out = torch.empty(full, dtype=torch.float32, pin_memory=self._pin)
            new_token = ctypes.c_uint64(0)
            if self._async:
                # If the token is stale the library uploads `in` on its own stream, i.e. BEHIND the copy that may still be
                # filling a pending `x` (page-locked, stream-ordered) - correct without a host-side wait.  A pageable or
                # converted input was produced by a torch operation, which has waited already.
                ticket = ctypes.c_uint64(0)
                self._check(self.lib.ivit_forward_host_async(self._h, begin, end, batch, ctypes.c_void_p(in_ptr),
                                                             ctypes.c_void_p(out.data_ptr()), out.numel(),
                                                             ctypes.c_uint64(token), ctypes.byref(new_token), ctypes.byref(ticket)))
                res = PendingTensor.wrap(out, self, ticket.value)
                with torch._C.DisableTorchFunctionSubclass():
                    self._last_out = (weakref.ref(res), res._version, new_token.value)
                self._hold(ticket.value, out)
                return res
            self._check(self.lib.ivit_forward_host_chained(self._h, begin, end, batch, ctypes.c_void_p(in_ptr),
                                                           ctypes.c_void_p(out.data_ptr()), out.numel(),
                                                           ctypes.c_uint64(token), ctypes.byref(new_token)))
            self._last_out = (weakref.ref(out), out._version, new_token.value)
            return out
        if x.device.type != "cuda" or (x.device.index or 0) != self.device:
            raise EngineError(f"input lives on {x.device}, engine on cuda:{self.device}")
        xin = x.detach().to(torch.float32).contiguous()
        out = torch.empty(full, dtype=torch.float32, device=x.device)
        cls = torch.empty((batch, self.cfg.dim), dtype=torch.float32, device=x.device) if want_cls else None
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self._check(self.lib.ivit_forward_device(self._h, begin, end, batch, ctypes.c_void_p(xin.data_ptr()),
                                                 ctypes.c_void_p(out.data_ptr()),
                                                 ctypes.c_void_p(cls.data_ptr()) if want_cls else None,
                                                 ctypes.c_void_p(stream)))
        return (out, cls) if want_cls else out

    def forward_into(self, x: torch.Tensor, out: torch.Tensor, cls: Optional[torch.Tensor], batch: int,
                     begin: int, end: int, stream: int) -> None:
        """Zero-overhead device call for the benchmark loop: preallocated buffers, explicit stream."""
        self._check(self.lib.ivit_forward_device(self._h, begin, end, batch, ctypes.c_void_p(x.data_ptr()),
                                                 ctypes.c_void_p(out.data_ptr()),
                                                 ctypes.c_void_p(cls.data_ptr()) if cls is not None else None,
                                                 ctypes.c_void_p(stream)))

    def attention_map(self, layer: int, x: torch.Tensor) -> torch.Tensor:
        """softmax(q k^T / sqrt(dh)) of encoder layer `layer` for a residual-stream input
        [N,D] / [B,N,D]: f32 [heads,N,N] / [B,heads,N,N] (CPU in -> CPU out, CUDA in -> CUDA out)."""
        batch, batched = self._split_batch(x, 3)      # every encoder layer takes [N, D]
        n, hds = self.cfg.tokens, self.cfg.heads
        full = (batch, hds, n, n) if batched else (hds, n, n)
        xin = x.detach().to(torch.float32).contiguous()
        if x.device.type == "cpu":
            out = torch.empty(full, dtype=torch.float32)
            self._check(self.lib.ivit_attention_map_host(self._h, layer, batch, ctypes.c_void_p(xin.data_ptr()),
                                                         ctypes.c_void_p(out.data_ptr()), out.numel()))
            return out
        out = torch.empty(full, dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self._check(self.lib.ivit_attention_map(self._h, layer, batch, ctypes.c_void_p(xin.data_ptr()),
                                                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)))
        return out

    def preprocess(self, x: torch.Tensor) -> torch.Tensor:
        """Raw image(s) [3,H,W] / [B,3,H,W] in [0,1], any size -> the normalised [.., 3,S,S] input of
        ``conv_proj`` (include/ivit.h: ivit_preprocess*).  CPU in -> CPU out, CUDA in -> CUDA out."""
        if x.dim() not in (3, 4) or x.shape[-3] != 3:
            raise EngineError(f"{self.cfg.name}:preprocess expects [3,H,W] or [B,3,H,W], got {list(x.shape)}")
        batched = x.dim() == 4
        batch = int(x.shape[0]) if batched else 1
        h, w = int(x.shape[-2]), int(x.shape[-1])
        s_ = self.cfg.image
        full = (batch, 3, s_, s_) if batched else (3, s_, s_)
        xin = x.detach().to(torch.float32).contiguous()
        if x.device.type == "cpu":
            out = torch.empty(full, dtype=torch.float32, pin_memory=self._pin)
            tok = ctypes.c_uint64(0)
            self._check(self.lib.ivit_preprocess_host(self._h, batch, ctypes.c_void_p(xin.data_ptr()), h, w,
                                                      ctypes.c_void_p(out.data_ptr()), out.numel(), ctypes.byref(tok)))
            self._last_out = (weakref.ref(out), out._version, tok.value)
            return out
        if x.device.type != "cuda" or (x.device.index or 0) != self.device:
            raise EngineError(f"input lives on {x.device}, engine on cuda:{self.device}")
        out = torch.empty(full, dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self._check(self.lib.ivit_preprocess(self._h, batch, ctypes.c_void_p(xin.data_ptr()), h, w,
                                             ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)))
        return out

    # -- multi-GPU: the one collective of the path, issued by the engine through RCCL ----------------
    def comm_init(self, rank: int, world: int, broadcast_bytes) -> None:
        """Binds this engine to an RCCL communicator of `world` ranks (include/ivit.h: ivit_comm_*).  `broadcast_bytes(b)`
        must return rank 0's bytes object on every rank (e.g. through torch.distributed's store / object broadcast)."""
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            self._check(self.lib.ivit_comm_unique_id(buf))
        ident = broadcast_bytes(bytes(buf.raw))
        assert len(ident) == 128
        self._check(self.lib.ivit_comm_init(self._h, ctypes.c_char_p(ident), int(rank), int(world)))
        self.comm_world = int(world)

    def allgather(self, local: torch.Tensor, out: torch.Tensor, stream: Optional[int] = None) -> torch.Tensor:
        """out[r * b : (r + 1) * b] = rank r's `local` ([b, width] f32, equal on every rank), enqueued on `stream`."""
        assert local.is_cuda and out.is_cuda and local.dtype == out.dtype == torch.float32 and local.is_contiguous() and out.is_contiguous()
        assert out.numel() == local.numel() * self.comm_world
        st = stream if stream is not None else torch.cuda.current_stream(local.device).cuda_stream
        self._check(self.lib.ivit_allgather_cls(self._h, ctypes.c_void_p(local.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                local.numel(), ctypes.c_void_p(st)))
        return out

    def forward_packed(self, x: torch.Tensor, packed: torch.Tensor, batch: int, begin: int, stream: int) -> None:
        """The multi-GPU step's forward (include/ivit.h: ivit_forward_device_packed): stages [begin, end of model) with row b of `packed`
        ([batch, >= classes + dim] f32, CUDA) written in place as [logits | class-token features] - no copy kernels before the collective."""
        assert packed.is_cuda and packed.dtype == torch.float32 and packed.dim() == 2 and packed.stride(1) == 1
        self._check(self.lib.ivit_forward_device_packed(self._h, begin, batch, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(packed.data_ptr()),
                                                        packed.stride(0), ctypes.c_void_p(stream)))

    def allgather_rows(self, local: torch.Tensor, total_rows: int, out: torch.Tensor, stream: Optional[int] = None) -> torch.Tensor:
        """ONE RCCL all-gather of every rank's [rows_local, width] block into out [total_rows, width], image order; ragged shards
        (total_rows % world != 0) are padded inside the engine (include/ivit.h: ivit_allgather_rows)."""
        assert local.is_cuda and out.is_cuda and local.dtype == out.dtype == torch.float32 and local.is_contiguous() and out.is_contiguous()
        assert out.shape[0] == total_rows and out.shape[1] == local.shape[1]
        st = stream if stream is not None else torch.cuda.current_stream(local.device).cuda_stream
        self._check(self.lib.ivit_allgather_rows(self._h, ctypes.c_void_p(local.data_ptr()), local.shape[0], local.shape[1], int(total_rows),
                                                 ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st)))
        return out

    def ln_fold_for(self, batch: int) -> bool:
        """Does a forward of `batch` images fold the encoder's LayerNorms into the following GEMMs
        (include/ivit.h: ivit_ln_fold)?  What the rounding-aware oracle has to mirror for that call."""
        return bool(self.lib.ivit_ln_fold(self._h, int(batch)))

    def fused_mlp_for(self, batch: int) -> int:
        """0: the MLP of a forward of `batch` images runs as two GEMM launches; else the form of the fused MLP kernel (include/ivit.h: ivit_fused_mlp).
        Either way the same bytes come out: the oracle's rounding-aware mode needs no switch for it."""
        return int(self.lib.ivit_fused_mlp(self._h, int(batch)))

    def calibrate_ln_fold(self, images: torch.Tensor, threshold: float = 0.5) -> float:
        """Calibration and guard of the LayerNorm fold for THIS weight set (include/ivit.h: ivit_ln_fold_calibrate): one forward of
        `images` ([B,3,S,S] in [0,1]) with LayerNorm kernels; the per-channel means of every LayerNorm input become the centre vectors
        the engine subtracts before rounding its 16-bit copies (IVIT_FOLD_CENTRE=0: none), and the guard statistic of the centred copy
        is returned - the engine keeps the fold only if that is <= threshold.  `ln_fold_ratio_plain` afterwards holds the statistic of
        the plain copy (max |mean| / std over the rows), `ln_centres()` the vectors."""
        batch, _ = self._split_batch(images, 0)
        xin = images.detach().to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()
        stream = torch.cuda.current_stream(xin.device).cuda_stream
        ratio = ctypes.c_float(0.0)
        self._check(self.lib.ivit_ln_fold_calibrate(self._h, batch, ctypes.c_void_p(xin.data_ptr()), ctypes.c_float(threshold),
                                                    ctypes.byref(ratio), ctypes.c_void_p(stream)))
        self.ln_fold = bool(self.lib.ivit_ln_fold(self._h, 1))
        plain = ctypes.c_float(0.0)
        self._check(self.lib.ivit_ln_fold_centres(self._h, None, 0, None, ctypes.byref(plain)))
        self.ln_fold_ratio_plain = float(plain.value)
        return float(ratio.value)

    def ln_centres(self) -> Optional[torch.Tensor]:
        """The centre vectors of the LayerNorm inputs ([2 * layers, dim] f32: LN1 of layer 0, LN2 of layer 0, LN1 of layer 1, ...) when the
        engine's 16-bit copies are centred (after calibrate_ln_fold), else None - what the rounding-aware oracle mirrors (LN_CENTRE)."""
        n = 2 * self.cfg.layers * self.cfg.dim
        buf = (ctypes.c_float * n)()
        on = ctypes.c_int(0)
        self._check(self.lib.ivit_ln_fold_centres(self._h, buf, n, ctypes.byref(on), None))
        if not on.value:
            return None
        return torch.frombuffer(buf, dtype=torch.float32).clone().reshape(2 * self.cfg.layers, self.cfg.dim)

    def layer_with_attn(self, layer: int, x: torch.Tensor):
        """Encoder layer `layer` on a residual-stream input [N,D] / [B,N,D] with its attention map as a second result
        (include/ivit.h: ivit_layer_with_attn): (out like the layer node, f32 [heads,N,N] / [B,heads,N,N] from the layer's own q|k|v)."""
        batch, batched = self._split_batch(x, 3)
        n, d, hds = self.cfg.tokens, self.cfg.dim, self.cfg.heads
        xin = x.detach().to(torch.float32).contiguous()
        oshape = (batch, n, d) if batched else (n, d)
        ashape = (batch, hds, n, n) if batched else (hds, n, n)
        if x.device.type == "cpu":
            out = torch.empty(oshape, dtype=torch.float32)
            attn = torch.empty(ashape, dtype=torch.float32)
            self._check(self.lib.ivit_layer_with_attn_host(self._h, layer, batch, ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                           out.numel(), ctypes.c_void_p(attn.data_ptr()), attn.numel()))
            return out, attn
        if x.device.type != "cuda" or (x.device.index or 0) != self.device:
            raise EngineError(f"input lives on {x.device}, engine on cuda:{self.device}")
        out = torch.empty(oshape, dtype=torch.float32, device=x.device)
        attn = torch.empty(ashape, dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self._check(self.lib.ivit_layer_with_attn(self._h, layer, batch, ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                  ctypes.c_void_p(attn.data_ptr()), ctypes.c_void_p(stream)))
        return out, attn

    def run_node_multi(self, suffix: str, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Nodes with more than one output channel: `encoder.layers.<i>.with_attn` -> {"o": layer output, "attn": attention map}."""
        if suffix.endswith(".with_attn"):
            out, attn = self.layer_with_attn(int(suffix.split(".")[-2]), x)
            return {"o": out, "attn": attn}
        return {"o": self.run_node(suffix, x)}

    def run_node(self, suffix: str, x: torch.Tensor) -> torch.Tensor:
        if suffix == "preprocess":
            return self.preprocess(x)
        if suffix == "forward":
            return self.forward(x, 0, len(self.stages))
        if suffix.endswith(".attn"):
            return self.attention_map(int(suffix.split(".")[-2]), x)
        s = self.stage_index(suffix)
        return self.forward(x, s, s + 1)

    # -- fp8 data path ------------------------------------------------------------------------
    def calibrate_fp8(self, images: torch.Tensor) -> List[float]:
        """One bf16 forward of `images` ([B,3,S,S] in [0,1]) to fix the static fp8 activation scales and
        quantise the weights (include/ivit.h: ivit_fp8_calibrate).  Returns the L*4 scales."""
        batch, _ = self._split_batch(images, 0)
        xin = images.detach().to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()
        stream = torch.cuda.current_stream(xin.device).cuda_stream
        self._check(self.lib.ivit_fp8_calibrate(self._h, batch, ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(stream)))
        return self.fp8_scales()

    def fp8_scales(self) -> List[float]:
        n = 4 * self.cfg.layers
        buf = (ctypes.c_float * n)()
        self._check(self.lib.ivit_fp8_scales(self._h, buf, n))
        return list(buf)

    @property
    def operand_dtype(self) -> torch.dtype:
        """The 16-bit type of the GEMM operand data path (what the oracle's rounding-aware mode has to mirror)."""
        return torch.float16 if self.precision in ("f16", "f16x") else torch.bfloat16

    @property
    def split_gemms(self) -> frozenset:
        """The GEMMs this engine multiplies as hi + lo pairs of f16 values (include/ivit.h: IVIT_PRECISION_F16 / F16X) - what the
        oracle's rounding-aware mode mirrors (oracle/vit_oracle.py: SPLIT_GEMMS)."""
        bits = int(self.lib.ivit_split_set(self._h))   # (the engine decides - per model and from its IVIT_F16X_* knobs; include/ivit.h: ivit_split_set)
        names = set()
        if bits & 1:
            names |= {"patch", "head"}
        if bits & 2:
            names.add("proj")
        if bits & 4:
            names.add("mlp1w")
        if bits & 8:
            names.add("mlp2w")
        return frozenset(names)

    TAPS = {"h1": 1, "qkv": 2, "att": 3, "proj": 4, "h2": 5, "u": 6, "out": 7}

    def layer_tap(self, layer: int, x: torch.Tensor, tap: str) -> torch.Tensor:
        """Runs encoder layer `layer` on x ([B,N,D] f32, CUDA) up to step `tap` and returns that step's output AS
        STORED by the engine (include/ivit.h: ivit_debug_layer_tap): float8_e4m3fn, bfloat16 / float16 or float32
        [B*N, width] - the exact operand bytes the next GEMM consumes."""
        batch, _ = self._split_batch(x, 3)
        xin = x.detach().to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()
        m = batch * self.cfg.tokens
        cap = m * max(3 * self.cfg.dim * 2, self.cfg.mlp * 2, self.cfg.dim * 4)   # (an F16X attention tap is [hi | lo]: 2 * dim * 2 bytes per row)
        raw = torch.empty(cap, dtype=torch.uint8, device=xin.device)
        rb, eb = ctypes.c_int64(0), ctypes.c_int(0)
        stream = torch.cuda.current_stream(xin.device).cuda_stream
        self._check(self.lib.ivit_debug_layer_tap(self._h, layer, batch, ctypes.c_void_p(xin.data_ptr()), self.TAPS[tap],
                                                  ctypes.c_void_p(raw.data_ptr()), cap, ctypes.byref(rb), ctypes.byref(eb), ctypes.c_void_p(stream)))
        flat = raw[:m * rb.value].reshape(m, rb.value)
        if eb.value == 1:
            return flat.view(torch.float8_e4m3fn)
        if eb.value == 2:
            return flat.view(self.operand_dtype if tap != "qkv" or self.precision != "fp8" else torch.bfloat16)
        return flat.view(torch.float32)

    def weight_fp8(self, layer: int, which: int):
        """(e4m3 matrix [rows, cols] as float8 CPU tensor, per-row scales f32 [rows]) of the fp8 data path
        (which: 0 in_proj, 1 out_proj, 2 mlp.0, 3 mlp.3)."""
        d, mlp = self.cfg.dim, self.cfg.mlp
        rows_max = max(3 * d, mlp)
        cap = rows_max * (max(d, mlp) + 128)
        buf = torch.empty(cap, dtype=torch.uint8)
        scale = torch.empty(rows_max, dtype=torch.float32)
        rows, cols, ld = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        self._check(self.lib.ivit_debug_weight_fp8(self._h, layer, which, ctypes.c_void_p(buf.data_ptr()), cap,
                                                   ctypes.cast(scale.data_ptr(), ctypes.POINTER(ctypes.c_float)), rows_max,
                                                   ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(ld)))
        w = buf[:rows.value * ld.value].reshape(rows.value, ld.value)[:, :cols.value].contiguous().view(torch.float8_e4m3fn)
        return w, scale[:rows.value].clone()

    def debug_unfold(self, x: torch.Tensor, normalise: bool) -> torch.Tensor:
        """bf16 unfold image the patch GEMM consumes, as f32 [B*Np, K] (parity-test inspection)."""
        batch, _ = self._split_batch(x, 0)
        xin = x.detach().to(torch.float32).contiguous()
        out = torch.empty((batch * self.cfg.patches, self.cfg.patch_k), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self._check(self.lib.ivit_debug_unfold(self._h, batch, ctypes.c_void_p(xin.data_ptr()),
                                               ctypes.c_void_p(out.data_ptr()), int(normalise), ctypes.c_void_p(stream)))
        return out

    # -- profiling -----------------------------------------------------------------------------
    def profile(self, on: bool) -> None:
        self._check(self.lib.ivit_profile_enable(self._h, int(on)))

    def profile_reset(self) -> None:
        self._check(self.lib.ivit_profile_reset(self._h))

    def profile_read(self) -> Dict[str, Dict[str, float]]:
        res = {}
        for c in range(self.lib.ivit_profile_class_count()):
            ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            self._check(self.lib.ivit_profile_read(self._h, c, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by)))
            res[self.lib.ivit_profile_class_name(c).decode()] = {
                "ms": ms.value, "launches": int(n.value), "flops": fl.value, "bytes": by.value}
        return res

    def profile_kernels(self) -> Dict[str, Dict[str, float]]:
        """Per launch site ("role:kernel name") totals since profile_reset: which kernel every GEMM of the
        forward was dispatched to, with its event time and algorithmic FLOPs / bytes."""
        res = {}
        name = ctypes.create_string_buffer(160)
        for i in range(self.lib.ivit_profile_kernel_count(self._h)):
            ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            self._check(self.lib.ivit_profile_kernel_read(self._h, i, name, len(name), ctypes.byref(ms), ctypes.byref(n),
                                                          ctypes.byref(fl), ctypes.byref(by)))
            res[name.value.decode()] = {"ms": ms.value, "launches": int(n.value), "flops": fl.value, "bytes": by.value}
        return res
