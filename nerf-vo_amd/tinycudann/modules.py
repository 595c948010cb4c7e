"""tcnn-compatible torch modules (upstream tiny-cuda-nn bindings/torch/tinycudann/modules.py is the
interface being mirrored; it is NOT vendored in /root/reference -- SURVEY.md section 8b/8c)."""
from __future__ import annotations

import ctypes as C
import json

import torch

from .. import _lib

_BATCH_GRANULARITY = 128


def batch_size_granularity() -> int:
    return _BATCH_GRANULARITY


def preferred_precision() -> torch.dtype:
    return torch.float16


def default_loss_scale(precision: torch.dtype = torch.float16) -> float:
    return 128.0 if precision == torch.float16 else 1.0


def free_temporary_memory() -> None:
    """The native side never caches scratch (callers own it), so there is nothing to free."""
    return None


def supports_jit_fusion() -> bool:
    return False


def _ptr(t: torch.Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device: torch.device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NativeModule:
    """Owns one nvo_module_t handle."""

    def __init__(self, handle: C.c_void_p):
        self.handle = handle
        lib = _lib.lib()
        self.n_input_dims = int(lib.nvo_n_input_dims(handle))
        self.n_output_dims = int(lib.nvo_n_output_dims(handle))
        self.padded_output_dims = int(lib.nvo_padded_output_dims(handle))
        self.n_params = int(lib.nvo_n_params(handle))

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().nvo_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def initial_params(self, seed: int) -> torch.Tensor:
        out = torch.empty(self.n_params, dtype=torch.float32)
        if self.n_params:
            _lib.check(_lib.lib().nvo_initial_params(self.handle, seed, C.c_void_p(out.data_ptr())), "initial_params")
        return out

    def ctx_bytes(self, batch: int) -> int:
        return int(_lib.lib().nvo_ctx_bytes(self.handle, batch))

    def set_option(self, key: str, value: int) -> None:
        _lib.check(_lib.lib().nvo_set_option(self.handle, key.encode(), int(value)), f"set_option({key})")

    def fwd(self, x: torch.Tensor, params: torch.Tensor, save_ctx: bool):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        batch = x.shape[0]
        out = torch.empty((batch, self.padded_output_dims), dtype=torch.float16, device=x.device)
        ctx = None
        if save_ctx or self.needs_ctx_for_inference:
            ctx = torch.empty(self.ctx_bytes(batch), dtype=torch.uint8, device=x.device)
        _lib.check(
            _lib.lib().nvo_fwd(self.handle, _stream(x.device), batch, _ptr(x), _ptr(params), _ptr(out), _ptr(ctx)),
            "fwd",
        )
        return ctx, out

    needs_ctx_for_inference = False

    def bwd(self, ctx, x, params, out, dout, need_dinput: bool, need_dparams: bool):
        batch = x.shape[0]
        dx = torch.empty_like(x) if need_dinput else None
        dp = torch.empty(self.n_params, dtype=torch.float32, device=x.device) if need_dparams else None
        _lib.check(
            _lib.lib().nvo_bwd(
                self.handle, _stream(x.device), batch, _ptr(x), _ptr(params), _ptr(out), _ptr(dout), _ptr(ctx),
                _ptr(dx), _ptr(dp)),
            "bwd",
        )
        return dx, dp


class _module_function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, native: _NativeModule, x: torch.Tensor, params: torch.Tensor, loss_scale: float):
        ctx.set_materialize_grads(False)
        needs_grad = x.requires_grad or params.requires_grad
        native_ctx, output = native.fwd(x, params, save_ctx=needs_grad)
        ctx.save_for_backward(x, params, output)
        ctx.native = native
        ctx.native_ctx = native_ctx
        ctx.loss_scale = loss_scale
        return output

    @staticmethod
    def backward(ctx, doutput):
        if doutput is None:
            return None, None, None, None
        x, params, output = ctx.saved_tensors
        if ctx.native_ctx is None:
            raise RuntimeError("tcnn module backward called but forward ran without gradient tracking")
        scaled = (doutput.to(torch.float32) * ctx.loss_scale).to(torch.float16).contiguous()
        need_dx = ctx.needs_input_grad[1]
        need_dp = ctx.needs_input_grad[2] and ctx.native.n_params > 0
        dx, dp = ctx.native.bwd(ctx.native_ctx, x, params, output, scaled, need_dx, need_dp)
        if dx is not None:
            dx = dx / ctx.loss_scale
        if dp is not None:
            dp = (dp / ctx.loss_scale).to(params.dtype)
        return None, dx, dp, None


class Module(torch.nn.Module):
    def __init__(self, seed: int = 1337):
        super().__init__()
        self.native_tcnn_module = self._native_tcnn_module()
        self.dtype = torch.float16
        self.seed = seed
        initial_params = self.native_tcnn_module.initial_params(seed)
        self.params = torch.nn.Parameter(initial_params, requires_grad=True)
        self.loss_scale = default_loss_scale(self.dtype)

    def _native_tcnn_module(self) -> _NativeModule:  # pragma: no cover - abstract
        raise NotImplementedError

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError(
                "tinycudann (nerf_vo_amd) modules run only on an MI355X device tensor; there is no CPU fallback")
        batch_size = x.shape[0]
        g = _BATCH_GRANULARITY
        padded = (batch_size + g - 1) // g * g
        x_padded = x if batch_size == padded else torch.nn.functional.pad(x, [0, 0, 0, padded - batch_size])
        output = _module_function.apply(
            self.native_tcnn_module,
            x_padded.to(torch.float).contiguous(),
            self.params.to(self.dtype).contiguous(),
            self.loss_scale,
        )
        return output[:batch_size, : self.n_output_dims]

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("native_tcnn_module", None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.native_tcnn_module = self._native_tcnn_module()

    def extra_repr(self) -> str:
        return f"n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, seed={self.seed}, dtype={self.dtype}"


def _create(fn_name: str, *args) -> _NativeModule:
    handle = C.c_void_p()
    fn = getattr(_lib.lib(), fn_name)
    _lib.check(fn(*args, C.byref(handle)), fn_name)
    return _NativeModule(handle)


class NetworkWithInputEncoding(Module):
    """Input encoding followed by a fully fused MLP; params = [network | encoding] (tcnn order)."""

    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337):
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.encoding_config = dict(encoding_config)
        self.network_config = dict(network_config)
        super().__init__(seed=seed)

    def _native_tcnn_module(self):
        m = _create(
            "nvo_create_network_with_input_encoding", self.n_input_dims, self.n_output_dims,
            json.dumps(self.encoding_config).encode(), json.dumps(self.network_config).encode())
        m.needs_ctx_for_inference = True
        return m


class Network(Module):
    """Fully fused MLP on raw (identity-encoded, 1-padded) inputs."""

    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.network_config = dict(network_config)
        super().__init__(seed=seed)

    def _native_tcnn_module(self):
        return _create("nvo_create_network", self.n_input_dims, self.n_output_dims,
                       json.dumps(self.network_config).encode())


class Encoding(Module):
    """HashGrid or SphericalHarmonics encoding; output width is exact (no padding), dtype fp16."""

    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        if dtype is not None and dtype != torch.float16:
            raise NotImplementedError("nerf_vo_amd tcnn.Encoding computes in fp16 only")
        super().__init__(seed=seed)
        self.n_output_dims = self.native_tcnn_module.n_output_dims

    def _native_tcnn_module(self):
        return _create("nvo_create_encoding", self.n_input_dims, json.dumps(self.encoding_config).encode())
