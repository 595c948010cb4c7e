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
    # bf16 does not need the scale for range, but the hash-grid backward accumulates in fixed point sized for the
    # SCALED gradient, so both 16-bit formats use tcnn's fp16 default
    return 128.0 if precision in (torch.float16, torch.bfloat16) else 1.0


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

    out_dtype = torch.float16  # torch.bfloat16 once the "bf16" option is set (Module.__init__)

    def fwd(self, x: torch.Tensor, params: torch.Tensor, save_ctx: bool):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        batch = x.shape[0]
        out = torch.empty((batch, self.padded_output_dims), dtype=self.out_dtype, device=x.device)
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

    def bwd_bwd_input(self, *args, **kwargs):
        """Second-order backward w.r.t. the input (tcnn's torch binding: Module::bwd_bwd_input).  tiny-cuda-nn itself
        implements it for encodings and CutlassMLP only -- FullyFusedMLP, the network type every NeRF-VO configuration
        uses, raises there too -- and nothing on the NeRF-VO path takes a double backward: nerfacto's analytic normals use
        torch.autograd.grad WITHOUT create_graph.  Kept as the raising stub section 8b of
        SURVEY.md asks for, so that a double backward fails loudly instead of silently treating the first-order
        gradient as a constant."""
        raise NotImplementedError(
            "nerf_vo_amd tinycudann: bwd_bwd_input (double backward through a tcnn module) is not implemented -- "
            "FullyFusedMLP has none upstream either; use first-order gradients (torch.autograd.grad without create_graph)")


class _module_function_backward(torch.autograd.Function):
    """The first-order backward as its own autograd node (as in tcnn's bindings/torch/tinycudann/modules.py [UPSTREAM]):
    differentiating THROUGH it -- a double backward -- reaches ``backward`` below, which is the raising stub."""

    @staticmethod
    def forward(ctx, ctx_fwd, doutput, x, params, output):
        ctx.ctx_fwd = ctx_fwd
        native = ctx_fwd.native
        scaled = (doutput.to(torch.float32) * ctx_fwd.loss_scale).to(native.out_dtype).contiguous()
        need_dx = ctx_fwd.needs_input_grad[1]
        need_dp = ctx_fwd.needs_input_grad[2] and native.n_params > 0
        with torch.no_grad():
            dx, dp = native.bwd(ctx_fwd.native_ctx, x, params, output, scaled, need_dx, need_dp)
            if dx is not None:
                dx = dx / ctx_fwd.loss_scale
            if dp is not None:
                dp = dp / ctx_fwd.loss_scale  # fp32, accumulated in fp32 by the kernels (tcnn hands back fp16 here)
        return dx, dp

    @staticmethod
    def backward(ctx, *_grads):
        return ctx.ctx_fwd.native.bwd_bwd_input()


class _module_function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, native: _NativeModule, x: torch.Tensor, master: torch.Tensor, params: torch.Tensor,
                loss_scale: float):
        """master: the fp32 parameters (what receives the gradient); params: their 16-bit working copy in the
        format the kernels read (not differentiated)."""
        ctx.set_materialize_grads(False)
        needs_grad = x.requires_grad or master.requires_grad
        native_ctx, output = native.fwd(x, params, save_ctx=needs_grad)
        ctx.save_for_backward(x, params, output)
        ctx.native = native
        ctx.native_ctx = native_ctx
        ctx.loss_scale = loss_scale
        return output

    @staticmethod
    def backward(ctx, doutput):
        if doutput is None:
            return None, None, None, None, None
        x, params, output = ctx.saved_tensors
        if ctx.native_ctx is None:
            raise RuntimeError("tcnn module backward called but forward ran without gradient tracking")
        dx, dp = _module_function_backward.apply(ctx, doutput, x, params, output)
        return None, dx, dp, None, None


class Module(torch.nn.Module):
    """``dtype``: torch.float16 (default, tcnn's precision) or torch.bfloat16 -- an extension over tcnn: outputs,
    hidden activations and MLP weights in bfloat16 on v_mfma_f32_16x16x16_bf16 (hash tables stay fp16 with fp32
    interpolation / gradient accumulation)."""

    def __init__(self, seed: int = 1337, dtype=None):
        super().__init__()
        dtype = torch.float16 if dtype is None else dtype
        if dtype not in (torch.float16, torch.bfloat16):
            raise NotImplementedError("nerf_vo_amd tcnn modules compute in fp16 or bf16 (fp32 accumulate) only")
        self.dtype = dtype
        self.native_tcnn_module = self._native_tcnn_module()
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
            self.params,
            self._working_copy(),
            self.loss_scale,
        )
        return output[:batch_size, : self.n_output_dims]

    def _working_copy(self) -> torch.Tensor:
        """16-bit copy of the parameters in the format the kernels read (raw bits for mixed layouts)."""
        return self.params.detach().to(self.dtype).contiguous()

    def _configure_native(self, m: _NativeModule) -> _NativeModule:
        if self.dtype == torch.bfloat16:
            m.set_option("bf16", 1)
            m.out_dtype = torch.bfloat16
        return m

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("native_tcnn_module", None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault("dtype", torch.float16)
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

    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337, dtype=None):
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.encoding_config = dict(encoding_config)
        self.network_config = dict(network_config)
        super().__init__(seed=seed, dtype=dtype)

    def _native_tcnn_module(self):
        m = _create(
            "nvo_create_network_with_input_encoding", self.n_input_dims, self.n_output_dims,
            json.dumps(self.encoding_config).encode(), json.dumps(self.network_config).encode())
        m.needs_ctx_for_inference = True
        enc = _create("nvo_create_encoding", self.n_input_dims, json.dumps(self.encoding_config).encode())
        self._n_network_params = m.n_params - enc.n_params  # params = [network | encoding]
        del enc
        return self._configure_native(m)

    def _working_copy(self) -> torch.Tensor:
        if self.dtype == torch.float16:
            return super()._working_copy()
        # bf16: network weights bfloat16, hash table fp16 -- one raw 16-bit buffer
        p = self.params.detach()
        k = self._n_network_params
        return torch.cat([p[:k].to(torch.bfloat16).view(torch.int16), p[k:].to(torch.float16).view(torch.int16)])


class Network(Module):
    """Fully fused MLP on raw (identity-encoded, 1-padded) inputs."""

    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337, dtype=None):
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.network_config = dict(network_config)
        super().__init__(seed=seed, dtype=dtype)

    def _native_tcnn_module(self):
        return self._configure_native(_create("nvo_create_network", self.n_input_dims, self.n_output_dims,
                                              json.dumps(self.network_config).encode()))


class Encoding(Module):
    """HashGrid or SphericalHarmonics encoding; output width is exact (no padding), dtype fp16."""

    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        super().__init__(seed=seed, dtype=dtype)
        self.n_output_dims = self.native_tcnn_module.n_output_dims

    def _native_tcnn_module(self):
        return self._configure_native(_create("nvo_create_encoding", self.n_input_dims,
                                              json.dumps(self.encoding_config).encode()))

    def _working_copy(self) -> torch.Tensor:
        return self.params.detach().to(torch.float16).contiguous()  # the table is fp16 in both output formats
