"""Native training / rendering engine of the depth-nerfacto model NeRF-VO maps with.

This is the host-side sequencing of the HIP kernels for ONE optimisation step (and for inference
rendering): PyTorch only provides device memory, the stream and the RNG; every arithmetic step of
the hot path is a kernel behind the C-ABI (include/nerfvo_hip.h).  The step mirrors what
``trainer.train_iteration`` does in the reference (/root/reference/nerf_vo/mapping/nerfstudio.py:151
with the configuration of :47-103; call stack in SURVEY.md section 3.1/3.3):

    raygen(+pose correction) -> lin-disp bins(256) -> proposal net 0 -> weights+PDF(96)
    -> proposal net 1 -> weights+PDF(48) -> hash grid + base MLP -> SH + colour MLP
    -> render + losses (+ per-sample gradients) -> colour/base MLP + grid backward
    -> [proposal losses + backward when the sampler schedule says so] -> fused Adam

All trainable parameters live in ONE flat fp32 buffer (+ fp16 working copy, gradient, Adam moments)
so that a multi-GPU step needs exactly one RCCL all-reduce and one optimiser pass per group.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import gc
import json
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib
from .tinycudann.modules import _NativeModule, _create, _ptr, _stream


@contextlib.contextmanager
def capture_graph(graph: "torch.cuda.CUDAGraph", **kw):
    """torch.cuda.graph(graph) with the cyclic garbage collector held off while the stream records: a finaliser that frees
    device memory or destroys another graph in the middle of a capture invalidates it (hipErrorStreamCaptureInvalidated --
    seen when engines of earlier runs were still waiting for the collector)."""
    was_on = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was_on:
            gc.enable()


@dataclass
class GridConfig:
    n_levels: int
    log2_hashmap_size: int
    base_resolution: int
    max_resolution: int
    n_features_per_level: int = 2

    @property
    def per_level_scale(self) -> float:
        # nerfstudio HashEncoding: growth = exp((ln max_res - ln base_res) / (L - 1))
        return float(np.exp((np.log(self.max_resolution) - np.log(self.base_resolution)) / (self.n_levels - 1)))

    def tcnn_dict(self) -> dict:
        return {"otype": "HashGrid", "n_levels": self.n_levels, "n_features_per_level": self.n_features_per_level,
                "log2_hashmap_size": self.log2_hashmap_size, "base_resolution": self.base_resolution,
                "per_level_scale": self.per_level_scale}


@dataclass
class EngineConfig:
    """Values = nerfacto defaults [UPSTREAM, SURVEY.md section 3.3] overridden as the reference does
    (/root/reference/nerf_vo/mapping/nerfstudio.py:62-82)."""
    num_images: int = 192
    num_rays: int = 4096
    near_plane: float = 0.05
    far_plane: float = 1000.0
    num_proposal_samples: tuple = (256, 96)
    num_nerf_samples: int = 48
    main_grid: GridConfig = field(default_factory=lambda: GridConfig(16, 19, 16, 2048))
    proposal_grids: tuple = (GridConfig(5, 17, 16, 128), GridConfig(5, 17, 16, 256))
    hidden_dim: int = 64
    geo_feat_dim: int = 15
    appearance_embed_dim: int = 32
    density_bias: float = -1.0            # sigma = trunc_exp(h0 - 1)
    # The backward of a TRAINED field: one sample of a ray's 48 carries the weight, two of its three 16-sample tiles carry no
    # gradient at all.  "sparse" steps have the render / loss kernel mark the tiles (nvo_main_loss_args::tile_live), the two
    # MLP backwards of the main field walk only the live ones and the hash grid's backward only the listed live rows
    # (-8 % per step on the trained field of the mapping run).  The list-capable kernels are separate instantiations -- with
    # the list code compiled in, the plain forms lost 10-15 % to scalar-register spills -- so WHICH step graph runs is
    # decided on the host: "auto" runs a sparse step every 64th step, reads the live-tile count it leaves behind
    # (asynchronously) and switches all steps to the sparse graphs while fewer than 60 % of the tiles are live (back above
    # 75 %); "on" / "off" pin it.  Either kind of step is correct on any field.
    sparse_backward: str = "auto"
    histogram_padding: float = 0.01
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    proposal_warmup: int = 5000
    proposal_update_every: int = 5
    rgb_loss_mult: float = 1.0
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    depth_loss_mult: float = 0.001
    depth_sigma: float = 0.001
    normal_loss_mult: float = 5e-6        # reference: nerf_vo/mapping/nerfstudio.py:77 (monosdf normal supervision)
    loss_scale: float = 128.0
    # torch.cuda.amp.GradScaler dynamics (the reference trains with mixed_precision=True,
    # /root/reference/nerf_vo/mapping/nerfstudio.py:59 -- nerfstudio wraps the step in GradScaler(): init 65536, x2 after
    # 2000 consecutive clean steps, x0.5 when any optimiser saw a non-finite gradient).  The state (scale, growth
    # tracker, per-group applied-step counters) lives on the device, so the captured step stays valid.  True is the
    # DEFAULT since round 4: it is the regime the reference trains in (at the static scale most proposal-loss gradients
    # underflow fp16 and the grid backward skips them -- a shortcut the reference never gets).  False = tcnn's static
    # loss scale (`loss_scale`) + skip-on-non-finite.
    dynamic_loss_scale: bool = True
    loss_scale_init: float = 65536.0
    loss_scale_growth: float = 2.0
    loss_scale_backoff: float = 0.5
    loss_scale_interval: int = 2000
    loss_scale_max: float = 16777216.0    # 2^24: keeps the fixed-point grid accumulators (|v| < 2^25) in range in bf16 mode
    # torch's GradScaler has NO lower bound: a run whose gradients overflow at every scale >= 1 (density pre-activations
    # beyond 15 multiply dL/dsigma by e^15 = 3.3e6 through trunc_exp's clamped backward) keeps halving until they fit
    # and goes on training.  Rounds 1-3 clamped at 1.0, which turns such an episode into a permanent skip (found with the
    # fixed-pose deterministic seed-42 run of round 4: one of 29 otherwise healthy 8192-step runs).
    loss_scale_min: float = 2.0 ** -24
    lr_fields: float = 1e-2
    lr_proposal: float = 1e-2
    lr_camera: float = 1e-4
    lr_camera_final: float = 1e-5
    adam_eps: float = 1e-15
    adam_betas: tuple = (0.9, 0.999)
    max_num_iterations: int = 8192
    optimize_poses: bool = False          # BASELINE configs[1] is "fixed poses"; True = the reference's SE3 optimiser
    camera_mode: str = "SE3"              # "SE3" (reference's explicit config) | "SO3xR3" (nerfacto model default)
    camera_trans_l2_penalty: float = 1e-2  # camera_opt_regularizer (nerfstudio >= 1.0 CameraOptimizer [UPSTREAM])
    camera_rot_l2_penalty: float = 1e-3
    # hash-grid parameter-gradient kernel per network (main field, proposal 0, proposal 1):
    # 3 = streamed binned (self-contained records; coarse levels slice-owner), 1 = LDS slice owner,
    # 2 = binned with gathers, 0 = global atomics.  An int applies to all three.
    grid_bwd_mode: int | tuple = (3, 1, 1)
    # run the proposal-network losses + backward on a second HIP stream beside the main-field backward
    # (they only share read-only inputs; forked after the render/loss kernel, joined before the optimiser)
    overlap_proposal_backward: bool = True
    # chunks of the proposal grids' DENSE slices relative to the even split of the one-round item table, percent;
    # None = 120 when most samples carry a gradient (bf16 MLPs or dynamic_loss_scale), else 100
    proposal_dense_share: int | None = None
    # Main grid: the forward also stores d(encoded)/d(position) (tcnn's prepare_input_gradients) whenever positions
    # need gradients (pose optimisation, analytic normals); the input backward then streams it (112 -> ~20 us) instead
    # of gathering the corners again.  None = on iff optimize_poses or expect_normals.
    store_input_gradients: bool | None = None
    # the data manager delivers normal images (enhancement modes containing 'normal'): the analytic-normal pass needs
    # d(density)/d(position) every step, so the main grid's forward stores its input gradients
    expect_normals: bool = False
    # pose optimisation: the main grid's parameter scatter runs on a second stream beside the pose-gradient chain
    overlap_pose_backward: bool = True
    # one-graph step (single GPU): the optimiser's commit (step counters, bias corrections, loss scale) is not a node of
    # the graph but rides in the eager launch behind the replay that also writes the NEXT step's scalars
    commit_behind_replay: bool = True
    # ... or (takes precedence) IS the last node of the graph and takes the next step's scalars from a device table the
    # host fills 1024 steps ahead (nvo_opt_commit_table): nothing eager behind the replay -- the trace showed 8 us of
    # launch latency between the graph's last kernel and the eager launch, plus its 5 us, on every step
    commit_from_table: bool = True
    # one-graph step (single GPU): the tile-local accumulate pass of the main grid applies Adam to the entries of the
    # hashed levels it has just summed (nvo_set_fused_adam) instead of storing their gradient -- 8 B of HBM traffic per
    # parameter less (11.5 M of the 12.2 M), and the optimiser launch covers the rest of the fields group only.
    # Bit-identical to the separate launch.  Needs the producer flags (the group's verdict must be final before the pass).
    fuse_grid_adam: bool = True
    # copies of every fused MLP's weight-gradient buffer: the backward's workgroups spread their block totals over
    # 1 + dw_replicas buffers (256-512 workgroups adding to the same few cache lines serialise at the L2's atomic units:
    # 7-10 us per launch), ONE launch per step folds the copies (nvo_fold_replicas).  0 = off.
    dw_replicas: int = 7
    # multi-GPU: launch the next iteration's sampling prefix (rays -> proposal sampling; reads the proposal networks and
    # poses only) inside this iteration's graph, while the fields gradient is still being exchanged (train_step_graphed;
    # bit-identical to the un-pipelined order).  After a step the workspace and the drawn-pixel buffers then already
    # hold the NEXT step's rays.
    pipeline_sampling_prefix: bool = True
    # the same software pipelining on ONE GPU: the graph ends with [Adam of the fields group || next sampling prefix].
    # Bit-identical, and MEASURED NEUTRAL (0.617 vs 0.608 ms/step; even the prefix beside the whole main backward
    # gains only 2 %): the Adam stream saturates HBM and the prefix's gathers then wait longer for their misses -- the
    # chip has no idle resource for a second kernel to use.  Off by default; kept for the test of the launch order.
    pipeline_single_gpu: bool = False
    # 16-bit format of everything the fused MLPs stream (weights, encoded features, hidden activations, outputs and
    # their gradients): "f16" = tcnn's precision (BASELINE configs[1-3]); "bf16" = v_mfma_f32_16x16x16_bf16 with
    # the hash tables kept fp16 + fp32 interpolation / fp32 gradient accumulation (BASELINE configs[4]:
    # "MFMA bf16 MLP + fp32 hash accumulate"; reference: mixed_precision=True, nerf_vo/mapping/nerfstudio.py:59)
    mlp_dtype: str = "f16"
    # nerfstudio evaluates the interlevel and proposal-level depth losses on EVERY step and returns them in loss_dict
    # (/root/reference/nerf_vo/mapping/nerfstudio.py:151-152), also on the steps where the proposal networks do not
    # train; here they are produced by the proposal backward, which only runs on update steps.  "logging": on the
    # steps the reference looks at its loss_dict (step % log_every == 0, nerfstudio.py:161-168) a value-only pass of
    # the proposal loss kernel fills them in (~40 us on those steps); "always": on every non-update step (exact a1
    # return values on every step); "never": the terms read 0 on non-update steps.  Gradients are identical in all
    # three (nerfstudio computes these values under no_grad on such steps).
    proposal_loss_values: str = "logging"
    # Bitwise reproducible training (debugging aid; several times slower): every float-atomic reduction of the step is
    # replaced by a fixed summation order -- MLP weight gradients (per-workgroup block totals + an ordered reduce),
    # the colour head's per-camera embedding / per-ray SH gradients, the per-camera pose gradient, and the hash grids
    # (ONE owner work item per slice / bin instead of sample chunks that meet in float atomics, integer accumulators
    # everywhere, no live-sample list).  Two runs from the same state then produce bit-identical parameters; only the
    # REPORTED loss values (64 float-atomic shards) may differ in their last bits.
    deterministic: bool = False
    # GradScaler's found_inf raised AT THE SOURCE (single GPU).  Every gradient of the step descends from the 16-bit
    # dL/d(pre-activation) / dL/d(rgb) the loss kernels store, through 16-bit buffers (d_base_out, dL/d(encoded)) into
    # fp32 accumulations; an fp32 sum of finite 16-bit products cannot overflow, so a non-finite gradient exists iff a
    # 16-bit value on that chain is non-finite -- and a non-finite value propagates down the chain to its end, the
    # hash-grid backward.  The loss kernels (the roots) and the grid backward (the leaves, which already had to detect
    # it for their integer accumulators) OR the flag word of their parameter group; the optimiser no longer re-reads the
    # 55 MB gradient buffer (nonfinite_flag: 13-17 us per step).  (Instrumenting the fused-MLP backward as well was
    # measured: +4.5 / +4.6 / +1.7 us on its three launches -- as much as the scan saved.)  [round 4] The argument has a
    # hole that GradScaler's growing scale walks into: a 16-bit value INSIDE the chain (a hidden dZ, d_base_out) can
    # overflow while the roots are finite, and a ReLU backward can drop it before a leaf sees it.  The layer it appears in
    # still multiplies it into its weight gradient, so the optimiser scans the ~30 K NON-GRID scalars of each group (MLP
    # weights, embedding, poses; one small launch) -- optimizer_step.  With a process group the check stays behind the
    # reduction (another rank may have overflowed).
    producer_overflow_flags: bool = True
    log_every: int = 10                   # LoggingConfig.steps_per_log of the trainer mirror
    seed: int = 1337


_MLP16 = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 16,
          "n_hidden_layers": 1}


def _call(name: str, *args) -> None:
    _lib.check(getattr(_lib.lib(), name)(*args), name)


class NerfactoEngine:
    """Owns parameters, optimiser state, scratch and the kernel sequence of one step."""

    LOSS_NAMES = ("rgb_loss", "distortion_loss", "depth_loss", "interlevel_loss", "prop_depth_loss")

    def __init__(self, config: EngineConfig, device: torch.device, world_size: int = 1, rank: int = 0):
        if device.type != "cuda":
            raise RuntimeError("NerfactoEngine needs an MI355X device; there is no CPU fallback")
        self.cfg = config
        self.device = device
        self.world_size = world_size
        self.rank = int(rank)  # mixed into the stateless pixel / jitter sampler: every rank draws its own rays
        cfg = config
        self.levels = (*cfg.num_proposal_samples, cfg.num_nerf_samples)
        if cfg.mlp_dtype not in ("f16", "bf16"):
            raise ValueError(f"mlp_dtype must be 'f16' or 'bf16' (got {cfg.mlp_dtype!r})")
        self.bf16 = cfg.mlp_dtype == "bf16"
        self.act_dtype = torch.bfloat16 if self.bf16 else torch.float16
        import os
        if os.environ.get("NVO_SPARSE_BACKWARD"):  # (A/B switch for measurements: auto | on | off)
            cfg.sparse_backward = os.environ["NVO_SPARSE_BACKWARD"]
        assert cfg.sparse_backward in ("auto", "on", "off"), cfg.sparse_backward
        if cfg.sparse_backward == "auto" and world_size > 1:
            # (the probe's answer arrives when the copy has landed -- a host-side race the ranks would each decide for
            # themselves; data-parallel ranks run the same kind of step: pin it explicitly to get the sparse one)
            cfg.sparse_backward = "off"

        # ---- native modules (tcnn NetworkWithInputEncoding: params = [mlp | grid])
        self.prop_nets = [
            _create("nvo_create_network_with_input_encoding", 3, 1, json.dumps(g.tcnn_dict()).encode(),
                    json.dumps(_MLP16).encode())
            for g in cfg.proposal_grids
        ]
        base_cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                    "n_neurons": cfg.hidden_dim, "n_hidden_layers": 1}
        self.base_net = _create("nvo_create_network_with_input_encoding", 3, 1 + cfg.geo_feat_dim,
                                json.dumps(cfg.main_grid.tcnn_dict()).encode(), json.dumps(base_cfg).encode())
        for m in self.prop_nets:  # one density per sample: [N] halfs instead of [N][16] rows
            m.set_option("compact_output", 1)
        for m in (*self.prop_nets, self.base_net):  # single hidden layer: recomputed in the backward, never stored
            m.set_option("recompute_hidden", 1)
        modes = cfg.grid_bwd_mode if isinstance(cfg.grid_bwd_mode, (tuple, list)) else (cfg.grid_bwd_mode,) * 3
        for m, mode in zip((self.base_net, *self.prop_nets), modes):
            m.set_option("grid_bwd_mode", int(mode))
            m.set_option("bf16", int(self.bf16))
            m.set_option("deterministic", int(bool(cfg.deterministic)))
        # Settled by measurement in rounds 2-4 (EXPERIMENTS.md), no longer switches: the main grid's four coarse levels stay
        # slice-owner with int32 accumulators and the L1-derived scale, as the proposal grids (half the slices per level,
        # a cheaper conversion: 1 M-sample grid 298 -> 227 us); dense levels use the run-merging scan; the record pass of
        # the streamed levels is the packed 2 x 32-bit form (the only one left in grid.hip).
        self.base_net.set_option("grid_acc_bits", 32)
        batches = (cfg.num_nerf_samples, *cfg.num_proposal_samples)
        for m, per_ray in zip((self.base_net, *self.prop_nets), batches):
            m.set_option("grid_bwd_runs", 1)
            m.set_option("grid_bwd_batch", int(cfg.num_rays * per_ray))
        # proposal grids (slice-owner form): int32 accumulators with the overflow-proof L1-derived scale -- half
        # the slices per level and a cheaper conversion (1 M-sample grid 298 -> 227 us, 393 K-sample grid 157 -> 126 us)
        # share of the one-round item table that goes to the dense levels: with most proposal samples carrying a gradient
        # (bf16 keeps what fp16 flushes; so does the reference's loss scale of 65536) their items are the slower kind
        dense_share = cfg.proposal_dense_share
        if dense_share is None:
            # (measured, scannet-shaped bf16 step: 100 / 120 / 140 / 170 / 200 -> 130.7 / 121.8 / 128.3 / 150.5 / 198 us per
            # launch; fp16 with the static scale, where few samples are live: 70 / 85 / 100 -> 77.8 / 71.0 / 61 us)
            dense_share = 120 if (self.bf16 or cfg.dynamic_loss_scale) else 100
            import os
            dense_share = int(os.environ.get("NVO_PROP_DENSE_SHARE", dense_share))  # measurements
        for m in self.prop_nets:
            m.set_option("grid_bwd_dense_share", int(dense_share))
            m.set_option("grid_acc_bits", 32)
            # (module option fuse_encoding -- the hash grid inside the MLP kernel's operand load -- stays off: measured
            # 51.5 us per launch against 35 + 8.2 us for the two kernels)
            # under tcnn's static loss scale most proposal samples carry an exactly zero gradient after a few hundred steps:
            # scan the live ones only (the launch leaves at once while >= 3/4 of the samples are live)
            m.set_option("grid_compact_live", 1)
        store = cfg.store_input_gradients
        if store is None:
            store = bool(cfg.optimize_poses or cfg.expect_normals)
        self.base_net.set_option("prepare_input_gradients", int(bool(store)))
        self._zero_plan = None  # built lazily (needs the flat gradient buffer): _step_zero_ranges()
        self._producer_flags = False  # set per call: the producers raise the groups' overflow flags (single GPU)
        color_in = 16 + cfg.geo_feat_dim + cfg.appearance_embed_dim
        assert color_in == 63 and cfg.hidden_dim == 64, "colour head kernel is specialised to 63 -> 64 -> 64 -> 3"
        self.n_color = 64 * 64 + 64 * 64 + 16 * 64

        # ---- flat parameter layout: name -> (offset, size, group)
        segs = []
        off = 0

        def add(name, size, group):
            nonlocal off
            segs.append((name, off, size, group))
            off += size

        add("field.base", self.base_net.n_params, "fields")
        add("field.color", self.n_color, "fields")
        add("field.embedding", cfg.num_images * cfg.appearance_embed_dim, "fields")
        # the fields group splits into 1 / 2 / 4 / 8 equal, 16-byte aligned shards (sharded optimiser of the multi-GPU
        # step); the padding parameters never receive a gradient and stay zero
        add("field.pad", (-off) % 512, "fields")
        for i, m in enumerate(self.prop_nets):
            add(f"proposal.{i}", m.n_params, "proposal_networks")
        add("camera_opt.pose_adjustment", cfg.num_images * 6, "camera_opt")
        self.segments = {n: (o, s, g) for n, o, s, g in segs}
        self.n_params = off
        self.group_ranges = {}
        for n, o, s, g in segs:
            lo, hi = self.group_ranges.get(g, (o, o))
            self.group_ranges[g] = (min(lo, o), max(hi, o + s))

        # bf16 mode: which element ranges of the 16-bit working copy are bfloat16 (fused-MLP weights + appearance
        # embedding); everything else -- the hash tables (and the unused copy of the poses) -- stays fp16
        self.bf16_ranges = []
        if self.bf16:
            n_base_mlp = self.base_net.n_params - self._grid_params(self.base_net)
            o = self.segments["field.base"][0]
            self.bf16_ranges.append((o, o + n_base_mlp))
            self.bf16_ranges.append((self.segments["field.color"][0], sum(self.segments["field.embedding"][:2])))
            for i, m in enumerate(self.prop_nets):
                o = self.segments[f"proposal.{i}"][0]
                self.bf16_ranges.append((o, o + m.n_params - self._grid_params(m)))
        self._bf16_lo = (C.c_uint64 * max(len(self.bf16_ranges), 1))(*[lo for lo, _ in self.bf16_ranges])
        self._bf16_hi = (C.c_uint64 * max(len(self.bf16_ranges), 1))(*[hi for _, hi in self.bf16_ranges])
        dev = device
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        # 16-bit working copy the kernels read: fp16, except the bf16_ranges in bf16 mode (raw bits; decode with
        # working_copy_float())
        self.params_half = torch.zeros(self.n_params, dtype=torch.float16, device=dev)
        self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.losses = torch.zeros(64, 8, dtype=torch.float32, device=dev)  # sharded accumulators
        self.skip_flag = torch.zeros(4, dtype=torch.int32, device=dev)  # one word per parameter group of the step
        # [anneal | (lr, bias1, bias2_sqrt) x 3 | ... | anneal, sampler step counter]: the sampling scalars are the last
        # two slots of the same buffer, so a single-GPU step refreshes everything with ONE tiny launch
        self.dev_scalars = torch.zeros(16, dtype=torch.float32, device=dev)
        self.dev_sampling = self.dev_scalars[14:16]
        self._pending_head = None  # stamp of the sampling prefix already launched for the next step (pipelined graphs)
        self._graphs = {}
        self._side_stream = None
        self._scatter_stream = None
        self._defer_commit = None   # list while the one-graph step is being captured (optimizer_step appends its masks)
        self._scalars_step = None   # step whose scalars _commit_and_write has already put into dev_scalars
        self._pix_scale = None
        self.corrections = torch.zeros(cfg.num_images, 3, 4, dtype=torch.float32, device=dev)
        self.d_corrections = torch.zeros(cfg.num_images, 3, 4, dtype=torch.float32, device=dev)
        self._pose_inputs = None  # (intrinsics, c2w) of the rays currently loaded (pose backward)
        # GradScaler-shaped optimiser state ON THE DEVICE: [applied steps x 4 groups | loss scale (float bits) | growth
        # tracker | pad]; slot of a group = its index in _GROUP_ORDER (also its skip-flag word)
        self.opt_state = torch.zeros(16, dtype=torch.int32, device=dev)
        self.dev_applied = self.opt_state[0:4]
        self.dev_loss_scale = self.opt_state[4:5].view(torch.float32)
        self.dev_growth_tracker = self.opt_state[5:6]
        self.dev_bias = self.opt_state[8:16].view(torch.float32)  # [group][2]: bias corrections of the next applied step
        self.dev_loss_scale.fill_(cfg.loss_scale_init if cfg.dynamic_loss_scale else cfg.loss_scale)
        self._write_bias([0, 0, 0, 0])
        self.step = 0
        self.steps_since_proposal_update = 0
        self._ws = {}  # (ray count, training) -> scratch; never evicted (captured graphs address it by pointer)
        self.init_params(cfg.seed)
        self._dw_rep = None
        self._dw_replica_plan()  # (allocated here, never under a graph capture)

    # ------------------------------------------------------------------------------------------
    # parameters
    # ------------------------------------------------------------------------------------------
    def view(self, name: str, buf: torch.Tensor | None = None) -> torch.Tensor:
        o, s, _ = self.segments[name]
        return (self.params if buf is None else buf)[o:o + s]

    def init_params(self, seed: int) -> None:
        """tcnn-style init for grids / MLPs (native PCG32 stream), N(0,1) appearance embedding
        (torch.nn.Embedding default), zero pose adjustment."""
        host = torch.zeros(self.n_params, dtype=torch.float32)
        host[self._slice("field.base")] = self.base_net.initial_params(seed)
        # colour head: Xavier-uniform like a tcnn.Network(63 -> 3, 64 neurons, 2 hidden layers)
        color = _create("nvo_create_network", 63, 3, json.dumps(
            {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid", "n_neurons": 64,
             "n_hidden_layers": 2}).encode())
        assert color.n_params == self.n_color
        host[self._slice("field.color")] = color.initial_params(seed)
        g = torch.Generator().manual_seed(seed)
        host[self._slice("field.embedding")] = torch.randn(self.segments["field.embedding"][1], generator=g)
        for i, m in enumerate(self.prop_nets):
            host[self._slice(f"proposal.{i}")] = m.initial_params(seed + 1 + i)
        self.set_params(host)

    def _slice(self, name: str) -> slice:
        o, s, _ = self.segments[name]
        return slice(o, o + s)

    def set_params(self, flat: torch.Tensor) -> None:
        self._pending_head = None  # (a sampling prefix launched ahead read the old proposal networks)
        self.params.copy_(flat.to(self.device, torch.float32))
        self.sync_half()

    @staticmethod
    def _grid_params(net: _NativeModule) -> int:
        levels = (C.c_uint32 * (4 * 32))()
        scales = (C.c_float * 32)()
        _call("nvo_grid_describe", net.handle, levels, scales)
        entries, l = 0, 0
        while l < 32 and levels[4 * l + 1] > 0:
            entries += levels[4 * l + 1]
            l += 1
        return 2 * entries

    def sync_half(self) -> None:
        _call("nvo_cast_working_copy", _stream(self.device), self.n_params, _ptr(self.params), _ptr(self.params_half),
              len(self.bf16_ranges), self._bf16_lo, self._bf16_hi)

    def working_copy_float(self) -> torch.Tensor:
        """The values the kernels actually consume (the 16-bit working copy decoded to fp32): fp16 everywhere in
        f16 mode; in bf16 mode bfloat16 inside bf16_ranges and fp16 elsewhere."""
        out = self.params_half.float()
        for lo, hi in self.bf16_ranges:
            out[lo:hi] = self.params_half[lo:hi].view(torch.bfloat16).float()
        return out

    @torch.no_grad()
    def sync_sharded_state(self, all_reduce) -> None:
        """Sharded optimiser (multi-GPU, all_reduce.shard_optimizer): every rank holds the CURRENT fp32 master weights
        and Adam moments of its own 1/W slice of the fields group only (the 16-bit working copy the kernels read is
        complete everywhere).  Call this before anything that needs the full fp32 state on one rank -- a checkpoint,
        sync_half(), a switch back to the replicated optimiser: the slices are all-gathered in fp32."""
        if all_reduce is None or not getattr(all_reduce, "shard_optimizer", False) or all_reduce.world == 1:
            return
        lo, hi = self.group_ranges["fields"]
        per = (hi - lo) // all_reduce.world
        r = all_reduce.rank
        for buf in (self.params, self.exp_avg, self.exp_avg_sq):
            full = buf[lo:hi]
            all_reduce.all_gather(full, full[r * per:(r + 1) * per].clone())

    def reset_optimizer(self) -> None:
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.opt_steps = {g: 0 for g in self.group_ranges}

    @property
    def opt_steps(self) -> dict:
        """Steps APPLIED to each parameter group so far (torch.optim.Adam's state['step']: a step GradScaler skipped
        does not count).  The counters live on the device (nvo_opt_commit advances them); reading them synchronises."""
        vals = self.dev_applied.tolist()
        return {g: int(vals[self._GROUP_ORDER.index(g)]) for g in self.group_ranges}

    @opt_steps.setter
    def opt_steps(self, steps: dict) -> None:
        vals = self.dev_applied.tolist()
        for g, n in steps.items():
            vals[self._GROUP_ORDER.index(g)] = int(n)
        self.dev_applied.copy_(torch.tensor(vals, dtype=torch.int32))
        self._write_bias(vals)

    def _write_bias(self, applied) -> None:
        """Bias corrections of every group's NEXT applied step (t = applied + 1), as nvo_opt_commit keeps them."""
        # (the kernel receives the betas as fp32 and raises them in double: do exactly that here, or a restored state
        # would differ from the device's own by an ulp of the bias correction)
        b1, b2 = (float(np.float32(b)) for b in self.cfg.adam_betas)
        vals = []
        for n in applied:
            t = int(n) + 1
            vals += [1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t)]
        self.dev_bias.copy_(torch.tensor(vals, dtype=torch.float64).to(torch.float32))

    def current_loss_scale(self) -> float:
        """The loss scale the next step will use (device read-back; static unless cfg.dynamic_loss_scale)."""
        return float(self.dev_loss_scale.item())

    def _loss_scale_ptr(self):
        return self.dev_loss_scale.data_ptr() if self.cfg.dynamic_loss_scale else None

    def _flag_ptr(self, group: str):
        """Overflow flag word of a parameter group (what the producers raise), or None when the optimiser scans."""
        if not self._producer_flags:
            return None
        return self.skip_flag.data_ptr() + 4 * self._GROUP_ORDER.index(group)

    def _use_producer_flags(self, on: bool) -> None:
        modes = self.cfg.grid_bwd_mode if isinstance(self.cfg.grid_bwd_mode, (tuple, list)) else (self.cfg.grid_bwd_mode,) * 3
        # (the leaves of the gradient chain that raise the flag are the slice-owner / tile-local grid kernels)
        on = bool(on and self.cfg.producer_overflow_flags and all(int(m) in (1, 3) for m in modes))
        if on == self._producer_flags and getattr(self, "_producer_flags_set", False):
            return
        self._producer_flags = on
        self._producer_flags_set = True
        self.base_net.set_option("nonfinite_flag_ptr", self._flag_ptr("fields") or 0)
        for m in self.prop_nets:
            m.set_option("nonfinite_flag_ptr", self._flag_ptr("proposal_networks") or 0)

    # ------------------------------------------------------------------------------------------
    # scratch
    # ------------------------------------------------------------------------------------------
    def _workspace(self, R: int, training: bool):
        """Scratch of one ray count.  TRAINING workspaces are kept for good (captured graphs address them by pointer).
        INFERENCE has ONE workspace, sized for the largest chunk seen so far: a smaller chunk (the remainder of an
        image, another resolution) uses the first R rows of every buffer -- the kernels take R as an argument --
        instead of a permanent 1-2 GB workspace per distinct chunk size."""
        key = (R, training)
        if key in self._ws:
            return self._ws[key]
        if not training:
            old = self._ws.get("inference")
            if old is not None and old["R_cap"] >= R:
                old["R"] = R
                return old
            if old is not None:  # grow (a scratch some render_image graph addresses stays alive with that graph)
                del self._ws["inference"]
                if not old.get("pinned"):
                    old.clear()
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        f16 = dict(dtype=self.act_dtype, device=dev)  # 16-bit activations / gradients (fp16 | bf16)
        ws = {"key": key, "R": R, "R_cap": R}
        ws["origins"] = torch.empty(R, 3, **f32)
        ws["directions"] = torch.empty(R, 3, **f32)
        ws["directions_norm"] = torch.empty(R, **f32)
        ws["pixel_area"] = torch.empty(R, **f32)
        ws["cam_idx"] = torch.empty(R, dtype=torch.int32, device=dev)
        ws["gt_rgb"] = torch.empty(R, 3, **f32)
        ws["gt_depth"] = torch.empty(R, **f32)
        ws["gt_normal"] = torch.empty(R, 3, **f32)
        ws["out_normals"] = torch.empty(R, 3, **f32)
        ws["dirs01"] = torch.empty(R, 3, **f32)
        ws["sh"] = torch.empty(R, 16, **f16)
        ws["out_rgb"] = torch.empty(R, 3, **f32)
        ws["out_depth"] = torch.empty(R, **f32)
        ws["out_expected_depth"] = torch.empty(R, **f32)
        ws["out_accumulation"] = torch.empty(R, **f32)
        nets = (*self.prop_nets, self.base_net)
        for k, (S, net) in enumerate(zip(self.levels, nets)):
            N = R * S
            assert N % 16 == 0, "rays x samples must be a multiple of 16"
            ws[f"sbins{k}"] = torch.empty(R, S + 1, **f32)
            ws[f"tbins{k}"] = torch.empty(R, S + 1, **f32)
            ws[f"x{k}"] = torch.empty(N, 3, **f32)
            compact = k < len(self.prop_nets)  # proposal density nets: column 0 only
            ws[f"out{k}"] = torch.empty(N, **f16) if compact else torch.empty(N, 16, **f16)
            ws[f"ctx{k}"] = torch.empty(net.ctx_bytes(N), dtype=torch.uint8, device=dev)
            ws[f"weights{k}"] = torch.empty(N, **f32)
            if training:
                ws[f"dout{k}"] = torch.empty(N, **f16) if compact else torch.empty(N, 16, **f16)
                if self.cfg.optimize_poses:
                    ws[f"dx{k}"] = torch.empty(N, 3, **f32)
        if training and self.cfg.optimize_poses:
            ws["ray_indices"] = torch.zeros(R, 3, dtype=torch.int64, device=dev)
            ws["d_origin"] = torch.empty(R, 3, **f32)
            ws["d_dir"] = torch.empty(R, 3, **f32)
            ws["d_sh"] = torch.empty(R, 16, **f32)
            ws["d_dirs01"] = torch.empty(R, 3, **f32)
        Nm = R * self.levels[-1]
        ws["rgb"] = torch.empty(Nm, 16, **f16)
        if training:
            # colour head: no stored hidden activations (recomputed in the backward)
            ws["drgb"] = torch.empty(Nm, 16, **f16)
            if self.levels[-1] % 16 == 0 and self.cfg.sparse_backward != "off":
                # one byte per 16-sample tile of the main level, written by the render / loss kernel: which tiles carry
                # any gradient at all.  The two MLP backwards behind it walk only those (two of three tiles of a trained
                # field carry none: nvo_main_loss_args::tile_live)
                ws["tile_live"] = torch.zeros(Nm // 16, dtype=torch.uint8, device=dev)
            if self.cfg.deterministic:  # scratch of the fixed-order reductions (colour head, pose gradient)
                ws["color_det"] = torch.empty(int(_lib.lib().nvo_color_det_scratch_bytes(R, self.levels[-1])),
                                              dtype=torch.uint8, device=dev)
                ws["pose_det"] = torch.empty(R, 12, **f32)
        self._ws[key if training else "inference"] = ws
        return ws

    # ------------------------------------------------------------------------------------------
    # forward pieces
    # ------------------------------------------------------------------------------------------
    def _param_ptr(self, name: str, buf: torch.Tensor):
        o, _, _ = self.segments[name]
        return C.c_void_p(buf.data_ptr() + o * buf.element_size())

    def _density_level(self, ws, k: int, net: _NativeModule, seg: str, stream):
        """positions -> contracted grid coords -> NetworkWithInputEncoding -> fp16 [N,16] (col 0)."""
        R, S = ws["R"], self.levels[k]  # ws[f"x{k}"] was written by the sampler kernel of this level
        _call("nvo_fwd", net.handle, stream, R * S, _ptr(ws[f"x{k}"]), self._param_ptr(seg, self.params_half),
              _ptr(ws[f"out{k}"]), _ptr(ws[f"ctx{k}"]))

    def _analytic_normal_grads(self, ws, stream) -> None:
        """d(density pre-activation)/d(x01) of the main field -> ws["dsigma_dx"] [R*48,3]: the quantity
        NerfactoField.get_normals takes with torch.autograd.grad(..., retain_graph=True) (no create_graph,
        so the normals are constants of the loss graph).  One input-only backward of the base network with
        dL/doutput = loss_scale * e_0 -- the same kernels the pose gradient uses."""
        km = len(self.prop_nets)
        N = ws["R"] * self.levels[km]
        if "dsigma_dx" not in ws:
            n_cap = ws["R_cap"] * self.levels[km]
            ws["dsigma_dx"] = torch.empty(n_cap, 3, dtype=torch.float32, device=self.device)
            seed = torch.zeros(n_cap, 16, dtype=self.act_dtype, device=self.device)
            seed[:, 0] = self.cfg.loss_scale
            ws["dsigma_seed"] = seed
        _call("nvo_bwd", self.base_net.handle, stream, N, _ptr(ws[f"x{km}"]),
              self._param_ptr("field.base", self.params_half), _ptr(ws[f"out{km}"]), _ptr(ws["dsigma_seed"]),
              _ptr(ws[f"ctx{km}"]), _ptr(ws["dsigma_dx"]), None)

    def _weights_pdf(self, ws, k: int, anneal: float, jitter, stream, resample: bool, anneal_dev: int | None = None):
        cfg = self.cfg
        R, S = ws["R"], self.levels[k]
        S_out = self.levels[k + 1] if resample else 0
        a = _lib.WeightsPdfArgs(
            R=R, S=S, S_out=S_out, pre=ws[f"out{k}"].data_ptr(), pre_stride=1 if ws[f"out{k}"].dim() == 1 else 16,
            x01=ws[f"x{k}"].data_ptr(),
            sbins=ws[f"sbins{k}"].data_ptr(), tbins=ws[f"tbins{k}"].data_ptr(), density_bias=cfg.density_bias,
            sigma=None, weights=ws[f"weights{k}"].data_ptr(), anneal=anneal,
            histogram_padding=cfg.histogram_padding, near_plane=cfg.near_plane, far_plane=cfg.far_plane,
            jitter=None if jitter is None else jitter.data_ptr(),
            sbins_out=ws[f"sbins{k + 1}"].data_ptr() if resample else None,
            tbins_out=ws[f"tbins{k + 1}"].data_ptr() if resample else None, anneal_dev=anneal_dev,
            origins=ws["origins"].data_ptr() if resample else None,
            directions=ws["directions"].data_ptr() if resample else None,
            x01_out=ws[f"x{k + 1}"].data_ptr() if resample else None, act_bf16=int(self.bf16))
        _call("nvo_weights_pdf", stream, C.byref(a))

    def _forward_head(self, ws, anneal: float, jitters, stream, anneal_dev: int | None = None,
                      skip_first_level: bool = False, levels=None) -> None:
        """Proposal sampling: lin-disp bins -> proposal net 0 -> PDF resample -> proposal net 1 -> PDF resample ->
        positions of the main-field samples.  Reads the PROPOSAL networks' parameters only (what the multi-GPU step
        exploits: this prefix of step k+1 runs while the fields gradient of step k is still being exchanged).
        ``levels``: the proposal levels to run (default: all) -- the multi-GPU step runs level 0 beside the
        reduce-scatter and level 1 beside the all-gather."""
        cfg = self.cfg
        R = ws["R"]
        j = jitters if jitters is not None else (None, None, None)
        if not skip_first_level:  # (nvo_ray_head already wrote the first level's bins and positions)
            _call("nvo_lindisp_positions", stream, R, self.levels[0], cfg.near_plane, cfg.far_plane, _ptr(j[0]),
                  _ptr(ws["origins"]), _ptr(ws["directions"]), _ptr(ws["sbins0"]), _ptr(ws["tbins0"]), _ptr(ws["x0"]))
        for k, net in enumerate(self.prop_nets):
            if levels is not None and k not in levels:
                continue
            self._density_level(ws, k, net, f"proposal.{k}", stream)
            self._weights_pdf(ws, k, anneal, j[k + 1], stream, resample=True, anneal_dev=anneal_dev)

    def _forward(self, ws, training: bool, anneal: float, jitters, cam_idx_for_embedding, embedding_ptr, stream,
                 anneal_dev: int | None = None, skip_head: bool = False):
        """Everything up to (and including) the colour head.  jitters: None or 3 tensors [R].  skip_head: the
        proposal-sampling prefix already ran (``_forward_head``)."""
        R = ws["R"]
        if not training:
            # inference: the hash grids' forwards walk runs of four consecutive samples and gather only where the cell
            # changes (module option grid_fwd_runs, bit-identical) -- the first proposal level's 256 lin-disp samples per
            # ray share cells, and a trained field concentrates the later levels' samples at the surface: per 32 768-ray
            # chunk grid_fwd[L16] 301 -> 250 us, grid_fwd[L5] 175 -> 158 us.  Off for training batches (EXPERIMENTS 9.6b).
            nets = (*self.prop_nets, self.base_net)
            import os
            runs_mask = int(os.environ.get("NVO_RENDER_RUNS", "7"))  # A/B: bit k = network k walks runs (proposal 0, 1, main)
            for k_, m in enumerate(nets):
                m.set_option("grid_fwd_runs", (runs_mask >> k_) & 1)
            try:
                return self._forward_body(ws, training, anneal, jitters, cam_idx_for_embedding, embedding_ptr, stream,
                                          anneal_dev, skip_head)
            finally:
                for m in nets:
                    m.set_option("grid_fwd_runs", 0)
        return self._forward_body(ws, training, anneal, jitters, cam_idx_for_embedding, embedding_ptr, stream, anneal_dev,
                                  skip_head)

    def _forward_body(self, ws, training: bool, anneal: float, jitters, cam_idx_for_embedding, embedding_ptr, stream,
                      anneal_dev: int | None = None, skip_head: bool = False):
        R = ws["R"]
        if not skip_head:
            self._forward_head(ws, anneal, jitters, stream, anneal_dev=anneal_dev)
        km = len(self.prop_nets)
        self._density_level(ws, km, self.base_net, "field.base", stream)
        if not ws.get("dirs01_ready", False):
            _call("nvo_dirs01", stream, 3 * R, _ptr(ws["directions"]), _ptr(ws["dirs01"]))
        ws["dirs01_ready"] = False
        if not ws.get("sh_ready", False):
            _call("nvo_sh_encode_t", stream, R, 4, _ptr(ws["dirs01"]), _ptr(ws["sh"]), int(self.bf16))
        ws["sh_ready"] = False
        ca = self._color_args(ws, training, cam_idx_for_embedding, embedding_ptr)
        _call("nvo_nerfacto_color_fwd", stream, C.byref(ca))
        return ca

    def _color_args(self, ws, training, cam_idx, embedding_ptr):
        km = len(self.prop_nets)
        return _lib.ColorArgs(
            R=ws["R"], S=self.levels[-1], sh=ws["sh"].data_ptr(), base_out=ws[f"out{km}"].data_ptr(),
            embedding=embedding_ptr, cam_idx=None if cam_idx is None else cam_idx.data_ptr(),
            weights=self._param_ptr("field.color", self.params_half).value, rgb=ws["rgb"].data_ptr(),
            hidden=None,
            drgb=ws["drgb"].data_ptr() if training else None,
            d_base_out=ws[f"dout{km}"].data_ptr() if training else None,
            d_embedding=self._param_ptr("field.embedding", self.grads).value if training else None,
            d_sh=ws["d_sh"].data_ptr() if (training and "d_sh" in ws) else None,
            d_weights=self._param_ptr("field.color", self.grads).value if training else None,
            act_bf16=int(self.bf16),
            det_scratch=ws["color_det"].data_ptr() if (training and "color_det" in ws) else None,
            nonfinite_flag=self._flag_ptr("fields") if training else None,
            dw_replicas=self._dw_replica_plan()["color"][0] if (training and self._dw_replica_plan()) else None,
            n_dw_replicas=self._dw_replica_plan()["color"][1] if (training and self._dw_replica_plan()) else 0,
            det_scratch_bytes=ws["color_det"].numel() if (training and "color_det" in ws) else 0,
            n_cameras=self.cfg.num_images,
            tile_live=ws["tile_live"].data_ptr() if (training and ws.get("sparse")) else None,
            tile_live_count=(self.losses.data_ptr() + 7 * 4) if (training and ws.get("sparse")) else None)

    def _main_loss_args(self, ws, training: bool, has_depth: bool, normals: bool = False,
                        has_gt_normal: bool = False):
        cfg = self.cfg
        km = len(self.prop_nets)
        R, S = ws["R"], self.levels[-1]
        inv_rays = 1.0 / (R * self.world_size)
        n_levels = len(self.levels)
        return _lib.MainLossArgs(
            R=R, S=S, pre=ws[f"out{km}"].data_ptr(), pre_stride=16, rgb=ws["rgb"].data_ptr(), rgb_stride=16,
            x01=ws[f"x{km}"].data_ptr(), sbins=ws[f"sbins{km}"].data_ptr(), tbins=ws[f"tbins{km}"].data_ptr(),
            density_bias=cfg.density_bias, gt_rgb=ws["gt_rgb"].data_ptr() if training else None,
            gt_depth=ws["gt_depth"].data_ptr() if (training and has_depth) else None,
            directions_norm=ws["directions_norm"].data_ptr(), rgb_mult=cfg.rgb_loss_mult,
            distortion_mult=cfg.distortion_loss_mult, depth_mult=cfg.depth_loss_mult if has_depth else 0.0,
            depth_sigma=cfg.depth_sigma, inv_rays=inv_rays, depth_level_div=1.0 / n_levels,
            loss_scale=cfg.loss_scale, out_rgb=ws["out_rgb"].data_ptr(), out_depth=ws["out_depth"].data_ptr(),
            out_expected_depth=ws["out_expected_depth"].data_ptr(),
            out_accumulation=ws["out_accumulation"].data_ptr(), weights=ws[f"weights{km}"].data_ptr(),
            losses=self.losses.data_ptr() if training else None,
            dpre=ws[f"dout{km}"].data_ptr() if training else None, dpre_stride=16,
            drgb=ws["drgb"].data_ptr() if training else None, drgb_stride=16,
            dsigma_dx=ws["dsigma_dx"].data_ptr() if normals else None, dsigma_inv_scale=1.0 / cfg.loss_scale,
            gt_normal=ws["gt_normal"].data_ptr() if (normals and training and has_gt_normal) else None,
            normal_mult=cfg.normal_loss_mult if (normals and has_gt_normal) else 0.0,
            out_normals=ws["out_normals"].data_ptr() if normals else None, act_bf16=int(self.bf16),
            loss_scale_dev=self._loss_scale_ptr() if training else None,
            nonfinite_flag=self._flag_ptr("fields") if training else None,
            tile_live=ws["tile_live"].data_ptr() if (training and ws.get("sparse")) else None)

    # ------------------------------------------------------------------------------------------
    # schedules (nerfacto callbacks)
    # ------------------------------------------------------------------------------------------
    def anneal_at(self, step: int) -> float:
        n = self.cfg.proposal_weights_anneal_max_num_iters
        frac = min(max(step / n, 0.0), 1.0)
        b = self.cfg.proposal_weights_anneal_slope
        return b * frac / ((b - 1) * frac + 1)

    def proposal_update_due(self, step: int) -> bool:
        cfg = self.cfg
        sched = min(max(step / cfg.proposal_warmup * cfg.proposal_update_every, 1.0), float(cfg.proposal_update_every))
        return self.steps_since_proposal_update > sched or step < 10

    def camera_lr(self, step: int) -> float:
        """ExponentialDecayScheduler(lr_final=1e-5, max_steps=mapping_iterations) on camera_opt."""
        cfg = self.cfg
        t = min(max(step / max(cfg.max_num_iterations, 1), 0.0), 1.0)
        return math.exp(math.log(cfg.lr_camera) * (1 - t) + math.log(cfg.lr_camera_final) * t)

    # ------------------------------------------------------------------------------------------
    # one optimisation step
    # ------------------------------------------------------------------------------------------
    def load_rays(self, ws, ray_indices, intrinsics, c2w, images, depths, corrections=None, normals=None):
        """ray_indices [R,3] int64 (camera,y,x) -> origins/directions/cam idx + gathered targets.
        ``normals``: optional [n,H,W,3] world-space normal images in the (n+1)/2 colour space
        (DynamicDataset.get_dataset()['normal_image'])."""
        stream = _stream(self.device)
        R = ws["R"]
        H, W = images.shape[1], images.shape[2]
        self._pending_head = None  # (a prefix launched ahead by a pipelined graph is overwritten here)
        if corrections is None and self.cfg.optimize_poses:
            # CameraOptimizer.forward: exp_map_SE3(pose_adjustment) for every camera, gathered by raygen
            _call("nvo_pose_exp_map", stream, self.cfg.num_images,
                  self._param_ptr("camera_opt.pose_adjustment", self.params), _ptr(self.corrections),
                  self._pose_mode())
            corrections = self.corrections
            if "ray_indices" in ws:
                ws["ray_indices"].copy_(ray_indices)
            self._pose_inputs = (intrinsics, c2w)
        _call("nvo_raygen", stream, R, _ptr(ray_indices), _ptr(intrinsics), _ptr(c2w), _ptr(corrections),
              _ptr(ws["origins"]), _ptr(ws["directions"]), _ptr(ws["directions_norm"]), _ptr(ws["pixel_area"]),
              _ptr(ws["cam_idx"]))
        # colour / depth / normal targets and the direction-encoding input in one launch
        _call("nvo_gather_targets", stream, R, _ptr(ray_indices), H, W, _ptr(images), _ptr(depths), _ptr(normals),
              _ptr(ws["directions"]), _ptr(ws["gt_rgb"]), _ptr(ws["gt_depth"]), _ptr(ws["gt_normal"]),
              _ptr(ws["dirs01"]))
        ws["dirs01_ready"] = True
        ws["sh_ready"] = False  # (may be left set by the capture of a pipelined graph, which ends with a sampling prefix)

    def load_ray_bundle(self, ws, origins, directions, directions_norm, cam_idx, gt_rgb=None, gt_depth=None,
                        gt_normal=None):
        """Inject an existing ray bundle (+ targets) instead of generating rays from pixel indices."""
        self._pending_head = None
        ws["dirs01_ready"] = ws["sh_ready"] = False
        ws["origins"].copy_(origins)
        ws["directions"].copy_(directions)
        ws["directions_norm"].copy_(directions_norm.reshape(-1))
        ws["cam_idx"].copy_(cam_idx.reshape(-1).to(torch.int32))
        if gt_rgb is not None:
            ws["gt_rgb"].copy_(gt_rgb)
        if gt_depth is not None:
            ws["gt_depth"].copy_(gt_depth.reshape(-1))
        if gt_normal is not None:
            ws["gt_normal"].copy_(gt_normal)

    def forward_backward(self, ws, jitters, has_depth: bool = True, update_proposals: bool | None = None,
                         anneal: float | None = None, anneal_dev: int | None = None, has_normals: bool = False,
                         skip_head: bool = False, proposal_values: bool | None = None, sparse: bool | None = None):
        """Forward + losses + backward for the rays loaded into ``ws``.  Fills self.grads (scaled by
        loss_scale) and self.losses; does NOT touch the parameters.  ``sparse``: EngineConfig.sparse_backward's kind of
        step (None: what the configuration / the last probe says)."""
        cfg = self.cfg
        stream = _stream(self.device)
        step = self.step
        ws["sparse"] = bool(self._sparse_default() if sparse is None else sparse) and "tile_live" in ws
        self._use_producer_flags(self.world_size == 1 and getattr(self, "_reducer", None) is None)
        if anneal is None:
            anneal = self.anneal_at(step)
        if update_proposals is None:
            update_proposals = self.proposal_update_due(step)
        if proposal_values is None:
            proposal_values = self.proposal_values_due(step, bool(update_proposals))
        # Only what is ACCUMULATED into needs zeroing (colour-head dW, appearance-embedding gradient): every other
        # gradient range is overwritten by its producer (grid slices by plain stores, MLP dW zeroed by nvo_bwd,
        # pose gradient by the exp-map backward).  Ranges of groups that do not train this step keep stale values
        # and are neither reduced, checked nor applied.
        pose_active = cfg.optimize_poses and "d_sh" in ws and self._pose_inputs is not None
        if not ws.pop("zeroed_by_head", False):  # (one-graph step: the ray head's launch already cleared them)
            self._zero_step_buffers(ws, bool(update_proposals), bool(pose_active), stream)
        emb_ptr = self._param_ptr("field.embedding", self.params_half).value
        ca = self._forward(ws, True, anneal, jitters, ws["cam_idx"], emb_ptr, stream, anneal_dev=anneal_dev,
                           skip_head=skip_head)
        km = len(self.prop_nets)
        R = ws["R"]
        pose = cfg.optimize_poses and "d_sh" in ws and self._pose_inputs is not None
        # monosdf normal supervision (enhancement 'normal' modes): needs the analytic normals; without a
        # target the predict_normals heads contribute exactly zero loss and are not evaluated
        normals = has_normals and cfg.normal_loss_mult > 0.0
        if normals:
            self._analytic_normal_grads(ws, stream)
        la = self._main_loss_args(ws, True, has_depth, normals=normals, has_gt_normal=normals)
        _call("nvo_main_render_loss", stream, C.byref(la))
        side = None
        # Roles of the two streams on an update step.  The proposal chain (losses, two MLP backwards, two hash-grid
        # scatters) is the LONGER one; with fixed poses it stays on the origin stream and the main-field backward goes
        # to the side stream, so that the proposal chain may fork once more (network 1's MLP backward beside network 0's
        # chain) -- a fork from a stream that is itself a fork crashes hipStreamEndCapture on
        # ROCm 7.2.  With pose optimisation the main-field backward forks (its grid scatter beside the pose chain) and
        # therefore keeps the origin stream, the proposal chain the side stream, unforked.
        swap = bool(update_proposals and cfg.overlap_proposal_backward and not pose and not cfg.deterministic)
        self._prop_fork_ok = swap
        if update_proposals and cfg.overlap_proposal_backward:
            # fork: everything the proposal backward reads (main-level weights / bins) exists now
            cur = torch.cuda.current_stream(self.device)
            if self._side_stream is None:
                # ONE side stream for both proposal networks (round 1: a stream each measured slower, 1.10 vs 1.01
                # ms/step -- three LDS-heavy scatter kernels at once thrash)
                self._side_stream = [torch.cuda.Stream(device=self.device)]
            side = self._side_stream
            if not swap:
                for si, st in enumerate(side):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        self._proposal_backward(ws, has_depth, pose, _stream(self.device),
                                                levels=None if len(side) == 1 else [si])

        def main_backward(st):
            _call("nvo_nerfacto_color_bwd", st, C.byref(ca))
            if ws.get("sparse"):
                # the base network's dL/doutput = {dpre | the colour head's dX}: zero wherever neither bit is set.  Only for
                # THIS backward (the analytic-normal pass runs the same module on another dL/doutput)
                self.base_net.set_option("bwd_tile_live_ptr", ws["tile_live"].data_ptr())
                self.base_net.set_option("bwd_tile_live_bits", 3)
                self.base_net.set_option("bwd_tile_live_count_ptr", self.losses.data_ptr() + 7 * 4)
            try:
                return main_backward_base(st)
            finally:
                if ws.get("sparse"):
                    self.base_net.set_option("bwd_tile_live_ptr", 0)

        def main_backward_base(st):
            if pose and cfg.overlap_pose_backward:
                # the pose chain below only needs dL/dx of the main field: the long parameter scatter of its hash grid
                # runs beside it on another stream (forked inside the call, joined at the end of this function)
                if self._scatter_stream is None:
                    self._scatter_stream = torch.cuda.Stream(device=self.device)
                _call("nvo_bwd_fork", self.base_net.handle, st, C.c_void_p(self._scatter_stream.cuda_stream),
                      R * self.levels[km], _ptr(ws[f"x{km}"]), self._param_ptr("field.base", self.params_half),
                      _ptr(ws[f"out{km}"]), _ptr(ws[f"dout{km}"]), _ptr(ws[f"ctx{km}"]), _ptr(ws[f"dx{km}"]),
                      self._param_ptr("field.base", self.grads))
                return self._scatter_stream
            _call("nvo_bwd", self.base_net.handle, st, R * self.levels[km], _ptr(ws[f"x{km}"]),
                  self._param_ptr("field.base", self.params_half), _ptr(ws[f"out{km}"]), _ptr(ws[f"dout{km}"]),
                  _ptr(ws[f"ctx{km}"]), _ptr(ws[f"dx{km}"]) if pose else None,
                  self._param_ptr("field.base", self.grads))
            return None

        if swap:
            side[0].wait_stream(cur)
            with torch.cuda.stream(side[0]):
                scatter_stream = main_backward(_stream(self.device))
            self._proposal_backward(ws, has_depth, pose, stream)
        else:
            scatter_stream = main_backward(stream)
        if update_proposals and side is None:
            self._proposal_backward(ws, has_depth, pose, stream)
        if proposal_values and not update_proposals:
            self._proposal_backward(ws, has_depth, False, stream, values_only=True)
        cur = torch.cuda.current_stream(self.device)
        if side is not None:
            for st in side:
                cur.wait_stream(st)  # join
        if pose:
            self._pose_backward(ws, update_proposals, stream)
        if scatter_stream is not None:
            cur.wait_stream(scatter_stream)  # join
        self._fold_dw_replicas(stream)
        return update_proposals

    # ---- EngineConfig.sparse_backward -----------------------------------------------------------------------------------
    _SPARSE_PROBE_EVERY = 64

    def _sparse_default(self) -> bool:
        mode = self.cfg.sparse_backward
        return mode == "on" or (mode == "auto" and bool(getattr(self, "_sparse_mode", False)))

    def _sparse_for_step(self, step: int) -> bool:
        """Kind of the step about to run ("auto": every 64th step is a sparse one whatever the mode -- the probe)."""
        mode = self.cfg.sparse_backward
        if mode != "auto":
            return mode == "on"
        self._sparse_poll()
        return bool(getattr(self, "_sparse_mode", False)) or step % self._SPARSE_PROBE_EVERY == 0

    def _sparse_probe_after(self, step: int, R: int) -> None:
        """Behind a sparse step of the "auto" mode: ask (asynchronously) how many of its tiles were live."""
        if self.cfg.sparse_backward != "auto" or step % self._SPARSE_PROBE_EVERY != 0 or getattr(self, "_sparse_pending", None):
            return
        if getattr(self, "_sparse_host", None) is None:
            self._sparse_host = torch.zeros(64, dtype=torch.float32).pin_memory()
            self._sparse_event = torch.cuda.Event()
        self._sparse_host.copy_(self.losses[:, 7], non_blocking=True)  # (slot 7 of the loss shards: nvo_main_loss_args::tile_live)
        self._sparse_event.record()
        self._sparse_pending = R * self.levels[-1] // 16

    def _sparse_poll(self) -> None:
        n_tiles = getattr(self, "_sparse_pending", None)
        if not n_tiles or not self._sparse_event.query():
            return
        frac = float(self._sparse_host.sum()) / n_tiles
        self._sparse_pending = None
        self._sparse_live_frac = frac
        if frac < 0.60:
            self._sparse_mode = True
        elif frac > 0.75:
            self._sparse_mode = False

    def _dw_replica_plan(self):
        """Zeroed copies of the MLP weight-gradient ranges (EngineConfig.dw_replicas) + the table nvo_fold_replicas takes."""
        if getattr(self, "_dw_rep", None) is not None:
            return self._dw_rep
        cfg = self.cfg
        G = int(cfg.dw_replicas)
        if G <= 0 or cfg.deterministic:
            self._dw_rep = False
            return False
        nets = [("field.base", self.base_net, self.base_net.n_params - self._grid_params(self.base_net)),
                ("field.color", None, self.segments["field.color"][1])]
        nets += [(f"proposal.{k}", m, m.n_params - self._grid_params(m)) for k, m in enumerate(self.prop_nets)]
        buf = torch.zeros(G * sum(n for _, _, n in nets), dtype=torch.float32, device=self.device)
        reps, dsts, off = [], [], 0
        for seg, net, n in nets:
            ptr = buf.data_ptr() + 4 * off
            if net is not None:  # (the MLP's weights lead the module's parameter block)
                net.set_option("dw_replicas_ptr", ptr)
                net.set_option("dw_replicas", G)
            reps.append(ptr)
            dsts.append(self._param_ptr(seg, self.grads).value)
            off += G * n
        k = len(nets)
        self._dw_rep = {"buf": buf, "G": G, "k": k, "color": (reps[1], G),
                        "reps": (C.c_void_p * k)(*reps), "n_rep": (C.c_uint32 * k)(*([G] * k)),
                        "n": (C.c_uint64 * k)(*[n for _, _, n in nets]), "dst": (C.c_void_p * k)(*dsts)}
        return self._dw_rep

    def _fold_dw_replicas(self, stream) -> None:
        plan = self._dw_replica_plan()
        if plan:
            _call("nvo_fold_replicas", stream, plan["k"], plan["reps"], plan["n_rep"], plan["n"], plan["dst"])

    def _net_zero_ranges(self, net, seg: str):
        """What nvo_bwd of ``net`` clears before it accumulates (MLP weight gradient, atomically flushed grid ranges,
        scale scratch), handed over to the step's single zero launch: module option external_zero."""
        cap = 16
        ptrs = (C.c_void_p * cap)()
        sizes = (C.c_uint64 * cap)()
        n = _lib.lib().nvo_bwd_zero_ranges(net.handle, self._param_ptr(seg, self.grads), ptrs, sizes, cap)
        if n < 0:
            raise RuntimeError(f"nvo_bwd_zero_ranges({seg}): {_lib.lib().nvo_last_error().decode()}")
        net.set_option("external_zero", 1)
        return [(int(ptrs[i]), int(sizes[i])) for i in range(n)]

    def _zero_step_buffers(self, ws, update_proposals: bool, pose: bool, stream) -> None:
        """Everything a step ACCUMULATES into is cleared by ONE launch at its start (colour-head dW, appearance-embedding
        gradient, loss shards, the MLP weight gradients and atomically flushed grid ranges of the networks that run
        backward, the pose-chain reductions).  It used to be ~9 launches (three torch fills + the zeroing inside every
        nvo_bwd), each ~5 us of dependent dispatch + drain in the replayed graph: 30-40 us per step.  Every other
        gradient range is overwritten by its producer; ranges of groups that do not train this step keep stale values
        and are neither reduced, checked nor applied."""
        if self._zero_plan is None:
            plan = {"always": [], "proposals": [], "pose": {}}
            for name in ("field.color", "field.embedding"):
                o, sz, _ = self.segments[name]
                plan["always"].append((self.grads.data_ptr() + 4 * o, 4 * sz))
            plan["always"].append((self.losses.data_ptr(), self.losses.numel() * 4))
            plan["always"].append((self.skip_flag.data_ptr(), self.skip_flag.numel() * 4))
            plan["always"] += self._net_zero_ranges(self.base_net, "field.base")
            for k, net in enumerate(self.prop_nets):
                plan["proposals"] += self._net_zero_ranges(net, f"proposal.{k}")
            self._zero_plan = plan
        ranges = list(self._zero_plan["always"])
        if update_proposals:
            ranges += self._zero_plan["proposals"]
        if pose:
            for name in ("d_sh", "d_origin", "d_dir"):
                ranges.append((ws[name].data_ptr(), ws[name].numel() * 4))
            ranges.append((self.d_corrections.data_ptr(), self.d_corrections.numel() * 4))
        ptrs = (C.c_void_p * len(ranges))(*[p for p, _ in ranges])
        sizes = (C.c_uint64 * len(ranges))(*[b for _, b in ranges])
        if stream is None:  # (the caller launches: nvo_ray_head_zero)
            return len(ranges), ptrs, sizes
        _call("nvo_zero_ranges", stream, len(ranges), ptrs, sizes)

    def proposal_values_due(self, step: int, updated: bool) -> bool:
        """A non-update step on which loss_dict must still carry the interlevel / proposal-level depth VALUES."""
        mode = self.cfg.proposal_loss_values
        if updated or mode == "never":
            return False
        return mode == "always" or step % max(1, int(self.cfg.log_every)) == 0

    def _proposal_backward(self, ws, has_depth: bool, pose: bool, stream, levels=None, values_only: bool = False) -> None:
        """Interlevel + depth loss of both proposal levels and their network backward (``values_only``: the loss
        VALUES alone -- what nerfstudio reports on steps where the proposal networks do not train)."""
        cfg = self.cfg
        R = ws["R"]
        km = len(self.prop_nets)
        inv_rays = 1.0 / (R * self.world_size)
        def loss_args(k):
            return _lib.PropLossArgs(
                R=R, S=self.levels[k], S_main=self.levels[km], pre=ws[f"out{k}"].data_ptr(), pre_stride=1,
                x01=ws[f"x{k}"].data_ptr(), sbins=ws[f"sbins{k}"].data_ptr(), tbins=ws[f"tbins{k}"].data_ptr(),
                sbins_main=ws[f"sbins{km}"].data_ptr(), weights_main=ws[f"weights{km}"].data_ptr(),
                density_bias=cfg.density_bias, gt_depth=ws["gt_depth"].data_ptr() if has_depth else None,
                directions_norm=ws["directions_norm"].data_ptr(), interlevel_mult=cfg.interlevel_loss_mult,
                depth_mult=cfg.depth_loss_mult if has_depth else 0.0, depth_sigma=cfg.depth_sigma,
                inv_rays=inv_rays, depth_level_div=1.0 / len(self.levels), loss_scale=cfg.loss_scale,
                losses=self.losses.data_ptr() + 3 * 4, dpre=None if values_only else ws[f"dout{k}"].data_ptr(),
                dpre_stride=1, act_bf16=int(self.bf16), loss_scale_dev=self._loss_scale_ptr(),
                nonfinite_flag=None if values_only else self._flag_ptr("proposal_networks"))

        # both levels' loss kernels are independent of each other (each is one round of 4096 waves that lasts as long as
        # one ray's dependent chain): ONE launch when both run on this stream
        paired = levels is None and len(self.prop_nets) == 2
        if paired:
            pa0, pa1 = loss_args(0), loss_args(1)
            _call("nvo_prop_loss_pair", stream, C.byref(pa0), C.byref(pa1))
        # network 1's MLP backward beside network 0's chain: both losses are done (paired launch), the networks share nothing
        mlp_stream = None
        if paired and not values_only and getattr(self, "_prop_fork_ok", False):
            if getattr(self, "_prop_mlp_stream", None) is None:
                self._prop_mlp_stream = torch.cuda.Stream(device=self.device)
            mlp_stream = self._prop_mlp_stream
            mlp_stream.wait_stream(torch.cuda.current_stream(self.device))
        # (forked form: the SHORT network -- level 1, 393 K samples -- runs its whole backward on `stream` while the long
        # MLP backward of level 0 -- 1 M samples -- hides beside it on mlp_stream; level 0's scatter then follows)
        order = [1, 0] if mlp_stream is not None else list(range(len(self.prop_nets)))
        for k in order:
            net = self.prop_nets[k]
            if levels is not None and k not in levels:
                continue
            if not paired:
                pa = loss_args(k)
                _call("nvo_prop_loss", stream, C.byref(pa))
            if values_only:
                continue
            args = (R * self.levels[k], _ptr(ws[f"x{k}"]), self._param_ptr(f"proposal.{k}", self.params_half),
                    _ptr(ws[f"out{k}"]), _ptr(ws[f"dout{k}"]), _ptr(ws[f"ctx{k}"]), _ptr(ws[f"dx{k}"]) if pose else None,
                    self._param_ptr(f"proposal.{k}", self.grads))
            if mlp_stream is not None and k == 0:
                # (the call forks inside: MLP backward [+ input gradient] on mlp_stream, the grid's parameter scatter on
                # `stream` behind an event -- i.e. behind network 1's scatter, which is already queued there)
                _call("nvo_bwd_fork", net.handle, C.c_void_p(mlp_stream.cuda_stream), stream, *args)
                torch.cuda.current_stream(self.device).wait_stream(mlp_stream)  # join (pose: the input gradient)
            else:
                _call("nvo_bwd", net.handle, stream, *args)

    def _pose_backward(self, ws, update_proposals: bool, stream) -> None:
        """dL/dx01 of every level that ran backward + the SH direction gradient -> dL/dpose_adjustment
        (written into the camera_opt range of self.grads, loss-scaled like every other gradient)."""
        cfg = self.cfg
        R = ws["R"]
        km = len(self.prop_nets)
        levels = ([0, 1] if update_proposals else []) + [km]
        for k in levels:
            _call("nvo_positions_bwd", stream, R, self.levels[k], _ptr(ws["origins"]), _ptr(ws["directions"]),
                  _ptr(ws[f"tbins{k}"]), _ptr(ws[f"dx{k}"]), _ptr(ws["d_origin"]), _ptr(ws["d_dir"]))
        _call("nvo_sh_bwd_input_f32", stream, R, 4, _ptr(ws["dirs01"]), _ptr(ws["d_sh"]), _ptr(ws["d_dirs01"]))
        intr, c2w = self._pose_inputs
        if "pose_det" in ws:  # deterministic mode: per-ray contributions summed per camera in a fixed order
            _call("nvo_pose_bwd_det", stream, R, _ptr(ws["ray_indices"]), _ptr(intr), _ptr(c2w), _ptr(ws["d_origin"]),
                  _ptr(ws["d_dir"]), _ptr(ws["d_dirs01"]), _ptr(self.d_corrections), _ptr(ws["pose_det"]), cfg.num_images)
        else:
            _call("nvo_pose_bwd_cams", stream, R, _ptr(ws["ray_indices"]), _ptr(intr), _ptr(c2w), _ptr(ws["d_origin"]),
                  _ptr(ws["d_dir"]), _ptr(ws["d_dirs01"]), _ptr(self.d_corrections), cfg.num_images)
        # regulariser: its value goes to loss slot 5 of shard 0, its gradient is scaled like the rest
        dyn = cfg.dynamic_loss_scale
        reg_scale = (1.0 if dyn else cfg.loss_scale) / self.world_size
        _call("nvo_se3_exp_map_bwd_scaled", stream, cfg.num_images,
              self._param_ptr("camera_opt.pose_adjustment", self.params), _ptr(self.d_corrections),
              cfg.camera_trans_l2_penalty, cfg.camera_rot_l2_penalty, reg_scale,
              self._param_ptr("camera_opt.pose_adjustment", self.grads),
              C.c_void_p(self.losses.data_ptr() + 5 * 4), self._pose_mode(), _ptr(self.dev_loss_scale) if dyn else None)

    def _pose_mode(self) -> int:
        return {"SE3": 0, "SO3xR3": 1}[self.cfg.camera_mode]

    _GROUP_ORDER = ("fields", "proposal_networks", "camera_opt")

    def _group_lr(self, g: str) -> float:
        cfg = self.cfg
        return {"fields": cfg.lr_fields, "proposal_networks": cfg.lr_proposal,
                "camera_opt": self.camera_lr(self.step)}[g]

    def optimizer_step(self, groups=("fields", "proposal_networks", "camera_opt"), from_device_scalars=False,
                       grads_half: torch.Tensor | None = None, flags_cleared: bool = False, step_groups=None,
                       shard: tuple | None = None, check: bool = True) -> None:
        """Non-finite check + fused Adam + commit, per group (one launch each for all groups).  GradScaler semantics
        (/root/reference/nerf_vo/mapping/nerfstudio.py:59, mixed_precision=True): a group whose gradients hold a non-finite
        value is skipped and its step counter does not advance (the counters live on the device: dev_applied); with
        cfg.dynamic_loss_scale the scale backs off / grows like torch's GradScaler.update().
        ``from_device_scalars``: the learning rates are read from self.dev_scalars (filled by _write_step_scalars)
        instead of kernel arguments -- the form a captured graph replays.
        ``step_groups``: all groups this STEP trains (default: ``groups``) -- a step that runs its optimisers in two
        calls hands it to the LAST one, which then updates the loss scale from all of the step's flags.
        ``shard``: (rank, world) -- this rank's 1/world slice of the FIELDS group only (sharded optimiser: the other ranks
        step the other slices and the 16-bit working copy is all-gathered afterwards).
        ``check``: False when the caller already raised the flag words (sharded exchange: they travel with the gradient)."""
        cfg = self.cfg
        stream = _stream(self.device)
        # grads_half: the 2-byte (bf16 | fp16) buffer a compressed all-reduce left behind -- consumed directly
        gbuf, gsz, ghalf = (self.grads, 4, 0) if grads_half is None else (
            grads_half, 2, 2 if grads_half.dtype == torch.bfloat16 else 1)
        active = [g for g in groups if g != "camera_opt" or cfg.optimize_poses]
        order = self._GROUP_ORDER

        def span(g):
            lo, hi = self.group_ranges[g]
            if shard is not None and g == "fields":
                rank, world = shard
                per = (hi - lo) // world
                assert per * world == hi - lo, "fields group is padded to a multiple of 8 x 64 elements"
                lo, hi = lo + rank * per, lo + (rank + 1) * per
            return lo, hi

        if check and self._producer_flags:
            # The producers raised the flags of the fields / proposal groups at the roots (loss kernels), at the leaves
            # (grid backward) AND inside the 16-bit gradient chain: a hidden dZ or d_base_out can overflow while the roots
            # are finite -- GradScaler keeps doubling the scale until something does -- and a ReLU backward may drop it
            # before it reaches a leaf, but the layer it appears in multiplies it into that layer's weight gradient
            # (dW = dZ^T H; inf * 0 = NaN), so every fused-MLP backward checks the dW totals it flushes
            # (NvoMlpArgsT::nf_flag; the appearance-embedding gradient is non-finite only together with the colour head's
            # dW0).  (Found by the fixed-pose 8192-step run of round 4, which nothing checked: NaN weights after the
            # scale reached 2^19; first fixed with a scan launch over the non-grid ranges, 4.6 us on every step.)  Only
            # the pose gradients, which no fused MLP produces, are still scanned.
            spans = []
            for g in active:
                gi = order.index(g)
                if g not in ("fields", "proposal_networks"):
                    lo, hi = self.group_ranges[g]
                    spans.append((lo, hi - lo, gi))
            if shard is not None:
                spans = [sp for sp in spans if sp[2] != order.index("fields")]
            if spans:
                offs = (C.c_uint64 * len(spans))(*[sp[0] for sp in spans])
                sizes = (C.c_uint64 * len(spans))(*[sp[1] for sp in spans])
                slots = (C.c_uint32 * len(spans))(*[sp[2] for sp in spans])
                _call("nvo_nonfinite_flag_spans_or", stream, len(spans), offs, sizes, slots, _ptr(gbuf), ghalf, _ptr(self.skip_flag))
        elif check:
            # one flag PER GROUP that trains this step (GradScaler.step decides per optimiser; ranges of idle groups
            # hold stale values and are neither checked nor applied), all in one launch; flag word = the group's slot
            offs = (C.c_uint64 * len(order))(*[span(g)[0] if g in active else 0 for g in order])
            sizes = (C.c_uint64 * len(order))(*[span(g)[1] - span(g)[0] if g in active else 0 for g in order])
            # (flags_cleared: the step's single zero launch already cleared the flag words -- one launch less)
            if not flags_cleared:
                slots = [order.index(g) for g in active]
                for sl in slots:  # only the words of the groups checked here (another call may own the others)
                    _call("nvo_zero_ranges", stream, 1, (C.c_void_p * 1)(self.skip_flag.data_ptr() + 4 * sl), (C.c_uint64 * 1)(4))
            _call("nvo_nonfinite_flag_ranges_or", stream, len(order), offs, sizes, _ptr(gbuf), ghalf, _ptr(self.skip_flag))
        batch = []
        mask = 0
        fused = getattr(self, "_fused_adam_range", None)  # (set around the capture of the one-graph step)
        for g in active:
            lo, hi = span(g)
            gi = order.index(g)
            mask |= 1 << gi
            hyper = self.dev_scalars.data_ptr() + 4 * (1 + 3 * gi) if from_device_scalars else None
            parts = [(lo, hi)]
            if g == "fields" and fused is not None:
                # the main grid's backward has already stepped [fused): the launch covers what lies around it
                assert shard is None and grads_half is None and lo <= fused[0] < fused[1] <= hi
                parts = [(lo, fused[0]), (fused[1], hi)]
            for a_, b_ in parts:
                if b_ > a_:
                    batch.append(_lib.AdamGroup(offset=a_, n=b_ - a_, lr=self._group_lr(g), step=0, hyper_dev=hyper,
                                                bias_dev=self.dev_bias.data_ptr() + 8 * gi, flag_slot=gi, flag_slot_set=1))
        if not batch:
            return
        arr = (_lib.AdamGroup * len(batch))(*batch)
        dyn = cfg.dynamic_loss_scale
        _call("nvo_adam_step_groups_scaled", stream, len(batch), arr, _ptr(self.params), _ptr(self.params_half), _ptr(gbuf),
              ghalf, _ptr(self.exp_avg), _ptr(self.exp_avg_sq), cfg.adam_betas[0], cfg.adam_betas[1], cfg.adam_eps,
              1.0 / cfg.loss_scale, 0.0, _ptr(self.skip_flag), len(self.bf16_ranges), self._bf16_lo, self._bf16_hi,
              _ptr(self.dev_loss_scale) if dyn else None)
        scale_mask = 0
        if dyn and "fields" in active:  # the fields group is stepped last in every launch order
            for g in (step_groups if step_groups is not None else active):
                if g != "camera_opt" or cfg.optimize_poses:
                    scale_mask |= 1 << order.index(g)
        if self._defer_commit is not None:
            # (capture of the one-graph step: the commit rides in the eager launch behind the replay, _commit_and_write)
            self._defer_commit.append((mask, scale_mask))
            return
        _call("nvo_opt_commit", stream, len(order), mask, scale_mask, _ptr(self.dev_applied), _ptr(self.skip_flag),
              _ptr(self.dev_loss_scale) if scale_mask else None, _ptr(self.dev_growth_tracker) if scale_mask else None,
              cfg.loss_scale_growth, cfg.loss_scale_backoff, int(cfg.loss_scale_interval), cfg.loss_scale_min, cfg.loss_scale_max,
              _ptr(self.dev_bias), cfg.adam_betas[0], cfg.adam_betas[1])

    def _fused_adam_plan(self):
        """(lo, hi) of the flat parameter buffer whose Adam step the main grid's backward can take over
        (EngineConfig.fuse_grid_adam), or None."""
        cfg = self.cfg
        modes = cfg.grid_bwd_mode if isinstance(cfg.grid_bwd_mode, (tuple, list)) else (cfg.grid_bwd_mode,) * 3
        flags = bool(cfg.producer_overflow_flags and all(int(m) in (1, 3) for m in modes) and self.world_size == 1
                     and getattr(self, "_reducer", None) is None)
        store = cfg.store_input_gradients
        if store is None:
            store = bool(cfg.optimize_poses or cfg.expect_normals)
        # (a gather-form input gradient behind the parameter backward would read the table after its step)
        if not (cfg.fuse_grid_adam and flags and (store or not (cfg.optimize_poses or cfg.expect_normals))):
            return None
        first, n = C.c_uint64(0), C.c_uint64(0)
        _call("nvo_fused_adam_range", self.base_net.handle, C.byref(first), C.byref(n))
        if n.value == 0:
            return None
        lo = self.segments["field.base"][0] + int(first.value)
        return lo, lo + int(n.value)

    def _set_fused_adam(self, on: bool, from_device_scalars: bool = True) -> None:
        """Arms / disarms the optimiser step inside the main grid's parameter backward (read at launch time: armed
        around the capture of the one-graph step only, so that eager steps keep storing the gradient)."""
        if not on:
            _call("nvo_set_fused_adam", self.base_net.handle, None)
            return
        cfg = self.cfg
        gi = self._GROUP_ORDER.index("fields")
        base = self.segments["field.base"][0]
        a = _lib.FusedAdamArgs(
            params=self.params.data_ptr() + 4 * base, params_half=self.params_half.data_ptr() + 2 * base,
            exp_avg=self.exp_avg.data_ptr() + 4 * base, exp_avg_sq=self.exp_avg_sq.data_ptr() + 4 * base,
            hyper_dev=self.dev_scalars.data_ptr() + 4 * (1 + 3 * gi) if from_device_scalars else None,
            bias_dev=self.dev_bias.data_ptr() + 8 * gi,
            loss_scale_dev=self.dev_loss_scale.data_ptr() if cfg.dynamic_loss_scale else None,
            skip_flag=self.skip_flag.data_ptr() + 4 * gi, lr=self._group_lr("fields"), grad_scale=1.0 / cfg.loss_scale,
            beta1=cfg.adam_betas[0], beta2=cfg.adam_betas[1], eps=cfg.adam_eps)
        _call("nvo_set_fused_adam", self.base_net.handle, C.byref(a))

    _TABLE_ROWS = 1024

    def _scalar_table(self):
        """Device ring of the per-step scalars (row s % 1024 = _scalar_row(s)) + the step counter nvo_opt_commit_table
        advances; see EngineConfig.commit_from_table."""
        if getattr(self, "dev_scalar_table", None) is None:
            self.dev_scalar_table = torch.zeros(self._TABLE_ROWS, 16, dtype=torch.float32, device=self.device)
            self.dev_next_step = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._table_valid = (0, 0, None)  # [lo, hi) steps whose rows are in the ring, stamp of what they depend on
            self._next_step_host = None
        return self.dev_scalar_table

    def _scalar_row(self, s: int):
        """What _commit_and_write puts into dev_scalars for step s (with self.step == s)."""
        vals = [0.0] * 16
        vals[0] = self.anneal_at(s)
        cfg = self.cfg
        for gi, g in enumerate(self._GROUP_ORDER):
            vals[1 + 3 * gi] = {"fields": cfg.lr_fields, "proposal_networks": cfg.lr_proposal}.get(g) if g != "camera_opt" \
                else self.camera_lr(s)
        vals[14], vals[15] = vals[0], float(s)
        return vals

    def _fill_scalar_table(self, s: int) -> None:
        """Makes sure row ``s`` of the ring holds step s's scalars (refilled 1024 steps at a time, in stream order; also
        when a learning rate or a schedule constant was changed from outside)."""
        cfg = self.cfg
        stamp = (cfg.lr_fields, cfg.lr_proposal, cfg.lr_camera, cfg.lr_camera_final, cfg.max_num_iterations,
                 cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope)
        lo, hi, have = self._table_valid
        if have == stamp and lo <= s < hi:
            return
        rows = [None] * self._TABLE_ROWS
        for t in range(s, s + self._TABLE_ROWS):
            rows[t % self._TABLE_ROWS] = self._scalar_row(t)
        self.dev_scalar_table.copy_(torch.tensor(rows, dtype=torch.float32))
        self._table_valid = (s, s + self._TABLE_ROWS, stamp)

    def _launch_commit_table(self, mask: int, scale_mask: int) -> None:
        cfg = self.cfg
        _call("nvo_opt_commit_table", _stream(self.device), len(self._GROUP_ORDER), mask, scale_mask, _ptr(self.dev_applied),
              _ptr(self.skip_flag), _ptr(self.dev_loss_scale) if scale_mask else None,
              _ptr(self.dev_growth_tracker) if scale_mask else None, cfg.loss_scale_growth, cfg.loss_scale_backoff,
              int(cfg.loss_scale_interval), cfg.loss_scale_min, cfg.loss_scale_max, _ptr(self.dev_bias), cfg.adam_betas[0],
              cfg.adam_betas[1], _ptr(self.dev_scalars), _ptr(self.dev_scalar_table), self._TABLE_ROWS, _ptr(self.dev_next_step))

    def _commit_and_write(self, masks, sampling_step: int) -> None:
        """GradScaler.update + step counters of the step that just replayed AND the scalars of step ``sampling_step`` (=
        self.step, already advanced) in one eager launch (nvo_opt_commit_write)."""
        cfg = self.cfg
        mask, scale_mask = masks
        vals = self._scalar_values(self.anneal_at(sampling_step), sampling_step)
        arr = (C.c_float * 16)(*vals)
        _call("nvo_opt_commit_write", _stream(self.device), len(self._GROUP_ORDER), mask, scale_mask, _ptr(self.dev_applied),
              _ptr(self.skip_flag), _ptr(self.dev_loss_scale) if scale_mask else None,
              _ptr(self.dev_growth_tracker) if scale_mask else None, cfg.loss_scale_growth, cfg.loss_scale_backoff,
              int(cfg.loss_scale_interval), cfg.loss_scale_min, cfg.loss_scale_max, _ptr(self.dev_bias), cfg.adam_betas[0], cfg.adam_betas[1],
              _ptr(self.dev_scalars), 16, arr)
        self._scalars_step = sampling_step

    # ------------------------------------------------------------------------------------------
    # hipGraph replay of the step
    # ------------------------------------------------------------------------------------------
    def _write_sampling_scalars(self, step: int) -> None:
        """[proposal-weight anneal, counter of the stateless pixel / jitter sampler] of ``step`` -> device memory (read
        by the sampling prefix of the captured step); the values travel as kernel arguments of a tiny eager launch."""
        arr = (C.c_float * 2)(self.anneal_at(step), float(step))
        _call("nvo_write_floats", _stream(self.device), _ptr(self.dev_sampling), 2, arr)

    def _scalar_values(self, anneal: float, sampling_step: int | None):
        """[anneal | lr, -, - per group | ... | anneal, counter of the sampling step] -- functions of self.step alone"""
        vals = [0.0] * 16
        vals[0] = anneal
        for gi, g in enumerate(self._GROUP_ORDER):
            vals[1 + 3 * gi] = self._group_lr(g)
        if sampling_step is not None:
            vals[14], vals[15] = self.anneal_at(sampling_step), float(sampling_step)
        return vals

    def _write_step_scalars(self, anneal: float, groups, sampling_step: int | None = None) -> None:
        """Learning rates of this step -> device memory (read by the captured optimiser; the Adam bias corrections are
        computed on the device from the applied-step counters).  (Slot 0 mirrors the anneal for inspection; the
        kernels read dev_sampling.)"""
        self._scalars_step = None  # (whatever was written ahead by _commit_and_write is overwritten)
        vals = self._scalar_values(anneal, sampling_step)
        if sampling_step is not None:  # (single GPU: the sampling scalars ride in the same launch)
            arr = (C.c_float * 16)(*vals)
            _call("nvo_write_floats", _stream(self.device), _ptr(self.dev_scalars), 16, arr)
        else:  # (multi GPU: slots 14-15 belong to the sampling prefix that may already run ahead)
            arr = (C.c_float * 14)(*vals[:14])
            _call("nvo_write_floats", _stream(self.device), _ptr(self.dev_scalars), 14, arr)

    def train_step_graphed(self, dataset, all_reduce=None):
        """One full iteration replayed from captured hipGraphs (torch.cuda.CUDAGraph).

        Single GPU: pixel sampling, jitters, ray generation, forward, losses, backward and the optimiser are ONE graph
        launch, which removes the ~45 inter-kernel launch gaps of the eager step (one graph per step variant: with /
        without the proposal-network update, with the value-only proposal losses).

        With ``all_reduce`` (one process per GPU; nerf_vo_amd.parallel.GradientAllReduce) the step is software-pipelined
        across iterations WITHOUT changing any value:

            body_k   : main field forward, losses, every backward, the 2-byte cast of the gradients (+ overflow flags)
            reduce A : proposal networks (+ camera poses) -- small, all-reduce -- then opt_A (their replicated Adam)
            reduce B : fields (24.5 MB as bf16): reduce-scatter, ASYNC on the collective's stream ...
            head_k+1 :   ... while the compute stream already runs the next iteration's sampling prefix up to proposal
                         level 0 (reads proposal-network + pose parameters only)
            opt_B    : Adam of the fields group on this rank's 1/W slice
            gather B : all-gather of the 16-bit working copy of the fields group, beside proposal level 1 of head_k+1.

        (``all_reduce.shard_optimizer`` False: reduce B is an all-reduce and opt_B the replicated Adam, no gather.)
        With RCCL the collectives are CAPTURED with the kernels -- the whole iteration, next prefix included, is ONE
        graph launch per step; with a backend that cannot be captured (gloo in the tests) or NVO_DIST_CAPTURE=0 the
        compute segments are separate graphs around eager collectives."""
        cfg = self.cfg
        R = cfg.num_rays
        step = self.step
        updated = self.proposal_update_due(step)
        groups = ["fields"] + (["proposal_networks"] if updated else []) + (["camera_opt"] if cfg.optimize_poses else [])
        has_depth = dataset.frames_depth is not None
        has_normals = bool(getattr(dataset, "use_normals", False)) and cfg.normal_loss_mult > 0.0
        values = self.proposal_values_due(step, updated)
        sparse = self._sparse_for_step(step) and self.levels[-1] % 16 == 0
        key = (R, updated, has_depth, all_reduce is not None, has_normals, values, sparse)
        self._reducer = all_reduce
        if self._pix_scale is None:
            self._pix_scale = torch.zeros(3, dtype=torch.float32, device=self.device)
            self._pix_scale_host = None
        extent = (dataset.num_active_frames, dataset.frame_height, dataset.frame_width)
        if self._pix_scale_host != extent:  # keyframes were added: refresh the sampler range in place
            self._pix_scale.copy_(torch.tensor(extent, dtype=torch.float32))
            self._pix_scale_host = extent
        entry = self._graphs.get(key)
        if entry is None:
            # Capture EVERY variant of the step right away (with / without the proposal-network update, and the
            # non-update step that still evaluates the proposal loss values): the first ten steps all refresh the
            # proposal networks, so the other graphs would otherwise be captured -- two eager warm-up steps + the
            # capture, a few ms -- in the middle of training (step 10), e.g. inside a timed window.
            mode = cfg.proposal_loss_values
            variants = [(True, False)] + ([(False, False)] if mode != "always" else []) + (
                [(False, True)] if mode != "never" else [])
            # (... and, EngineConfig.sparse_backward "auto", both kinds of every variant: the probe runs a sparse step every
            # 64th step from the start, and the switch to sparse steps comes in the middle of a run)
            kinds = [False, True] if (cfg.sparse_backward == "auto" and self.levels[-1] % 16 == 0) else [sparse]
            for v_upd, v_val in variants:
                for v_sparse in kinds:
                    vkey = (R, v_upd, has_depth, all_reduce is not None, has_normals, v_val, v_sparse)
                    if vkey in self._graphs:
                        continue
                    v_groups = ["fields"] + (["proposal_networks"] if v_upd else []) + (
                        ["camera_opt"] if cfg.optimize_poses else [])
                    self._graphs[vkey] = self._capture_step(dataset, R, v_upd, has_depth, v_groups, all_reduce is not None,
                                                            has_normals, v_val, sparse=v_sparse)
            entry = self._graphs[key]
        if all_reduce is None and not entry.get("pipelined"):
            self._pending_head = None
            table = bool(entry.get("table_commit"))
            if table:
                self._fill_scalar_table(step + 1)  # (the commit node at the end of this replay loads row step + 1)
            if self._scalars_step != step:  # (not written ahead: first step, externally set step, eager work in between)
                self._write_step_scalars(self.anneal_at(step), groups, sampling_step=step)
            if table and self._next_step_host != step + 1:
                self.dev_next_step.fill_(step + 1)
            entry["main"].replay()
            if table:
                self._next_step_host = step + 2
                self._scalars_step = step + 1
        elif all_reduce is None:
            # single GPU, pipelined: this graph ends with [Adam of the fields group || sampling prefix of step + 1]
            stamp = (step, getattr(dataset, "version", 0), extent)
            if self._pending_head != stamp:  # not launched ahead (first step, new keyframes, eager work in between)
                self._write_sampling_scalars(step)
                entry["head"].replay()
            self._write_step_scalars(self.anneal_at(step), groups, sampling_step=step + 1)
            entry["main"].replay()
            self._pending_head = (step + 1, getattr(dataset, "version", 0), extent)
        else:
            stamp = (step, getattr(dataset, "version", 0), extent)
            if self._pending_head != stamp:  # not launched ahead (first step, new keyframes, externally set step)
                self._write_sampling_scalars(step)
                entry["head"].replay()
            self._pending_head = None
            pipeline = bool(cfg.pipeline_sampling_prefix)
            # learning rates of THIS iteration and, when its prefix runs inside this one, the sampling scalars of the
            # NEXT iteration: one tiny launch
            self._write_step_scalars(self.anneal_at(step), groups, sampling_step=step + 1 if pipeline else None)
            entry["run"](pipeline)
            if pipeline:
                self._pending_head = (step + 1, getattr(dataset, "version", 0), extent)
        if sparse:
            self._sparse_probe_after(step, R)  # (reads the loss shards: before anything of the next step clears them)
        if updated:
            self.steps_since_proposal_update = 0
        self.steps_since_proposal_update += 1
        self.step += 1
        if all_reduce is None and entry.get("commit") is not None:
            self._commit_and_write(entry["commit"], self.step)  # (self.step: the NEXT step's learning rates and sampler state)
        return updated

    def _capture_step(self, dataset, R, updated, has_depth, groups, split, has_normals=False, values=False, sparse=False):
        dev = self.device
        cfg = self.cfg
        ws = self._workspace(R, True)
        self._pending_head = None  # (the warm-up steps below overwrite a prefix launched ahead)
        intr = dataset.camera_intrinsics
        c2w_full = dataset.camera_extrinsics
        scale = self._pix_scale
        anneal_ptr = self.dev_sampling.data_ptr()
        # ONE set of pose / drawn-pixel / jitter buffers for all step variants of this ray count: a pipelined graph runs
        # the NEXT step's prefix, and that step may replay another variant
        if not hasattr(self, "_step_buffers"):
            self._step_buffers = {}
        bkey = (R, int(c2w_full.shape[0]))
        if bkey not in self._step_buffers:
            self._step_buffers[bkey] = (torch.empty(c2w_full.shape[0], 3, 4, device=dev),
                                        torch.zeros((R, 3), dtype=torch.int64, device=dev),
                                        torch.zeros((3, R), dtype=torch.float32, device=dev))
        c2w, ray_indices, jit = self._step_buffers[bkey]
        # follows torch.manual_seed; the rank is mixed in so that data-parallel ranks never draw the same rays even when
        # every process was seeded identically (same multiplier as the data manager's rank-offset generator)
        rng_seed = int((torch.initial_seed() + 1000003 * self.rank) & 0xFFFFFFFF)
        step_ptr = C.c_void_p(self.dev_sampling.data_ptr() + 4)
        jits = (jit[0], jit[1], jit[2])

        zero_with_head = False  # set around the capture of the one-graph step (whole())

        def body_head(levels=None):
            """Sampling prefix.  levels: None = all of it; (0,) = rays + proposal level 0; (1,) = proposal level 1."""
            first = levels is None or 0 in levels
            return body_head_fused(levels, first)

        def body_head_fused(levels, first):
            # everything per ray up to the first sampler level in ONE launch (nvo_ray_head)
            stream = _stream(dev)
            if not first:
                return self._forward_head(ws, 1.0, jits, stream, anneal_dev=anneal_ptr, skip_first_level=True, levels=levels)
            images, depths = dataset.frames_color, dataset.frames_depth if has_depth else None
            normals = dataset.world_normals01() if has_normals else None
            corr, poses, stride, ridx = None, c2w_full, 16, ray_indices
            if cfg.optimize_poses:
                # CameraOptimizer.forward for every camera; the pose backward wants [F][3][4] poses + the drawn pixels
                _call("nvo_pose_exp_map", stream, cfg.num_images, self._param_ptr("camera_opt.pose_adjustment", self.params),
                      _ptr(self.corrections), self._pose_mode())
                c2w.copy_(c2w_full[:, :3, :4])
                corr, poses, stride = self.corrections, c2w, 12
                self._pose_inputs = (intr, c2w)
                if "ray_indices" in ws:
                    ridx = ws["ray_indices"]
            ra = _lib.RayHeadArgs(
                R=R, S=self.levels[0], seed=rng_seed, n_jitter=3, step_dev=step_ptr.value, extent_dev=scale.data_ptr(),
                intrinsics=intr.data_ptr(), c2w=poses.data_ptr(), c2w_stride=stride,
                corrections=None if corr is None else corr.data_ptr(), H=images.shape[1], W=images.shape[2],
                images=images.data_ptr(), depths=None if depths is None else depths.data_ptr(),
                normals=None if normals is None else normals.data_ptr(), near_plane=cfg.near_plane,
                far_plane=cfg.far_plane, ray_indices=ridx.data_ptr(), jitter=jit.data_ptr(),
                origins=ws["origins"].data_ptr(), directions=ws["directions"].data_ptr(),
                directions_norm=ws["directions_norm"].data_ptr(), pixel_area=ws["pixel_area"].data_ptr(),
                cam_idx=ws["cam_idx"].data_ptr(), gt_rgb=ws["gt_rgb"].data_ptr(), gt_depth=ws["gt_depth"].data_ptr(),
                gt_normal=ws["gt_normal"].data_ptr(), dirs01=ws["dirs01"].data_ptr(), sh=ws["sh"].data_ptr(),
                sh_bf16=int(self.bf16), sbins=ws["sbins0"].data_ptr(), tbins=ws["tbins0"].data_ptr(), x01=ws["x0"].data_ptr())
            if zero_with_head:
                # one-graph step: the launch that opens the step also clears its accumulate-into buffers
                pose_z = cfg.optimize_poses and "d_sh" in ws
                nz, zp, zs = self._zero_step_buffers(ws, bool(updated), bool(pose_z), None)
                _call("nvo_ray_head_zero", stream, C.byref(ra), nz, zp, zs)
                ws["zeroed_by_head"] = True
            else:
                _call("nvo_ray_head", stream, C.byref(ra))
            ws["dirs01_ready"] = True
            ws["sh_ready"] = True
            self._forward_head(ws, 1.0, jits, stream, anneal_dev=anneal_ptr, skip_first_level=True, levels=levels)

        # ---- the exchange (multi-GPU) -------------------------------------------------------------------------
        red = getattr(self, "_reducer", None) if split else None
        compress = getattr(red, "compress", None)
        half = wire = shard_out = None
        world, rank = (red.world, red.rank) if red is not None else (1, 0)
        sharded = bool(split and getattr(red, "shard_optimizer", False) and compress in ("bf16", "fp16"))
        wire_dt = torch.bfloat16 if compress == "bf16" else torch.float16
        f_lo, f_hi = self.group_ranges["fields"]
        per = (f_hi - f_lo) // max(world, 1)
        kPad = 8  # flag slots behind every chunk of the wire buffer (16 bytes: chunks stay 16-byte aligned)
        if split and compress in ("bf16", "fp16"):
            if not hasattr(self, "_wire_half") or self._wire_half.dtype != wire_dt:
                self._wire_half = torch.zeros(self.n_params, dtype=wire_dt, device=dev)
            half = self._wire_half  # (shared by the step variants: they never run concurrently)
            if sharded:
                assert per * world == f_hi - f_lo and per % 8 == 0, "fields group must split into 16-byte aligned shards"
                if not hasattr(self, "_wire_fields") or self._wire_fields.numel() != world * (per + kPad) or self._wire_fields.dtype != wire_dt:
                    self._wire_fields = torch.zeros(world * (per + kPad), dtype=wire_dt, device=dev)
                    self._shard_out = torch.zeros(per + kPad, dtype=wire_dt, device=dev)
                wire, shard_out = self._wire_fields, self._shard_out
        # multi-GPU: the small groups (proposal networks, camera poses) are reduced and stepped FIRST -- the next
        # iteration's sampling prefix needs them -- the fields group last
        pipe1 = (not split) and bool(cfg.pipeline_single_gpu)  # single GPU: [fields Adam || next prefix]
        groups_a = [g for g in groups if g != "fields"] if (split or pipe1) else []
        groups_b = ["fields"] if (split or pipe1) else list(groups)
        fields_slot = self._GROUP_ORDER.index("fields")
        fields_flag = C.c_void_p(self.skip_flag.data_ptr() + 4 * fields_slot)

        def body_rest():
            self.forward_backward(ws, jits, has_depth=has_depth, update_proposals=updated, anneal=1.0,
                                  anneal_dev=anneal_ptr, has_normals=has_normals, skip_head=True, proposal_values=values,
                                  sparse=sparse)
            if half is None:
                return
            cast = "nvo_cast_bf16" if half.dtype == torch.bfloat16 else "nvo_cast_half"
            for g in (groups_a if sharded else groups):  # 2-byte copy of the ranges the all-reduce will exchange
                lo, hi = self.group_ranges[g]
                _call(cast, _stream(dev), hi - lo, C.c_void_p(self.grads.data_ptr() + 4 * lo),
                      C.c_void_p(half.data_ptr() + 2 * lo))
            if sharded:  # fields: W chunks + flag slots; the cast raises this rank's overflow flag on the way
                _call("nvo_cast_shards", _stream(dev), f_hi - f_lo, world, kPad, C.c_void_p(self.grads.data_ptr() + 4 * f_lo),
                      _ptr(wire), 2 if wire_dt == torch.bfloat16 else 1, fields_flag)

        def body_opt(gs):
            # every group owns ONE flag word (its slot) and the step's zero launch cleared them all, so neither the
            # single optimiser call of the one-graph step nor the two calls of the split step reset anything
            if sharded and gs == ["fields"]:
                # the reduced chunk carries the number of ranks whose gradient overflowed in its first flag slot
                _call("nvo_flag_from_wire", _stream(dev), C.c_void_p(shard_out.data_ptr() + 2 * per), fields_flag)
                # (the kernels index the gradient by flat element offset: rebase the pointer of this rank's chunk)
                class _Rebased:  # noqa: N801 - minimal tensor stand-in for optimizer_step
                    dtype = shard_out.dtype

                    @staticmethod
                    def data_ptr():
                        return shard_out.data_ptr() - 2 * (f_lo + rank * per)
                self.optimizer_step(gs, from_device_scalars=True, grads_half=_Rebased, flags_cleared=True, step_groups=groups,
                                    shard=(rank, world), check=False)
                return
            self.optimizer_step(gs, from_device_scalars=True, grads_half=half, flags_cleared=True, step_groups=groups)

        def reduce_a():
            segs = [(lo, hi - lo) for lo, hi in (self.group_ranges[g] for g in groups_a)]
            if half is not None:
                red.reduce_half(self.grads, half, segs, already_cast=True)
            else:
                red(self.grads, segments=segs)

        def reduce_b_start():
            if sharded:
                return red.reduce_scatter(shard_out, wire, async_op=True)
            if half is not None:
                return red.reduce_half(self.grads, half, [(f_lo, f_hi - f_lo)], already_cast=True, async_op=True)
            return red(self.grads, segments=[(f_lo, f_hi - f_lo)], async_op=True)

        def gather_b_start():
            # every rank stepped its slice of the fp32 master and wrote the 16-bit working copy of that slice
            full = self.params_half[f_lo:f_hi]
            return red.all_gather(full, full[rank * per:(rank + 1) * per], async_op=True)

        def program(pipeline: bool, seg):
            """The iteration behind the sampling prefix; seg(name, fn) runs a COMPUTE segment (captured on its own in
            the eager-collective mode, inline when the whole program is captured)."""
            seg("body", body_rest)
            if groups_a:
                reduce_a()
                seg("opt_a", lambda: body_opt(groups_a))
            pending = reduce_b_start()
            if pipeline:
                seg("head0", lambda: body_head((0,)))
            red.wait(pending)
            seg("opt_b", lambda: body_opt(groups_b))
            pending = gather_b_start() if sharded else []
            if pipeline:
                seg("head1", lambda: body_head((1,)))
            red.wait(pending)

        # warm-up on a side stream (allocations, lazy module state), then capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        saved = (self.params.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.params_half.clone(),
                 self.opt_state.clone())
        # the warm-up runs on real step scalars (learning rates, anneal, sampler counter): all-zero scalars used to
        # put lr / bias1 = 0 / 0 = NaN into every parameter of the first warm-up step
        self._write_step_scalars(self.anneal_at(self.step), groups, sampling_step=self.step)
        with torch.cuda.stream(side):
            for _ in range(2):
                body_head()
                if split:
                    program(False, lambda name, fn: fn())
                else:
                    body_rest()
                    if groups_a:
                        body_opt(groups_a)
                    body_opt(groups_b)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for dst, src in zip((self.params, self.exp_avg, self.exp_avg_sq, self.params_half, self.opt_state), saved):
            dst.copy_(src)  # the warm-up steps must not count as training (nor as applied optimiser steps)
        # the graphs address these buffers by pointer: they must outlive this call (a freed block would be handed
        # to the next small allocation and every replay would scribble over it)
        drawn = ws["ray_indices"] if (cfg.optimize_poses and "ray_indices" in ws) else ray_indices
        entry = {"half": half, "buffers": (c2w, drawn, jit, scale, ray_indices), "ws": ws, "sharded": sharded,
                 "captured_collectives": False}

        def capture(fn, **kw):
            if split:
                # (a process group is up: its watchdog thread polls events, which a GLOBAL-mode capture on this thread
                # forbids -- "operation not permitted when stream is capturing", seen once the sparse / dense kinds doubled
                # the number of captures; thread-local mode confines the restriction to this thread)
                kw.setdefault("capture_error_mode", "thread_local")
            g = torch.cuda.CUDAGraph()
            with capture_graph(g, **kw):
                fn()
            return g

        def capture_one_graph():
            if cfg.commit_from_table:
                def whole_and_commit():
                    self._defer_commit = []
                    try:
                        whole()
                        mask = scale_mask = 0
                        for m_, s_ in self._defer_commit:  # (disjoint group bits)
                            mask |= m_
                            scale_mask |= s_
                        assert self._defer_commit, "the one-graph step runs at least one optimiser launch"
                    finally:
                        self._defer_commit = None
                    self._launch_commit_table(mask, scale_mask)

                self._scalar_table()  # (allocated outside the capture)
                entry["main"] = capture(whole_and_commit)
                entry["table_commit"] = True
                return entry
            if not cfg.commit_behind_replay:
                entry["main"] = capture(whole)
                return entry
            self._defer_commit = []
            try:
                entry["main"] = capture(whole)
                # (one optimiser launch, or two when the fields group is stepped early: ONE commit for the step)
                mask = scale_mask = 0
                for m_, s_ in self._defer_commit:  # (disjoint group bits)
                    mask |= m_
                    scale_mask |= s_
                entry["commit"] = (mask, scale_mask) if self._defer_commit else None
            finally:
                self._defer_commit = None
            assert entry["commit"] is not None, "the one-graph step runs at least one optimiser launch"
            return entry
        if not split and not pipe1:
            def whole():
                body_head()
                body_rest()
                body_opt(groups_b)
            zero_with_head = True  # (the launch that opens the step also clears its accumulate-into buffers)
            self._fused_adam_range = self._fused_adam_plan()
            entry["fused_adam"] = self._fused_adam_range
            if self._fused_adam_range is not None:
                self._set_fused_adam(True)
            try:
                return capture_one_graph()
            finally:
                if self._fused_adam_range is not None:
                    self._set_fused_adam(False)
                self._fused_adam_range = None

        if pipe1:
            opt_stream = torch.cuda.Stream(device=dev)

            def whole_pipelined():
                body_rest()
                if groups_a:  # proposal networks / poses first: the next prefix reads them
                    body_opt(groups_a)
                cur = torch.cuda.current_stream(dev)
                opt_stream.wait_stream(cur)
                with torch.cuda.stream(opt_stream):
                    body_opt(groups_b)
                body_head()
                cur.wait_stream(opt_stream)
            entry["head"] = capture(body_head)
            entry["main"] = capture(whole_pipelined)
            entry["pipelined"] = True
            entry["streams"] = (opt_stream,)
            return entry

        entry["head"] = capture(body_head)
        import os

        want = os.environ.get("NVO_DIST_CAPTURE", "auto")
        backend = red.dist.get_backend(red.group) if red.dist.is_initialized() else "none"
        if want != "0" and (backend == "nccl" or want == "1"):
            # RCCL collectives issued under capture become nodes of the graph (the process group's stream joins the
            # capture through its event dependencies): the whole iteration is then ONE graph launch.  thread_local: the
            # process group's watchdog thread polls events with calls a global-mode capture would be invalidated by.
            try:
                def whole_program(p):
                    # the fused ray head of the graph that ran before (entry["head"] or the previous iteration's
                    # prefix) already produced dirs01 / SH: without the flags every program captured after the first
                    # would re-record nvo_dirs01 + nvo_sh_encode_t (two redundant dependent launches per step)
                    ws["dirs01_ready"] = ws["sh_ready"] = True
                    program(p, lambda name, fn: fn())

                whole_g = {p: capture(lambda p=p: whole_program(p), capture_error_mode="thread_local")
                           for p in (False, True)}
                entry["captured_collectives"] = True
                entry["run"] = lambda pipeline: whole_g[bool(pipeline)].replay()
                entry["whole"] = whole_g
                self._selfcheck_captured_exchange(whole_g[False], saved, red)
                return entry
            except Exception as exc:  # noqa: BLE001 - fall back to eager collectives between captured segments
                import sys

                sys.stderr.write(f"[nerf_vo_amd] collectives could not be captured ({type(exc).__name__}: {exc}); "
                                 "using eager collectives between captured compute segments\n")
                torch.cuda.synchronize(dev)
                for dst, src in zip((self.params, self.exp_avg, self.exp_avg_sq, self.params_half, self.opt_state), saved):
                    dst.copy_(src)
        segs = {}

        def seg_capture(name, fn):
            segs[name] = capture(fn)

        # capture every compute segment once (the collectives in between run eagerly, here and at replay)
        ws["dirs01_ready"] = ws["sh_ready"] = True  # (as whole_program above: the head graph already produced dirs01 / SH)
        program(True, lambda name, fn: (seg_capture(name, fn), segs[name].replay()))
        torch.cuda.synchronize(dev)
        for dst, src in zip((self.params, self.exp_avg, self.exp_avg_sq, self.params_half, self.opt_state), saved):
            dst.copy_(src)
        entry["segments"] = segs
        entry["run"] = lambda pipeline: program(bool(pipeline), lambda name, fn: segs[name].replay())
        return entry

    def _selfcheck_captured_exchange(self, graph, saved, red) -> None:
        """Start-up self-check of a step graph that holds CAPTURED RCCL collectives (first capture per process): one replay
        under a watchdog, then cross-rank agreement of what the exchange produced.  A hang (a captured collective that
        never completes with real peers) or a disagreement ends the process with a NON-ZERO exit code and a message that
        names the way out -- NVO_DIST_CAPTURE=0: eager collectives between captured compute segments, the same program.
        Never a re-exec or an in-process relaunch: this process has touched the GPU.  NVO_DIST_SELFCHECK=0 skips it,
        NVO_DIST_SELFCHECK_TIMEOUT (seconds, default 120) bounds the wait."""
        import os
        import sys
        import time

        if getattr(self, "_exchange_checked", False) or os.environ.get("NVO_DIST_SELFCHECK", "1") == "0":
            return
        self._exchange_checked = True
        timeout = float(os.environ.get("NVO_DIST_SELFCHECK_TIMEOUT", "120"))
        dev = self.device

        def die(why: str):
            sys.stderr.write(f"[nerf_vo_amd] FATAL (rank {red.rank} of {red.world}): {why}.  The step graph with captured RCCL "
                             "collectives failed its start-up self-check; restart with NVO_DIST_CAPTURE=0 (eager collectives "
                             "between captured compute segments -- the same program, +0.04 ms per step).\n")
            sys.stderr.flush()
            os._exit(3)

        done = torch.cuda.Event()
        graph.replay()
        done.record(torch.cuda.current_stream(dev))
        t0 = time.time()
        while not done.query():
            if time.time() - t0 > timeout:
                die(f"one replay did not complete within {timeout:.0f} s")
            time.sleep(0.002)
        # every rank must now hold the same 16-bit working copy (what the kernels read) and finite fp32 state
        chk = self.params_half.view(torch.int16).to(torch.int64).sum()
        ok = (torch.isfinite(self.params).all() & torch.isfinite(self.exp_avg).all()).to(torch.int64)
        t = torch.stack([chk, -chk, -ok])
        red.dist.all_reduce(t, op=red.dist.ReduceOp.MAX, group=red.group)  # (eager, outside any graph)
        hi, neg_lo, neg_ok = (int(v) for v in t.tolist())
        if hi != -neg_lo:
            die("the ranks' working copies differ after one replayed step")
        if neg_ok != -1:
            die("non-finite parameters or Adam moments after one replayed step")
        torch.cuda.synchronize(dev)
        for dst, src in zip((self.params, self.exp_avg, self.exp_avg_sq, self.params_half, self.opt_state), saved):
            dst.copy_(src)  # (the check's step does not count as training)

    def train_step(self, ray_indices, intrinsics, c2w, images, depths, jitters=None, all_reduce=None, normals=None):
        """One full iteration.  ``all_reduce``: optional callable(flat_grad_tensor) for multi-GPU."""
        R = ray_indices.shape[0]
        ws = self._workspace(R, True)
        self._reducer = all_reduce
        if jitters is None:
            jitters = tuple(torch.rand(R, device=self.device) for _ in range(3))
        self.load_rays(ws, ray_indices, intrinsics, c2w, images, depths, normals=normals)
        # (single GPU: the main grid's backward takes the Adam step of its hashed levels, as in the captured step --
        # forward_backward() on its own keeps storing the whole gradient)
        fused = self._fused_adam_plan() if all_reduce is None else None
        if fused is not None:
            self._set_fused_adam(True, from_device_scalars=False)  # (eager: the learning rate travels as an argument)
        try:
            updated = self.forward_backward(ws, jitters, has_depth=depths is not None, has_normals=normals is not None)
        finally:
            if fused is not None:
                self._set_fused_adam(False)
        groups = ["fields"] + (["proposal_networks"] if updated else []) + ["camera_opt"]
        if fused is not None:
            self._fused_adam_range = fused
            try:
                self.optimizer_step(groups)
            finally:
                self._fused_adam_range = None
        elif all_reduce is not None:
            # one exchange per iteration: the gradient ranges that are non-zero on this step
            active = [g for g in groups if g != "camera_opt" or self.cfg.optimize_poses]
            reduced_half = all_reduce(self.grads, segments=[(lo, hi - lo) for lo, hi in (self.group_ranges[g] for g in active)],
                                      keep_half=True)
            self.optimizer_step(groups, grads_half=reduced_half)
        else:
            self.optimizer_step(groups)
        if updated:
            self.steps_since_proposal_update = 0
        self.steps_since_proposal_update += 1
        self.step += 1
        return updated

    # ------------------------------------------------------------------------------------------
    # inference: a whole image as ONE captured graph
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def render_image(self, origins, directions, directions_norm, normals: bool = False, chunk: int = 1 << 15,
                     use_graph: bool = True) -> dict:
        """Eval forward of a full-image ray bundle ([N,3] origins / directions, [N] norms): every chunk of
        ``eval_num_rays_per_chunk`` rays (nerfstudio default 1 << 15; /root/reference/evaluation/nerf_renderer.py:157 goes
        through model.get_outputs_for_camera_ray_bundle, which chunks) is the same kernel sequence as ``render_rays`` --
        captured ONCE per (ray count, chunk, normals) with all chunks in one hipGraph: the kernels read the rays from and
        write the outputs to persistent full-image buffers at the chunk's offset (no per-chunk staging copies, no
        per-chunk clones), the mean appearance embedding is taken once per image, and a frame costs one graph launch
        instead of ~20 launches x 25 chunks.  Same values as render_rays chunk by chunk, bit for bit
        (tests/test_engine_gpu.py::test_render_image_graph_matches_eager_chunks)."""
        N = int(origins.shape[0])
        key = (N, int(chunk), bool(normals))
        if not hasattr(self, "_render_graphs"):
            self._render_graphs = {}
        entry = self._render_graphs.get(key)
        if entry is None:
            entry = self._render_graphs[key] = self._capture_render(N, int(chunk), bool(normals))
        entry["origins"].copy_(origins.reshape(N, 3))
        entry["directions"].copy_(directions.reshape(N, 3))
        entry["directions_norm"].copy_(directions_norm.reshape(N))
        if use_graph:
            entry["graph"].replay()
        else:  # (per-kernel profiling: the launchers' event hooks only see eager launches)
            self._render_chunks(entry, entry["ws"], lambda: _stream(self.device))
        out = {"rgb": entry["out_rgb"].clone(), "depth": entry["out_depth"].clone()[:, None],
               "expected_depth": entry["out_expected_depth"].clone()[:, None],
               "accumulation": entry["out_accumulation"].clone()[:, None]}
        if normals:
            out["normals"] = entry["out_normals"].clone()
        return out

    def _render_chunks(self, entry, ws, stream_fn) -> None:
        """The chunk loop of render_image on the current stream (eagerly for the warm-up, under capture for the graph)."""
        N, chunk, normals = entry["N"], entry["chunk"], entry["normals"]
        entry["mean_emb"].copy_(self.mean_appearance_embedding().to(self.act_dtype))  # once per image
        for lo in range(0, N, chunk):
            R = min(chunk, N - lo)
            view = dict(ws)  # same scratch for every chunk; rays / outputs at the chunk's offset of the image buffers
            view["R"] = R
            for k in ("origins", "directions", "directions_norm", "out_rgb", "out_depth", "out_expected_depth",
                      "out_accumulation", "out_normals"):
                view[k] = entry[k][lo:lo + R]
            view["dirs01_ready"] = view["sh_ready"] = False
            stream = stream_fn()
            self._forward(view, False, 1.0, None, None, entry["mean_emb"].data_ptr(), stream)
            if normals:
                self._analytic_normal_grads(view, stream)
            la = self._main_loss_args(view, False, False, normals=normals)
            _call("nvo_main_render_loss", stream, C.byref(la))
        entry["out_rgb"].clamp_(0.0, 1.0)

    def _capture_render(self, N: int, chunk: int, normals: bool) -> dict:
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        ws = self._workspace(min(chunk, N), False)
        if normals and "dsigma_dx" not in ws:  # (lazily allocated scratch of the analytic-normal pass: before the views)
            ws["R"] = ws["R_cap"]
            self._analytic_normal_grads(ws, _stream(dev))
        ws["pinned"] = True
        entry = {"N": N, "chunk": chunk, "normals": normals, "ws": ws,  # (ws: kept alive -- the graph addresses it)
                 "origins": torch.zeros(N, 3, **f32), "directions": torch.zeros(N, 3, **f32),
                 "directions_norm": torch.ones(N, **f32), "out_rgb": torch.zeros(N, 3, **f32),
                 "out_depth": torch.zeros(N, **f32), "out_expected_depth": torch.zeros(N, **f32),
                 "out_accumulation": torch.zeros(N, **f32), "out_normals": torch.zeros(N, 3, **f32),
                 "mean_emb": torch.zeros(1, self.cfg.appearance_embed_dim, dtype=self.act_dtype, device=dev)}
        entry["directions"][:, 2] = -1.0  # (finite rays for the warm-up pass)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up: lazy module state, native scratch growth -- none of it may happen under capture
            self._render_chunks(entry, ws, lambda: _stream(dev))
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with capture_graph(g):
            self._render_chunks(entry, ws, lambda: _stream(dev))
        entry["graph"] = g
        return entry

    def mean_appearance_embedding(self) -> torch.Tensor:
        """[1, 32] fp32 mean of the appearance embedding (nerfacto use_average_appearance_embedding), decoded from the
        16-bit WORKING COPY: with the sharded optimiser only the rank that owns the slice holds current fp32 master
        values of it, while the working copy is all-gathered every step and complete everywhere."""
        o, n, _ = self.segments["field.embedding"]
        raw = self.params_half[o:o + n]
        vals = raw.view(torch.bfloat16).float() if self.bf16 else raw.float()
        return vals.view(self.cfg.num_images, -1).mean(dim=0, keepdim=True)

    def layout(self) -> list:
        """Segment table of the flat parameter buffer: [(name, offset, size, group)] -- what a checkpoint is keyed by."""
        return [(n, int(o), int(sz), g) for n, (o, sz, g) in self.segments.items()]

    def loss_dict(self, totals: torch.Tensor | None = None) -> dict:
        """Loss terms of the last step as Python floats (ONE device sync).  ``totals``: a snapshot taken earlier
        with loss_totals()."""
        vals = (self.losses.sum(dim=0) if totals is None else totals).tolist()
        d = {"rgb_loss": vals[0], "distortion_loss": vals[1], "depth_loss": vals[2] + vals[4],
             "interlevel_loss": vals[3]}
        if self.cfg.optimize_poses:
            d["camera_opt_regularizer"] = vals[5]
        if vals[6] != 0.0:
            d["normal_loss"] = vals[6]
        return d

    def loss_totals(self) -> torch.Tensor:
        """[8] device tensor: the sharded loss accumulators of the last step summed (enqueued, no host sync)."""
        return self.losses.sum(dim=0)

    # ------------------------------------------------------------------------------------------
    # inference
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def render_rays(self, origins, directions, directions_norm, mean_embedding_half=None, normals: bool = False):
        """Eval forward for R rays (R*48 multiple of 16): rgb [R,3] clamped to [0,1], median depth,
        expected depth, accumulation.  Uses un-jittered bins and the mean appearance embedding
        (nerfacto use_average_appearance_embedding=True)."""
        R = origins.shape[0]
        ws = self._workspace(R, False)
        ws["origins"][:R].copy_(origins)
        ws["directions"][:R].copy_(directions)
        ws["directions_norm"][:R].copy_(directions_norm.reshape(-1))
        stream = _stream(self.device)
        if mean_embedding_half is None:
            mean_embedding_half = self.mean_appearance_embedding()
        if mean_embedding_half.dtype != self.act_dtype:  # (callers may hand in fp32 / fp16: the colour head reads
            mean_embedding_half = mean_embedding_half.float().to(self.act_dtype)  # its operand format)
        mean_embedding_half = mean_embedding_half.contiguous()
        self._forward(ws, False, 1.0, None, None, mean_embedding_half.data_ptr(), stream)
        if normals:
            self._analytic_normal_grads(ws, stream)
        la = self._main_loss_args(ws, False, False, normals=normals)
        _call("nvo_main_render_loss", stream, C.byref(la))
        out = {"rgb": ws["out_rgb"][:R].clamp(0.0, 1.0), "depth": ws["out_depth"][:R].clone()[:, None],
               "expected_depth": ws["out_expected_depth"][:R].clone()[:, None],
               "accumulation": ws["out_accumulation"][:R].clone()[:, None]}
        if normals:
            out["normals"] = ws["out_normals"][:R].clone()
        return out
