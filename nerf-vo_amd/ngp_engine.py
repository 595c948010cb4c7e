"""Occupancy-grid ("instant-ngp") training / rendering engine -- the second mapping back-end of the
reference (`mapping_module: 'instant-ngp'`, /root/reference/nerf_vo/mapping/instant_ngp.py:19-117,
which drives NVlabs' testbed through pyngp).

One step = what ``Testbed::frame()`` -> ``train(batch)`` does [UPSTREAM]:
    rays -> DDA march through the cascaded Morton bitfield (packed samples, deterministic offsets)
    -> hash grid + density MLP -> SH + rgb MLP -> front-to-back compositing + L2 rgb / depth loss
    (+ per-sample gradients) -> rgb / density MLP + grid backward -> Adam;
    every ``density_update_every`` steps the density grid is re-estimated (EMA), thresholded into the
    bitfield and max-pooled up the cascades.
Network shapes follow instant-ngp's configs/nerf/base.json: HashGrid(L16, F2, T2^19, base 16),
density MLP 32->64->16, rgb MLP [16 | SH16]->64->64->3, exponential density, logistic rgb.

Frame convention: the engine works in instant-ngp's normalised frame (cascade 0 = [0,1]^3, scene
aabb = [0.5 - aabb_scale/2, 0.5 + aabb_scale/2]^3); poses are camera-to-world in that frame, OpenGL
axes (the mapper mirror converts).  All arithmetic is HIP kernels behind include/nerfvo_hip.h.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import time
from dataclasses import dataclass, field

import torch

from . import _lib
from .engine import GridConfig, _call, capture_graph
from .tinycudann.modules import _create, _ptr, _stream

CELLS = 128 ** 3


@dataclass
class NgpConfig:
    num_images: int = 192
    num_rays: int = 4096
    capacity: int = 1 << 18               # packed sample slots per step (instant-ngp's target batch)
    # packed sample slots of an INFERENCE launch (its own workspace): every sample the march finds is shaded, ~110 per ray
    # in a trained room, so the slots decide how many rays a launch takes (0 = the training capacity).  2^23 slots (~300 B
    # of workspace each: 2.5 GB, plus 8 KB of march scratch per ray of the bundle) hold the first round of 131 072 rays --
    # sized for 288 GB of HBM, and for the ray-per-lane march large launches take (pyngp._MAX_BUNDLE_RAYS)
    render_capacity: int = 1 << 23
    # Inference stops a ray where its transmittance has fallen below this (instant-ngp's render_min_transmittance; the
    # reference sets 1e-4, evaluation/nerf_renderer.py:154): render_rays shades the first `render_first_round` samples of
    # every ray, then only the rays still alive take up their march where it stopped (nvo_occ_march_resume).  0 = one pass
    # over every sample up to the scene box.
    render_min_transmittance: float = 1e-4
    # training: samples behind the point where a ray's transmittance falls below this get exactly zero gradients (upstream
    # stops the ray there, EPSILON = 1e-4 in compute_loss_kernel_train_nerf, and trains on the samples in front); the
    # backwards skip zero-gradient samples, so late in training most of a batch costs the backward nothing.  0 = off.
    train_min_transmittance: float = 1e-4
    # Training batch = the samples IN FRONT of that point [UPSTREAM Testbed::train_nerf_step: generate_training_samples_nerf
    # marches up to 16 x the target batch, the network is evaluated on everything marched, compute_loss_kernel_train_nerf
    # stops each ray at T < 1e-4 and compacts the samples in front into the 2^18-sample batch that is trained on; the rays
    # per batch adapt to the samples AFTER compaction].  Here: march (ray-major runs) -> pack everything found into
    # `march_capacity` slots -> density network alone, on the slots in use (the count stays on the device) ->
    # nvo_ngp_count_alive -> pack the first kept[r] samples of every run into `capacity` slots -> the training pass.
    # False: every marched sample is trained on, the ones behind the threshold with zero gradients (rounds 1-4).
    compact_training: bool = True
    march_capacity: int = 1 << 22         # upstream: max_samples = target_batch_size * 16
    # The pass that finds where the rays end runs in ROUNDS: the first train_rounds[0] samples of every ray, then the next
    # train_rounds[1] of the rays that are neither cut nor out of the scene box yet, ..., at last the rest of the march of
    # those still going (late in training a ray keeps ~20 of the ~230 samples its march finds: the density network sees a
    # sixth of them, and most rays never march past their first 32 samples; measured at step 5000 of the bench
    # scene: one round 1.34 ms/step, (32,) 0.95, (24, 24) 0.98, (16, 16, 32) 1.05 -- a round costs ~40 us whatever it holds).  Same kept counts as one pass up to
    # transmittances within fp32 rounding of the threshold; everything stays on the device.  () = one round.
    train_rounds: tuple = (32,)
    render_first_round: int = 48
    aabb_scale: int = 4                   # /root/reference/nerf_vo/mapping/instant_ngp.py:41
    cone_angle: float = 1.0 / 256.0       # instant-ngp: 0 for aabb_scale <= 1, else 1/256
    near_distance: float = 0.1
    desired_resolution: int = 2048
    density_update_every: int = 16
    # Past the first `density_warmup_steps` steps a refresh no longer visits every cell: cells / 4 per cascade drawn
    # uniformly + as many among the cells the grid already holds above the occupancy threshold (half the network
    # evaluations of a full sweep) [UPSTREAM Testbed::training_prep_nerf: `m_training_step < 256` -> all cells, else
    # n_cells / 4 * n_cascades uniform + the same number non-uniform; SURVEY.md section 2.4 K16]
    density_warmup_steps: int = 256
    # cells no training camera sees are excluded from training (grid value -1: never occupied, never refreshed), at step 0
    # and whenever images were added [UPSTREAM mark_untrained_density_grid; SURVEY.md section 2.4 K16]
    mark_untrained: bool = True
    # upstream keeps a cell iff one of its CORNERS projects inside an image, which also drops cells a frustum merely clips
    # (on the 120 x 68 test images the border rows of a view no other camera shares rendered black: 17 -> 14.9 dB); the
    # image is therefore grown by this many projected cell diagonals -- a superset of upstream's trainable cells (0 =
    # upstream's rule)
    mark_untrained_margin: float = 1.0
    density_decay: float = 0.95
    occupancy_threshold: float = 0.01
    rgb_loss_mult: float = 1.0
    depth_loss_mult: float = 1.0          # NeRF-SLAM fork's depth term, LossType.L2 (instant_ngp.py:48)
    loss_scale: float = 128.0
    lr: float = 1e-2
    adam_betas: tuple = (0.9, 0.99)
    adam_eps: float = 1e-15
    l2_reg: float = 1e-6                  # on MLP weights only
    random_background: bool = False
    # training.optimize_extrinsics = True (/root/reference/nerf_vo/mapping/instant_ngp.py:47): per-image pose
    # offsets (rotation vector + translation, applied like nerfstudio's SO3xR3 camera optimiser) trained from
    # the position gradients of the packed samples.  Learning rate / L2 pull toward zero follow instant-ngp's
    # defaults (extrinsic_learning_rate 1e-3, extrinsic_l2_reg 1e-4) [UPSTREAM, unpinned].
    optimize_extrinsics: bool = True
    extrinsic_lr: float = 1e-3
    extrinsic_l2_reg: float = 1e-4
    # The camera optimiser does not step with the network [UPSTREAM Testbed::train_nerf, unpinned]: the per-camera gradient
    # is ACCUMULATED over n_steps_between_cam_updates = 16 training steps, then scaled by n_images / 16 (the mean over the
    # window of a camera's share of the batch), the L2 pull is added and Adam -- with its own step count -- moves the
    # offsets with a learning rate extrinsic_lr * 0.33^(camera step / 128), not below lr / 1000.  (Stepping the offsets with
    # every batch at a constant 1e-3 lets Adam's normalised steps jitter exact poses by ~2e-3: 33 -> 27 dB on the bench
    # scene, EXPERIMENTS 8.17.)  1 = every step (the former behaviour, with the constant rate).
    extrinsic_update_every: int = 16
    extrinsic_lr_decay: float = 0.33
    extrinsic_lr_decay_steps: int = 128
    # "optimizer": {"otype": "Ema", "decay": 0.95, "nested": Adam} of instant-ngp's configs/nerf/base.json (the file the
    # reference loads, instant_ngp.py:45) [UPSTREAM tcnn EmaOptimizer]: inference reads the debiased moving average of
    # the weights, training the raw ones.  0 switches it off.
    ema_decay: float = 0.95
    # single GPU: the grid backward takes the Adam step + weight average of its streamed hashed levels itself
    # (nvo_set_fused_adam; bit-identical to the separate launches)
    fuse_grid_adam: bool = True
    # the weight average and the step's commit ride in the Adam launch (nvo_adam_step_groups_tail; bit-identical to the
    # four launches it replaces: two nvo_ema_update_dev, k_ema_commit, nvo_opt_commit)
    fuse_optimizer_tail: bool = True
    # copies of the two MLPs' weight-gradient buffers the backward's workgroups spread their adds over (0 = off;
    # nvo_fold_replicas sums them once per step)
    dw_replicas: int = 7
    # Testbed::train adapts the rays per batch so that the marched samples meet the target batch (1 << 18):
    # rays <- rays * target / measured, rounded up to the batch granularity (128), every `density_update_every` steps
    # (upstream does it where it reads the loss back, every 16 steps) [UPSTREAM NerfCounters::update_after_training].
    # num_rays is the first batch; rays beyond the packed capacity are dropped for that step, as upstream.
    adaptive_rays: bool = True
    # single GPU: the whole step (ray set-up ... optimiser + step counter) is captured once per ray count and replayed as
    # ONE hipGraph; the caller's ray indices and the march jitter are copied / drawn into fixed buffers in front of the
    # replay, the Adam bias corrections follow a device counter (nvo_opt_commit), the density-grid refresh stays eager.
    graph_step: bool = True
    min_rays: int = 128
    max_rays: int = 1 << 16               # (upstream clamps at 1 << 18; the march scratch is rays x 1024 steps x 8 B)
    seed: int = 1337

    @property
    def n_levels(self) -> int:
        k = 0
        while (1 << k) < self.aabb_scale:
            k += 1
        return k + 1

    @property
    def aabb(self) -> tuple:
        h = 0.5 * self.aabb_scale
        return (0.5 - h, 0.5 + h)

    @property
    def grid(self) -> GridConfig:
        return GridConfig(16, 19, 16, self.desired_resolution * self.aabb_scale)


class NgpEngine:
    def __init__(self, config: NgpConfig, device: torch.device, world_size: int = 1):
        if device.type != "cuda":
            raise RuntimeError("NgpEngine needs an MI355X device; there is no CPU fallback")
        self.cfg = config
        self.device = device
        self.world_size = world_size
        cfg = config
        net_cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
                   "n_hidden_layers": 1}
        self.density_net = _create("nvo_create_network_with_input_encoding", 3, 16,
                                   json.dumps(cfg.grid.tcnn_dict()).encode(), json.dumps(net_cfg).encode())
        # streamed binned scatter for the 2^19 hashed levels (as the nerfacto main field: 650 -> ~250 us at 2^18
        # samples); with extrinsics optimisation the forward stores d(encoded)/d(position) for a streamed input backward
        self.density_net.set_option("grid_bwd_mode", 3)
        self.density_net.set_option("grid_stream_acc_bits", 32)  # packed 2 x 32-bit fixed-point record accumulate
        # the coarse (dense) levels that stay slice-owner: int32 accumulators with the L1-derived scale, run-merging scan,
        # chunk table sized for the packed capacity -- what the nerfacto main grid uses (76 -> ~35 us)
        self.density_net.set_option("grid_acc_bits", 32)
        # every DENSE level stays slice-owner (a hashed 2^19 table is 128 bins of 4096): in an uncontracted scene the samples
        # crowd into a few cells of the finest dense level, its streamed bins become the record pass's long tail
        # (tl_accumulate 110 -> 60 us with that level in the owner launch, which grows 33 -> 55 us; step 0.964 -> 0.936 ms).
        # (The nerfacto main grid is the other way round -- contracted, evenly filled: 116 us streamed vs 126 us owner.)
        self.density_net.set_option("grid_stream_owner_slices", 127)
        self.density_net.set_option("grid_bwd_runs", 1)
        self.density_net.set_option("grid_bwd_batch", int(cfg.capacity))
        self._pig = int(bool(cfg.optimize_extrinsics))  # (the option's value as constructed; launches toggle it around themselves)
        self.density_net.set_option("prepare_input_gradients", self._pig)
        # neither network stores its hidden activations: the backward recomputes them (bit-identical; less traffic both
        # ways, and the recomputing backward is the one that runs in chain / dW roles)
        self.density_net.set_option("recompute_hidden", 1)
        self._bwd_zero_plan = None
        self._leaf_flags = False  # set in forward_backward: the grid backward raises skip_flag itself (single GPU)
        self.n_rgb = 64 * 32 + 64 * 64 + 16 * 64
        self.n_density_mlp = 64 * 32 + 16 * 64
        self.segments = {"density": (0, self.density_net.n_params), "rgb": (self.density_net.n_params, self.n_rgb)}
        self.n_params = self.density_net.n_params + self.n_rgb
        self._dw_rep = None  # (set below, once the gradient buffer exists)
        dev = device
        z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=dev)  # noqa: E731
        self.params, self.grads, self.exp_avg, self.exp_avg_sq = z(self.n_params), z(self.n_params), z(self.n_params), z(self.n_params)
        G = int(cfg.dw_replicas)
        if G > 0:
            buf = z(G * (self.n_density_mlp + self.n_rgb))
            self.density_net.set_option("dw_replicas_ptr", buf.data_ptr())
            self.density_net.set_option("dw_replicas", G)
            rgb_rep = buf.data_ptr() + 4 * G * self.n_density_mlp
            self._dw_rep = {"buf": buf, "G": G, "rgb": rgb_rep,
                            "reps": (C.c_void_p * 2)(buf.data_ptr(), rgb_rep), "n_rep": (C.c_uint32 * 2)(G, G),
                            "n": (C.c_uint64 * 2)(self.n_density_mlp, self.n_rgb),
                            "dst": (C.c_void_p * 2)(self.grads.data_ptr(), self.grads.data_ptr() + 4 * self.density_net.n_params)}
        self.params_half = z(self.n_params, torch.float16)
        # moving average of the weights (inference copy) -- allocated on first use
        self.params_ema = None
        self.params_ema_half = None
        # number of averages applied, ON THE DEVICE: a step the overflow check skipped advances neither the average nor
        # its debias factor (ema_step reads it back)
        self._ema_step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._ema_started = False
        self.rays_per_batch = int(cfg.num_rays)
        self.losses = torch.zeros(64, 8, dtype=torch.float32, device=dev)
        self.skip_flag = z(1, torch.int32)
        self.density_grid = z(cfg.n_levels * CELLS)
        self.bitfield = z(cfg.n_levels * CELLS // 8, torch.uint8)
        self._scratch8 = z(8, torch.uint8)
        self.step = 0
        self.opt_step = 0
        # device side of the optimiser's step count: [0] = learning rate (unused: by value), [1..2] = {1 - beta1^t,
        # sqrt(1 - beta2^t)} of the NEXT applied step (nvo_adam_step's hyper_dev layout; [1..2] alone = nvo_adam_group::bias_dev), advanced by
        # nvo_opt_commit behind the optimiser launches iff the step was not skipped -- nothing of a step depends on a host
        # scalar, so a captured step can be replayed
        self._opt_dev = z(4)
        # the camera optimiser's own device scalars (same layout: [0] = its current learning rate, [1..2] = bias corrections
        # of ITS next applied step), step counter, overflow verdict of the accumulated gradient
        self._cam_dev = z(4)
        self._cam_applied_dev = z(1, torch.int32)
        self._cam_flag = z(1, torch.int32)
        self.cam_step = 0          # camera-optimiser steps attempted (host mirror; drives the learning-rate schedule)
        self._cam_synced = None    # (cam_step, lr) the device scalars were written for
        self._cam_window = 0       # training steps accumulated into d_corrections since the last camera update
        self._applied_dev = z(1, torch.int32)
        self._tail_done = z(1, torch.int32)  # check-in counter of nvo_adam_step_groups_tail (the launch leaves it zero)
        self._dev_synced = None   # the host opt_step the two buffers above were written for
        self._graphs = {}         # captured steps by (ray count, inputs' addresses, ...)
        self._kernels_loaded = False
        self.graph_captures, self.graph_capture_seconds = 0, 0.0  # (diagnostics: tools/ngp_bench.py)
        self.last_render_samples = 0    # samples the march found for the last render_rays bundle (its largest round)
        self.render_shaded_total = 0
        self.params_version = 0         # bumped whenever the weights inference reads may have changed (render caches)
        self.n_training_images = None   # images in use (pyngp: nerf.training.n_images_for_training); None = all slots
        self._marked_images = None      # the image count the untrained cells were last marked for
        self._measured_acc = torch.zeros(1, dtype=torch.int64, device=dev)  # samples per step (after compaction) since the last adaptation
        # rays in use, ON THE DEVICE: a captured step covers the workspace's rows and reads the batch size here, so the
        # adaptive ray batch moves without a new capture
        self._R_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._R_dev_host = -1
        # epoch block of nvo_occ_pack_fused (scan + compact + positions in one launch): zeroed once, the kernels' own from then on
        self._pack_state = torch.zeros(int(_lib.lib().nvo_occ_pack_state_bytes()), dtype=torch.uint8, device=dev)
        self._measured_n = 0
        self._ws = None                      # the workspace used last
        self._wss = {True: None, False: None}  # one for training, one for inference (switching keeps both, and the captured steps)
        # camera offsets [F][6] = (translation, rotation vector) and their optimiser state
        F6 = cfg.num_images * 6
        self.pose_adjustment, self.pose_grads = z(F6), z(F6)
        self.pose_exp_avg, self.pose_exp_avg_sq = z(F6), z(F6)
        self._pose_half = z(F6, torch.float16)  # the fused Adam always writes a half copy
        self.corrections = z(cfg.num_images * 12)
        self.d_corrections = z(cfg.num_images * 12)
        self._pose_inputs = None
        self.init_params(cfg.seed)

    # ---- parameters --------------------------------------------------------------------------
    def init_params(self, seed: int) -> None:
        host = torch.zeros(self.n_params)
        host[: self.density_net.n_params] = self.density_net.initial_params(seed)
        rgb = _create("nvo_create_network", 32, 3, json.dumps(
            {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
             "n_hidden_layers": 2}).encode())
        assert rgb.n_params == self.n_rgb
        host[self.density_net.n_params:] = rgb.initial_params(seed + 1)
        self.set_params(host)

    def set_params(self, flat: torch.Tensor) -> None:
        self.params.copy_(flat.to(self.device, torch.float32))
        _call("nvo_cast_half", _stream(self.device), self.n_params, _ptr(self.params), _ptr(self.params_half))
        self.params_ema = self.params_ema_half = None  # the average restarts from the new weights
        self.ema_step = 0
        self.params_version += 1

    def inference_params_half(self) -> torch.Tensor:
        """The 16-bit weights inference reads: the moving average once it exists (tcnn: the optimiser's
        custom_weights()), the raw working copy otherwise."""
        return self.params_ema_half if (self.params_ema_half is not None and self._ema_started) else self.params_half

    @property
    def ema_step(self) -> int:
        """Averages applied so far (device counter; reading it synchronises)."""
        return int(self._ema_step_dev.item())

    @ema_step.setter
    def ema_step(self, value: int) -> None:
        self._ema_step_dev.fill_(int(value))
        self._ema_started = int(value) > 0
        self.params_version = getattr(self, "params_version", 0) + 1

    def _pp(self, name: str, buf: torch.Tensor):
        o, _ = self.segments[name]
        return C.c_void_p(buf.data_ptr() + o * buf.element_size())

    # ---- scratch -----------------------------------------------------------------------------
    def _workspace(self, R: int, training: bool):
        """Scratch for up to R_cap >= R rays (power of two: the adaptive batch changes R every few steps; the kernels
        take R as an argument and only touch the first R rows)."""
        ws = self._wss[training]
        if ws is not None and R <= ws["R_cap"] and ("dx01" in ws or not (training and self.cfg.optimize_extrinsics)):
            self._ws = ws
            return self._ray_views(ws, R)
        # (training: at least 16384 rows -- the adaptive batch ends between 10 K and 16 K rays on a carved scene, and a
        # workspace that grows clears the captured steps)
        R_req, R = R, max(16384 if training else 4096, 1 << max(0, (R - 1).bit_length()))
        key = (R, training)
        dev = self.device
        cap = int(self.cfg.capacity if (training or not self.cfg.render_capacity) else self.cfg.render_capacity)
        f32 = dict(dtype=torch.float32, device=dev)
        f16 = dict(dtype=torch.float16, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        ws = {"key": key, "R": R_req, "R_cap": R, "training": training, "cap": cap}
        for name, shape in (("origins", (R, 3)), ("directions", (R, 3)), ("directions_norm", (R,)), ("pixel_area", (R,)),
                            ("gt_rgb", (R, 3)), ("gt_depth", (R,)), ("gt_depth_cov", (R,)), ("dirs01", (R, 3)), ("out_rgb", (R, 3)),
                            ("out_depth", (R,)), ("out_accumulation", (R,)), ("t", (cap,)), ("dt", (cap,)),
                            ("x01", (cap, 3)), ("d_density_pre", (cap,))):
            ws[name] = torch.zeros(*shape, **f32)
        if not training:
            for name in ("t_next", "t_resume", "carry"):
                ws[name] = torch.zeros(R, **f32)
        ws["cam_idx"] = torch.zeros(R, **i32)
        ws["counts"] = torch.zeros(R, **i32)
        ws["offsets"] = torch.zeros(R + 1, **i32)
        ws["ray_idx"] = torch.full((cap,), -1, **i32)
        ws["sh"] = torch.zeros(R, 16, **f16)
        ws["density_out"] = torch.zeros(cap, 16, **f16)
        ws["rgb_out"] = torch.zeros(cap, 16, **f16)
        ws["ctx"] = torch.empty(self.density_net.ctx_bytes(cap), dtype=torch.uint8, device=dev)
        # staging area of the march (ray-major runs of accepted samples): owned here, not by the native side
        ws["march_scratch"] = torch.empty(int(_lib.lib().nvo_occ_march_scratch_bytes(R)), dtype=torch.uint8, device=dev)
        ws["totals"] = torch.zeros(2, **i32)   # {samples to pack, slots in use} of the last nvo_occ_pack
        if training and self.cfg.compact_training:
            # the pass that finds where each ray ends: everything the march found, density network only
            cap_m = int(max(self.cfg.march_capacity, cap))
            ws["cap_m"] = cap_m
            for name, shape in (("t_m", (cap_m,)), ("dt_m", (cap_m,)), ("x01_m", (cap_m, 3))):
                ws[name] = torch.zeros(*shape, **f32)
            ws["ray_idx_m"] = torch.full((cap_m,), -1, **i32)
            ws["density_m"] = torch.zeros(cap_m, **f16)  # (compact output: column 0 alone)
            if self._pig:
                self.density_net.set_option("prepare_input_gradients", 0)  # (sizes the ctx without the dy/dx block)
            ws["ctx_m"] = torch.empty(self.density_net.ctx_bytes(cap_m), dtype=torch.uint8, device=dev)
            if self._pig:
                self.density_net.set_option("prepare_input_gradients", 1)
            ws["counts_m"] = torch.zeros(R, **i32)
            ws["offsets_m"] = torch.zeros(R + 1, **i32)
            ws["ray_state"] = torch.zeros(R, **i32)
            ws["kept"] = torch.zeros(R, **i32)
            ws["totals_m"] = torch.zeros(8, 2, **i32)   # per round: {samples found, slots in use}
            for name in ("t_next", "t_resume", "carry"):
                ws[name] = torch.zeros(R, **f32)
        if training:
            ws["d_rgb_out"] = torch.zeros(cap, 16, **f16)
            ws["d_density_out"] = torch.zeros(cap, 16, **f16)
            # fixed homes of the per-step inputs (a captured step reads them by address)
            ws["ray_indices"] = torch.zeros(R, 3, dtype=torch.int64, device=dev)
            ws["jitter"] = torch.zeros(R, **f32)
            if self.cfg.random_background:
                ws["background"] = torch.zeros(R, 3, **f32)
            if self.cfg.optimize_extrinsics:
                ws["dx01"] = torch.zeros(cap, 3, **f32)
                ws["d_origin"] = torch.zeros(R, 3, **f32)
                ws["d_dir"] = torch.zeros(R, 3, **f32)
        ws["_per_ray"] = {k: ws[k] for k in self._PER_RAY if k in ws}
        if training:
            self._graphs.clear()  # (captured steps address the old buffers)
        self._wss[training] = ws
        self._ws = ws
        return self._ray_views(ws, R_req)

    _PER_RAY = ("origins", "directions", "directions_norm", "pixel_area", "gt_rgb", "gt_depth", "gt_depth_cov", "dirs01", "out_rgb",
                "out_depth", "out_accumulation", "cam_idx", "counts", "offsets", "sh", "d_origin", "d_dir", "ray_indices",
                "jitter", "background", "t_next", "t_resume", "carry", "counts_m", "offsets_m", "ray_state", "kept")

    @staticmethod
    def _ray_views(ws, R: int):
        """ws[name] = the first R rows of every per-ray buffer (same storage, same base address)."""
        ws["R"] = R
        for k, full in ws["_per_ray"].items():
            ws[k] = full[:R + 1] if k in ("offsets", "offsets_m") else full[:R]
        return ws

    # ---- density grid ------------------------------------------------------------------------
    @torch.no_grad()
    def update_density_grid(self, jitter: bool = True, all_reduce=None) -> None:
        """update_density_grid_nerf: during the first `density_warmup_steps` steps sample every cell of every cascade
        (jittered), afterwards cells / 4 per cascade uniformly + as many among the occupied cells (nvo_occ_sample_cells);
        evaluate the density network, EMA the optical thickness into the grid (cells without a sample only decay),
        rebuild bitfield + cascade max-pool.
        ``all_reduce`` (multi-GPU): the fresh estimates are MAX-reduced over the ranks before the update, so the
        density grid and bitfield stay identical everywhere (parallel.GradientAllReduce.reduce_max)."""
        # (The density network is evaluated with the RAW training weights, not the EMA copy inference reads: upstream's
        # update_density_grid_nerf calls m_nerf_network->density(..., use_inference_params = false) [UPSTREAM, unpinned --
        # the submodule is not vendored].  Marching therefore follows the weights being trained, shading at inference
        # the averaged ones, exactly as in the testbed.)
        cfg = self.cfg
        stream = _stream(self.device)
        lo, hi = cfg.aabb
        chunk = 1 << 19
        x01 = torch.empty(chunk, 3, device=self.device)
        out = torch.empty(chunk, dtype=torch.float16, device=self.device)  # (compact: the density column alone)
        net = self.density_net
        # density alone, no d(encoded)/d(position): the refresh evaluates 3.1 M cell samples every 16 steps -- as many
        # network evaluations per step as the training batch -- and ran them as TRAINING forwards until round 5 (16-column
        # rows, 192 B of dy/dx per sample nobody read: 164 us per 2^19 samples against 116)
        net.set_option("compact_output", 1)
        if self._pig:
            net.set_option("prepare_input_gradients", 0)
        try:
            ctx = torch.empty(net.ctx_bytes(chunk), dtype=torch.uint8, device=self.device)
            if self.step >= cfg.density_warmup_steps:
                # scattered refresh: a uniform pass (every trained cell qualifies) and a pass over occupied cells
                fresh = torch.zeros(cfg.n_levels * CELLS, device=self.device)
                cells = torch.empty(chunk, dtype=torch.int32, device=self.device)
                n_pass = CELLS // 4 * cfg.n_levels
                for pass_id, thresh in ((0, -0.01), (1, cfg.occupancy_threshold)):
                    for c0 in range(0, n_pass, chunk):
                        m = min(chunk, n_pass - c0)
                        _call("nvo_occ_sample_cells", stream, m, c0, n_pass, self.step, cfg.seed & 0xFFFFFFFF, pass_id,
                              cfg.n_levels, _ptr(self.density_grid), thresh, lo, hi, _ptr(x01), _ptr(cells))
                        # (the network runs on whole chunks: rows past m hold an earlier chunk's points and are not splatted)
                        _call("nvo_fwd", net.handle, stream, chunk, _ptr(x01), self._pp("density", self.params_half),
                              _ptr(out), _ptr(ctx))
                        _call("nvo_ngp_thickness_splat", stream, m, _ptr(out), 1, _ptr(cells), _ptr(fresh))
            else:
                fresh = torch.empty(cfg.n_levels * CELLS, device=self.device)
                pos = torch.empty(CELLS, 3, device=self.device)
                for level in range(cfg.n_levels):
                    jit = torch.rand(CELLS, 3, device=self.device) if jitter else None
                    _call("nvo_occ_cell_positions", stream, level, _ptr(jit), _ptr(pos))
                    for c0 in range(0, CELLS, chunk):
                        x01.copy_(((pos[c0:c0 + chunk] - lo) / (hi - lo)).clamp_(0.0, 1.0))
                        _call("nvo_fwd", net.handle, stream, chunk, _ptr(x01), self._pp("density", self.params_half),
                              _ptr(out), _ptr(ctx))
                        _call("nvo_ngp_thickness", stream, chunk, _ptr(out), 1, level,
                              C.c_void_p(fresh.data_ptr() + 4 * (level * CELLS + c0)))
        finally:
            net.set_option("compact_output", 0)
            if self._pig:
                net.set_option("prepare_input_gradients", 1)
        if all_reduce is not None:
            all_reduce.reduce_max(fresh)
        _call("nvo_occ_update", stream, cfg.n_levels, _ptr(self.density_grid), _ptr(fresh), cfg.density_decay,
              cfg.occupancy_threshold, _ptr(self.bitfield), _ptr(self._scratch8))
        self.params_version += 1  # (the bitfield inference marches through)

    @torch.no_grad()
    def mark_untrained_cells(self, intrinsics, c2w, n_images: int, H: int, W: int) -> None:
        """mark_untrained_density_grid: cells none of the first ``n_images`` cameras sees get -1 (excluded from training and
        from the bitfield), cells that gained a view come back as 0.  Takes effect with the next update_density_grid()."""
        _call("nvo_occ_mark_untrained", _stream(self.device), self.cfg.n_levels, _ptr(self.density_grid), int(n_images),
              _ptr(intrinsics), _ptr(c2w), int(H), int(W), float(self.cfg.mark_untrained_margin))
        self._marked_images = int(n_images)

    # ---- forward / backward ------------------------------------------------------------------
    def load_rays(self, ws, ray_indices, intrinsics, c2w, images, depths, depths_cov=None) -> None:
        """``depths_cov`` [F,H,W,1] (optional): per-pixel variance of the depth targets, what the reference hands to
        update_training_images on every instant-ngp configuration (nerf_vo/mapping/instant_ngp.py:77-86,93-94); the
        depth residual of a ray is weighted by its inverse (nvo_ngp_loss_args::gt_depth_cov)."""
        stream = _stream(self.device)
        R = ws.get("R_launch", ws["R"])
        H, W = images.shape[1], images.shape[2]
        corr = None
        self._pose_inputs = None
        if self.cfg.optimize_extrinsics and "dx01" in ws:
            _call("nvo_pose_exp_map", stream, self.cfg.num_images, _ptr(self.pose_adjustment), _ptr(self.corrections), 1)
            corr = self.corrections
            if ray_indices.data_ptr() != ws["ray_indices"].data_ptr():  # (the graphed step has copied them already)
                ws["ray_indices"].copy_(ray_indices)
            self._pose_inputs = (intrinsics, c2w)
        # rays, targets, direction-encoding input and its SH encoding in ONE launch (they were five)
        _call("nvo_rays_given", stream, R, _ptr(ray_indices), _ptr(intrinsics), _ptr(c2w), _ptr(corr), H, W, _ptr(images),
              _ptr(depths) if depths is not None else None, _ptr(ws["origins"]), _ptr(ws["directions"]),
              _ptr(ws["directions_norm"]), _ptr(ws["pixel_area"]), _ptr(ws["cam_idx"]), _ptr(ws["gt_rgb"]),
              _ptr(ws["gt_depth"]), _ptr(ws["dirs01"]), _ptr(ws["sh"]),
              _ptr(depths_cov) if (depths is not None and depths_cov is not None) else None, _ptr(ws["gt_depth_cov"]),
              self._rdev(ws))
        ws["has_depth_cov"] = depths is not None and depths_cov is not None
        ws["sh_ready"] = True

    def _rdev(self, ws):
        """Device ray count of a launch that covers the workspace's rows (graphed steps), or None."""
        return _ptr(self._R_dev) if ws.get("R_launch") else None

    @staticmethod
    def _pack_group(ws) -> int:
        """Rays a workgroup of nvo_occ_pack_fused owns: 16 for a small batch, 64 for a large one (speed only)."""
        return 64 if int(ws["R"]) >= 6144 else 16

    def _forward(self, ws, training: bool, jitter, stream) -> None:
        if training and self.cfg.compact_training and "cap_m" in ws:
            self._march_compact(ws, jitter, stream)
        else:
            self._march(ws, jitter, stream)
        self._shade(ws, training, stream)

    def _march(self, ws, jitter, stream, t_resume=None, max_new: int = 1024, t_next=None) -> None:
        """Packed samples of the workspace's rays: counts / offsets (offsets[R] = totals[0] = samples found, before rays were
        dropped at the capacity), ray_idx / t / dt.  ``t_resume`` / ``max_new`` / ``t_next``: one round of a march in rounds
        (nvo_occ_march_resume)."""
        cfg = self.cfg
        R, cap = ws.get("R_launch", ws["R"]), ws["cap"]
        rdev = self._rdev(ws)
        ws.pop("ray_state_on", None)
        _call("nvo_fill_i32", stream, cap, _ptr(ws["ray_idx"]), -1)
        _call("nvo_occ_march_runs", stream, R, _ptr(ws["origins"]), _ptr(ws["directions"]), _ptr(self.bitfield),
              cfg.n_levels, cfg.cone_angle, cfg.near_distance, _ptr(jitter), _ptr(ws["counts"]), _ptr(ws["march_scratch"]),
              ws["march_scratch"].numel(), _ptr(t_resume), int(max_new), _ptr(t_next), rdev, 0)
        lo, hi = cfg.aabb
        _call("nvo_occ_pack_fused", stream, R, _ptr(ws["counts"]), cap, _ptr(ws["counts"]), _ptr(ws["offsets"]), _ptr(ws["totals"]),
              _ptr(ws["march_scratch"]), ws["march_scratch"].numel(), _ptr(ws["ray_idx"]), _ptr(ws["t"]), _ptr(ws["dt"]), rdev, 0,
              _ptr(self._pack_state), _ptr(ws["origins"]), _ptr(ws["directions"]), lo, hi, _ptr(ws["x01"]), self._pack_group(ws))
        ws["x01_ready"] = True  # (the pack wrote the network input of every packed sample)

    def _march_compact(self, ws, jitter, stream) -> None:
        """The training batch as upstream builds it (NgpConfig.compact_training): march -> everything found packed into the
        `march_capacity` slots -> density network on the slots in use -> where each ray's transmittance falls below
        train_min_transmittance (kept[r], ray_state[r]) -> the first kept[r] samples of every run packed into the training
        slots.  Leaves counts / offsets / ray_idx / t / dt of the COMPACTED batch (totals[0] = its size before rays were
        dropped at the capacity: what the adaptive ray batch measures) and ray_state for the loss kernel."""
        cfg = self.cfg
        R, cap, cap_m = ws.get("R_launch", ws["R"]), ws["cap"], ws["cap_m"]
        rdev = self._rdev(ws)
        lo, hi = cfg.aabb
        nscr = ws["march_scratch"].numel()
        net = self.density_net
        budgets = [int(b) for b in cfg.train_rounds if int(b) > 0][:7]
        assert sum(budgets) < 1024, "NgpConfig.train_rounds: the rounds in front of the last one must leave it samples"
        budgets.append(1024 - sum(budgets))
        base = 0
        for k, budget in enumerate(budgets):
            last = k + 1 == len(budgets)
            resume = ws["t_resume"] if k else None
            tot_ptr = ws["totals_m"].data_ptr() + 8 * k
            n_live = C.c_void_p(tot_ptr + 4)
            # slots this round can fill at most: its budget per ray
            B = min(cap_m, (R * budget + 4095) // 4096 * 4096)
            # (no clearing of ray_idx_m: nothing of this pass reads the slots no ray owns -- the pack writes position, t and
            # dt of the slots in use, the network stops at their count, the count kernel walks the rays' own ranges)
            _call("nvo_occ_march_runs", stream, R, _ptr(ws["origins"]), _ptr(ws["directions"]), _ptr(self.bitfield),
                  cfg.n_levels, cfg.cone_angle, cfg.near_distance, _ptr(jitter), _ptr(ws["counts_m"]), _ptr(ws["march_scratch"]),
                  nscr, _ptr(resume), budget, None if last else _ptr(ws["t_next"]), rdev, base)
            _call("nvo_occ_pack_fused", stream, R, _ptr(ws["counts_m"]), B, _ptr(ws["counts_m"]), _ptr(ws["offsets_m"]),
                  C.c_void_p(tot_ptr), _ptr(ws["march_scratch"]), nscr, _ptr(ws["ray_idx_m"]), _ptr(ws["t_m"]), _ptr(ws["dt_m"]), rdev, base,
                  _ptr(self._pack_state), _ptr(ws["origins"]), _ptr(ws["directions"]), lo, hi, _ptr(ws["x01_m"]), self._pack_group(ws))
            # density alone (column 0, compact), no d(encoded)/d(position), tiles past the slots in use skipped; the raw
            # weights (the training pass behind this evaluates the same ones)
            net.set_option("n_live_ptr", n_live.value)
            net.set_option("compact_output", 1)
            if self._pig:
                net.set_option("prepare_input_gradients", 0)
            try:
                _call("nvo_fwd", net.handle, stream, B, _ptr(ws["x01_m"]), self._pp("density", self.params_half),
                      _ptr(ws["density_m"]), _ptr(ws["ctx_m"]))
            finally:
                net.set_option("n_live_ptr", 0)
                net.set_option("compact_output", 0)
                if self._pig:
                    net.set_option("prepare_input_gradients", 1)
            aa = _lib.NgpAliveArgs(R=R, counts=ws["counts_m"].data_ptr(), offsets=ws["offsets_m"].data_ptr(),
                                   dt=ws["dt_m"].data_ptr(), density_out=ws["density_m"].data_ptr(), density_stride=1,
                                   min_transmittance=float(cfg.train_min_transmittance), kept=ws["kept"].data_ptr(),
                                   state=ws["ray_state"].data_ptr(), R_dev=None if rdev is None else rdev.value,
                                   resume_in=None if resume is None else resume.data_ptr(),
                                   carry_in=None if resume is None else ws["carry"].data_ptr(), kept_base=base,
                                   t_next=None if last else ws["t_next"].data_ptr(),
                                   resume_out=None if last else ws["t_resume"].data_ptr(),
                                   carry_out=None if last else ws["carry"].data_ptr())
            _call("nvo_ngp_count_alive", stream, C.byref(aa))
            base += budget
        _call("nvo_fill_i32", stream, cap, _ptr(ws["ray_idx"]), -1)
        _call("nvo_occ_pack_fused", stream, R, _ptr(ws["kept"]), cap, _ptr(ws["counts"]), _ptr(ws["offsets"]), _ptr(ws["totals"]),
              _ptr(ws["march_scratch"]), nscr, _ptr(ws["ray_idx"]), _ptr(ws["t"]), _ptr(ws["dt"]), rdev, 0,
              _ptr(self._pack_state), _ptr(ws["origins"]), _ptr(ws["directions"]), lo, hi, _ptr(ws["x01"]), self._pack_group(ws))
        ws["ray_state_on"] = True
        ws["x01_ready"] = True

    def _shade(self, ws, training: bool, stream) -> None:
        cfg = self.cfg
        R, cap = ws["R"], ws.get("n_launch", ws["cap"])
        lo, hi = cfg.aabb
        # training evaluates the raw weights, inference the moving average (tcnn Trainer: params vs. params_inference)
        self._fwd_half = self.params_half if training else self.inference_params_half()
        if not ws.pop("x01_ready", False):  # (the pack in front of this wrote them; callers that fill the packed arrays themselves)
            _call("nvo_ngp_positions", stream, cap, _ptr(ws["ray_idx"]), _ptr(ws["t"]), _ptr(ws["origins"]),
                  _ptr(ws["directions"]), lo, hi, _ptr(ws["x01"]))
        # (inference needs no d(encoded)/d(position) from the forward; the option is read at launch time)
        no_dydx = (not training) and bool(cfg.optimize_extrinsics)
        if no_dydx:
            self.density_net.set_option("prepare_input_gradients", 0)
        try:
            _call("nvo_fwd", self.density_net.handle, stream, cap, _ptr(ws["x01"]), self._pp("density", self._fwd_half),
                  _ptr(ws["density_out"]), _ptr(ws["ctx"]))
        finally:
            if no_dydx:
                self.density_net.set_option("prepare_input_gradients", 1)
        if not ws.pop("sh_ready", False):  # (rays that did not come through load_rays: inference bundles)
            _call("nvo_dirs01", stream, 3 * R, _ptr(ws["directions"]), _ptr(ws["dirs01"]))
            _call("nvo_sh_encode", stream, R, 4, _ptr(ws["dirs01"]), _ptr(ws["sh"]))
        ra = self._rgb_args(ws, training)
        _call("nvo_ngp_rgb_fwd", stream, C.byref(ra))

    def _rgb_args(self, ws, training: bool):
        return _lib.NgpRgbArgs(
            capacity=ws.get("n_launch", ws["cap"]), sh=ws["sh"].data_ptr(), density_out=ws["density_out"].data_ptr(),
            ray_idx=ws["ray_idx"].data_ptr(),
            weights=self._pp("rgb", self.params_half if training else self.inference_params_half()).value,
            rgb_out=ws["rgb_out"].data_ptr(), hidden=None,
            d_rgb_out=ws["d_rgb_out"].data_ptr() if training else None,
            d_density_out=ws["d_density_out"].data_ptr() if training else None,
            d_density_pre=ws["d_density_pre"].data_ptr() if training else None,
            d_weights=self._pp("rgb", self.grads).value if training else None,
            nonfinite_flag=self.skip_flag.data_ptr() if (training and self._leaf_flags) else None,
            dw_replicas=self._dw_rep["rgb"] if (training and self._dw_rep) else None,
            n_dw_replicas=self._dw_rep["G"] if (training and self._dw_rep) else 0)

    def _loss_args(self, ws, training: bool, has_depth: bool, background, carry_in=None, carry_out=None,
                   accumulate: bool = False):
        cfg = self.cfg
        R = ws["R"]
        R_launch = ws.get("R_launch", R) if training else R
        rdev = self._rdev(ws) if training else None
        compact = bool(training and ws.get("ray_state_on"))
        return _lib.NgpLossArgs(
            R=R_launch, capacity=ws.get("n_launch", ws["cap"]), counts=ws["counts"].data_ptr(), offsets=ws["offsets"].data_ptr(),
            ray_idx=ws["ray_idx"].data_ptr(), t=ws["t"].data_ptr(), dt=ws["dt"].data_ptr(),
            density_out=ws["density_out"].data_ptr(), density_stride=16, rgb_out=ws["rgb_out"].data_ptr(), rgb_stride=16,
            background=None if background is None else background.data_ptr(),
            gt_rgb=ws["gt_rgb"].data_ptr() if training else None,
            gt_depth=ws["gt_depth"].data_ptr() if (training and has_depth) else None,
            directions_norm=ws["directions_norm"].data_ptr(), rgb_mult=cfg.rgb_loss_mult,
            depth_mult=cfg.depth_loss_mult if has_depth else 0.0, inv_rays=1.0 / (R * self.world_size),
            loss_scale=cfg.loss_scale, out_rgb=ws["out_rgb"].data_ptr(), out_depth=ws["out_depth"].data_ptr(),
            out_accumulation=ws["out_accumulation"].data_ptr(), losses=self.losses.data_ptr() if training else None,
            d_rgb_out=ws["d_rgb_out"].data_ptr() if training else None, d_rgb_stride=16,
            d_density_pre=ws["d_density_pre"].data_ptr() if training else None,
            carry_in=None if carry_in is None else carry_in.data_ptr(),
            carry_out=None if carry_out is None else carry_out.data_ptr(), accumulate_outputs=int(bool(accumulate)),
            # (a compacted batch holds no sample behind the threshold: nothing to switch off)
            train_min_transmittance=float(cfg.train_min_transmittance) if (training and not compact) else 0.0,
            gt_depth_cov=ws["gt_depth_cov"].data_ptr() if (training and has_depth and ws.get("has_depth_cov")) else None,
            ray_state=ws["ray_state"].data_ptr() if compact else None, R_dev=None if rdev is None else rdev.value,
            world_size=int(self.world_size))

    def forward_backward(self, ws, jitter, has_depth: bool = True, background=None, leaf_flags: bool = False,
                         fused_adam=None) -> None:
        """``leaf_flags`` (single GPU): the producers raise skip_flag themselves -- the hash-grid backward (the leaf of the
        16-bit gradient chain) where it meets a non-finite dL/d(encoded), the two fused-MLP backwards where a weight-
        gradient total is not finite (an overflow inside the chain lands in dW = dZ^T H) -- instead of a scan of all
        12.6 M gradients behind the backward (nonfinite_flag: 17.6 us per step).
        ``fused_adam``: (lo, hi) from _fused_adam_plan(): the grid backward steps those parameters itself (Adam + weight
        average, nvo_set_fused_adam) and leaves their gradient unwritten."""
        stream = _stream(self.device)
        cap = self.cfg.capacity
        if leaf_flags != self._leaf_flags:
            self.density_net.set_option("nonfinite_flag_ptr", self.skip_flag.data_ptr() if leaf_flags else 0)
            self._leaf_flags = leaf_flags
        # everything the step accumulates into, cleared by ONE launch (they were six fills + two inside nvo_bwd): the
        # gradient outside the range the grid backward steps itself (nothing accumulates there), the loss slots, the
        # overflow flag, what the density network's backward used to clear itself
        g0 = self.grads.data_ptr()
        spans = [(g0, 4 * self.n_params)] if fused_adam is None else \
            [(g0, 4 * fused_adam[0]), (g0 + 4 * fused_adam[1], 4 * (self.n_params - fused_adam[1]))]
        spans.append((self.losses.data_ptr(), 4 * self.losses.numel()))
        if leaf_flags:
            spans.append((self.skip_flag.data_ptr(), 4))
        # (the per-camera gradient d_corrections accumulates across the camera optimiser's window: cleared behind its step)
        spans += self._bwd_zero_spans()
        spans = [sp for sp in spans if sp[1] > 0]
        _call("nvo_zero_ranges", stream, len(spans), (C.c_void_p * len(spans))(*[a for a, _ in spans]),
              (C.c_uint64 * len(spans))(*[b for _, b in spans]))
        self._forward(ws, True, jitter, stream)
        la = self._loss_args(ws, True, has_depth, background)
        _call("nvo_ngp_composite_loss", stream, C.byref(la))
        ra = self._rgb_args(ws, True)
        _call("nvo_ngp_rgb_bwd", stream, C.byref(ra))
        pose = self.cfg.optimize_extrinsics and self._pose_inputs is not None and "dx01" in ws
        if fused_adam is not None:
            self._set_fused_adam(True)
        try:
            _call("nvo_bwd", self.density_net.handle, stream, cap, _ptr(ws["x01"]), self._pp("density", self.params_half),
                  _ptr(ws["density_out"]), _ptr(ws["d_density_out"]), _ptr(ws["ctx"]), _ptr(ws["dx01"]) if pose else None,
                  self._pp("density", self.grads))
        finally:
            if fused_adam is not None:
                self._set_fused_adam(False)
        if self._dw_rep:
            _call("nvo_fold_replicas", stream, 2, self._dw_rep["reps"], self._dw_rep["n_rep"], self._dw_rep["n"], self._dw_rep["dst"])
        if pose:
            self._pose_backward(ws, stream)

    def _bwd_zero_spans(self):
        """What the density network's nvo_bwd clears before it accumulates (MLP weight gradient, atomically flushed grid
        ranges, scale scratch), handed to the step's single zero launch (module option external_zero: two launches less
        per step).  From here on every nvo_bwd of that network must come through forward_backward."""
        if self._bwd_zero_plan is None:
            cap = 16
            ptrs = (C.c_void_p * cap)()
            sizes = (C.c_uint64 * cap)()
            n = _lib.lib().nvo_bwd_zero_ranges(self.density_net.handle, self._pp("density", self.grads), ptrs, sizes, cap)
            if n < 0:
                raise RuntimeError(f"nvo_bwd_zero_ranges: {_lib.lib().nvo_last_error().decode()}")
            self.density_net.set_option("external_zero", 1)
            self._bwd_zero_plan = [(int(ptrs[i]), int(sizes[i])) for i in range(n)]
        return list(self._bwd_zero_plan)

    def _pose_backward(self, ws, stream) -> None:
        """Position gradients of the packed samples -> per-ray dL/do, dL/dd -> per-camera correction gradient ->
        dL/d(offset) (+ the L2 pull toward zero); the direction-encoding path is not differentiated (instant-ngp
        trains the extrinsics from the position gradient alone)."""
        cfg = self.cfg
        lo, hi = cfg.aabb
        R = ws.get("R_launch", ws["R"])
        _call("nvo_ngp_positions_bwd_dev", stream, R, cfg.capacity, _ptr(ws["counts"]), _ptr(ws["offsets"]), _ptr(ws["t"]),
              _ptr(ws["origins"]), _ptr(ws["directions"]), lo, hi, _ptr(ws["dx01"]), _ptr(ws["d_origin"]),
              _ptr(ws["d_dir"]), self._rdev(ws))
        intr, c2w = self._pose_inputs
        _call("nvo_pose_bwd_cams", stream, R, _ptr(ws["ray_indices"]), _ptr(intr), _ptr(c2w), _ptr(ws["d_origin"]),
              _ptr(ws["d_dir"]), None, _ptr(self.d_corrections), cfg.num_images)  # (adds into the window's total)

    @torch.no_grad()
    def camera_gradient(self) -> torch.Tensor:
        """dL/d(camera offsets) [F * 6] of what the current window has accumulated: the loss-scaled SUM over its training
        steps, without the L2 pull (tests; the optimiser's own step adds the pull and the window / camera scaling)."""
        _call("nvo_se3_exp_map_bwd", _stream(self.device), self.cfg.num_images, _ptr(self.pose_adjustment),
              _ptr(self.d_corrections), 0.0, 0.0, 0.0, _ptr(self.pose_grads), C.c_void_p(self.losses.data_ptr() + 6 * 4), 1)
        return self.pose_grads.clone()

    def _camera_update_due(self) -> bool:
        """Does the step about to run end a window of the camera optimiser?"""
        return self._cam_window + 1 >= max(1, int(self.cfg.extrinsic_update_every))

    def _camera_grad_scale(self) -> float:
        """What turns the window's accumulated, loss-scaled gradient into upstream's per-camera figure: the mean over the
        window, times the number of cameras the batch is spread over."""
        n = int(self.n_training_images) if self.n_training_images else int(self.cfg.num_images)
        return float(n) / (self.cfg.loss_scale * max(1, int(self.cfg.extrinsic_update_every)))

    def _camera_lr(self) -> float:
        cfg = self.cfg
        if int(cfg.extrinsic_update_every) <= 1:
            return float(cfg.extrinsic_lr)
        return max(cfg.extrinsic_lr * cfg.extrinsic_lr_decay ** (self.cam_step // max(1, int(cfg.extrinsic_lr_decay_steps))),
                   cfg.lr / 1000.0)

    def _sync_cam_dev(self) -> None:
        """Device scalars of the camera optimiser for the host's ``cam_step`` and the scheduled learning rate (written when
        either moved from outside the optimiser's own commit: first use, snapshot load, a decay boundary)."""
        lr = self._camera_lr()
        if self._cam_synced == (self.cam_step, lr):
            return
        b1, b2 = self.cfg.adam_betas
        if self._cam_synced is None or self._cam_synced[0] != self.cam_step:
            t = float(self.cam_step + 1)
            self._cam_dev.copy_(torch.tensor([lr, 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), 0.0], dtype=torch.float64).float())
            self._cam_applied_dev.fill_(int(self.cam_step))
        else:  # (only the rate moved: the optimiser's commit keeps the bias corrections)
            self._cam_dev[0:1].fill_(lr)
        self._cam_synced = (self.cam_step, lr)

    def _camera_optimizer_step(self, stream, all_reduce=None) -> None:
        """End of a window: accumulated per-camera gradient -> dL/d(offset) + L2 pull -> Adam with the camera optimiser's own
        step count and rate; the window's total is cleared behind it.  An overflow anywhere in the window skips the step."""
        cfg = self.cfg
        scale = self._camera_grad_scale()
        _call("nvo_se3_exp_map_bwd", stream, cfg.num_images, _ptr(self.pose_adjustment), _ptr(self.d_corrections),
              cfg.extrinsic_l2_reg, cfg.extrinsic_l2_reg, 1.0 / (scale * self.world_size), _ptr(self.pose_grads),
              C.c_void_p(self.losses.data_ptr() + 5 * 4), 1)
        if all_reduce is not None:
            all_reduce(self.pose_grads)
        n6 = cfg.num_images * 6
        _call("nvo_nonfinite_flag", stream, n6, _ptr(self.pose_grads), 0, _ptr(self._cam_flag))
        _call("nvo_adam_step", stream, n6, _ptr(self.pose_adjustment), _ptr(self._pose_half), _ptr(self.pose_grads), 0,
              _ptr(self.pose_exp_avg), _ptr(self.pose_exp_avg_sq), cfg.extrinsic_lr, cfg.adam_betas[0],
              cfg.adam_betas[1], cfg.adam_eps, 1, scale, 0.0, _ptr(self._cam_flag), _ptr(self._cam_dev))
        # (hyper_dev = self._cam_dev overrides the by-value learning rate and step)
        _call("nvo_opt_commit", stream, 1, 1, 0, _ptr(self._cam_applied_dev), _ptr(self._cam_flag), None, None, 2.0, 0.5,
              2000, 0.0, 0.0, C.c_void_p(self._cam_dev.data_ptr() + 4), cfg.adam_betas[0], cfg.adam_betas[1])
        _call("nvo_zero_ranges", stream, 1, (C.c_void_p * 1)(self.d_corrections.data_ptr()),
              (C.c_uint64 * 1)(4 * self.d_corrections.numel()))

    def _fused_adam_plan(self):
        """(lo, hi) of the flat parameter buffer the grid backward can step itself (its streamed hashed levels), or None.
        Single GPU with producer flags only: the verdict of the step must be final before that backward runs."""
        if not self.cfg.fuse_grid_adam:
            return None
        first, n = C.c_uint64(0), C.c_uint64(0)
        _call("nvo_fused_adam_range", self.density_net.handle, C.byref(first), C.byref(n))
        if n.value == 0:
            return None
        lo = self.segments["density"][0] + int(first.value)
        return lo, lo + int(n.value)

    def _set_fused_adam(self, on: bool) -> None:
        if not on:
            _call("nvo_set_fused_adam", self.density_net.handle, None)
            return
        self._sync_opt_dev()
        cfg = self.cfg
        ema = cfg.ema_decay > 0.0
        if ema and self.params_ema is None:
            self.params_ema = torch.zeros_like(self.params)
            self.params_ema_half = torch.zeros_like(self.params_half)
        o = self.segments["density"][0]
        a = _lib.FusedAdamArgs(
            params=self.params.data_ptr() + 4 * o, params_half=self.params_half.data_ptr() + 2 * o,
            exp_avg=self.exp_avg.data_ptr() + 4 * o, exp_avg_sq=self.exp_avg_sq.data_ptr() + 4 * o, hyper_dev=None,
            bias_dev=self._opt_dev.data_ptr() + 4, loss_scale_dev=None, skip_flag=self.skip_flag.data_ptr(), lr=cfg.lr,
            grad_scale=1.0 / cfg.loss_scale, beta1=cfg.adam_betas[0], beta2=cfg.adam_betas[1], eps=cfg.adam_eps, step=0,
            ema=self.params_ema.data_ptr() + 4 * o if ema else None,
            ema_half=self.params_ema_half.data_ptr() + 2 * o if ema else None, ema_decay=cfg.ema_decay,
            ema_step_dev=self._ema_step_dev.data_ptr() if ema else None)
        _call("nvo_set_fused_adam", self.density_net.handle, C.byref(a))

    def _sync_opt_dev(self) -> None:
        """Writes the device side of the step count (self._opt_dev, self._applied_dev) for the host's ``opt_step`` -- at
        the first step and whenever ``opt_step`` was set from outside (snapshot load); otherwise nvo_opt_commit keeps it."""
        if self._dev_synced == self.opt_step:
            return
        b1, b2 = self.cfg.adam_betas
        t = float(self.opt_step + 1)
        # ([0] is unused here: the camera optimiser keeps its own scalars, self._cam_dev)
        self._opt_dev.copy_(torch.tensor([self.cfg.lr, 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), 0.0],
                                         dtype=torch.float64).float())
        self._applied_dev.fill_(int(self.opt_step))
        self._dev_synced = self.opt_step

    @property
    def applied_steps(self) -> int:
        """Optimiser steps actually applied (a step whose gradients overflowed is skipped and does not count; device
        counter, reading it synchronises)."""
        self._sync_opt_dev()
        return int(self._applied_dev.item())

    def optimizer_step(self, fused_adam=None, camera_update: bool = True, all_reduce=None) -> None:
        """Adam of the three parameter ranges (one launch), weight average, Adam of the camera offsets, step counter.
        The bias corrections come from the device (self._opt_dev): t = applied steps + 1."""
        cfg = self.cfg
        stream = _stream(self.device)
        self._sync_opt_dev()
        bias_dev = self._opt_dev.data_ptr() + 4
        if not self._leaf_flags:
            _call("nvo_nonfinite_flag", stream, self.n_params, _ptr(self.grads), 0, _ptr(self.skip_flag))
        n_grid = self.density_net.n_params - self.n_density_mlp
        if fused_adam is not None:  # (the backward stepped [fused_adam): the tail of the grid's range)
            assert self.n_density_mlp <= fused_adam[0] and fused_adam[1] == self.density_net.n_params
            n_grid = fused_adam[0] - self.n_density_mlp
        # (offset, size, weight decay): density MLP | hash grid | rgb MLP -- l2_reg on MLP weights only; ONE launch
        batch = [_lib.AdamGroup(offset=off, n=size, lr=cfg.lr, step=0, hyper_dev=None, bias_dev=bias_dev, flag_slot=0,
                                flag_slot_set=1, weight_decay=wd, weight_decay_set=1)
                 for off, size, wd in ((0, self.n_density_mlp, cfg.l2_reg), (self.n_density_mlp, n_grid, 0.0),
                                       (self.density_net.n_params, self.n_rgb, cfg.l2_reg)) if size > 0]
        arr = (_lib.AdamGroup * len(batch))(*batch)
        if cfg.ema_decay > 0.0:
            if self.params_ema is None:
                self.params_ema = torch.zeros_like(self.params)
                self.params_ema_half = torch.zeros_like(self.params_half)
            # (Should the very first step be skipped, inference would read an all-zero average for one step: instant-ngp
            # has the same window; the debias factor itself is exact -- it follows the device counter.)
            self._ema_started = True
        if cfg.fuse_optimizer_tail:
            # ONE launch: Adam of the three ranges, the weight average of the same elements and -- by its last workgroup --
            # both counters with the next step's bias corrections (nvo_adam_step_groups_tail)
            ema = cfg.ema_decay > 0.0
            tail = _lib.AdamTail(ema=self.params_ema.data_ptr() if ema else None,
                                 ema_half=self.params_ema_half.data_ptr() if ema else None, ema_decay=cfg.ema_decay,
                                 ema_step_dev=self._ema_step_dev.data_ptr() if ema else None, ema_flag_slot=0, ema_commit=1,
                                 done_counter=self._tail_done.data_ptr(), n_commit_groups=1, active_mask=1, scale_mask=0,
                                 applied=self._applied_dev.data_ptr(), scale=None, growth_tracker=None, growth_factor=2.0,
                                 backoff_factor=0.5, growth_interval=2000, min_scale=0.0, max_scale=0.0, bias=bias_dev)
            _call("nvo_adam_step_groups_tail", stream, len(batch), arr, _ptr(self.params), _ptr(self.params_half),
                  _ptr(self.grads), 0, _ptr(self.exp_avg), _ptr(self.exp_avg_sq), cfg.adam_betas[0], cfg.adam_betas[1],
                  cfg.adam_eps, 1.0 / cfg.loss_scale, 0.0, _ptr(self.skip_flag), 0, None, None, None, C.byref(tail))
            if camera_update and cfg.optimize_extrinsics and self._pose_inputs is not None:
                self._sync_cam_dev()
                self._camera_optimizer_step(stream, all_reduce)
            return
        _call("nvo_adam_step_groups", stream, len(batch), arr, _ptr(self.params), _ptr(self.params_half), _ptr(self.grads), 0,
              _ptr(self.exp_avg), _ptr(self.exp_avg_sq), cfg.adam_betas[0], cfg.adam_betas[1], cfg.adam_eps,
              1.0 / cfg.loss_scale, 0.0, _ptr(self.skip_flag))
        if cfg.ema_decay > 0.0:
            lo, hi = (0, self.n_params)
            if fused_adam is not None:
                # the backward averaged [fused_adam) already (with the counter as it stands): the head here without
                # committing the counter, the tail below with it
                _call("nvo_ema_update_dev_part", stream, fused_adam[0], _ptr(self.params), _ptr(self.params_ema),
                      _ptr(self.params_ema_half), cfg.ema_decay, _ptr(self._ema_step_dev), _ptr(self.skip_flag))
                lo = fused_adam[1]
            _call("nvo_ema_update_dev", stream, hi - lo, C.c_void_p(self.params.data_ptr() + 4 * lo),
                  C.c_void_p(self.params_ema.data_ptr() + 4 * lo), C.c_void_p(self.params_ema_half.data_ptr() + 2 * lo),
                  cfg.ema_decay, _ptr(self._ema_step_dev), _ptr(self.skip_flag))
        if camera_update and cfg.optimize_extrinsics and self._pose_inputs is not None:
            self._sync_cam_dev()  # (a no-op once train_step has written them: nothing is uploaded inside a capture)
            self._camera_optimizer_step(stream, all_reduce)
        # the step counter and the next step's bias corrections, behind every launch that read them
        _call("nvo_opt_commit", stream, 1, 1, 0, _ptr(self._applied_dev), _ptr(self.skip_flag), None, None, 2.0, 0.5, 2000,
              0.0, 0.0, C.c_void_p(bias_dev), cfg.adam_betas[0], cfg.adam_betas[1])

    def train_step(self, ray_indices, intrinsics, c2w, images, depths, all_reduce=None, depths_cov=None):
        R = ray_indices.shape[0]
        ws = self._workspace(R, True)
        if self.step % self.cfg.density_update_every == 0:
            n_train = int(self.n_training_images) if self.n_training_images else int(images.shape[0])
            if self.cfg.mark_untrained and (self.step == 0 or n_train != self._marked_images):
                self.mark_untrained_cells(intrinsics, c2w, n_train, int(images.shape[1]), int(images.shape[2]))
            self.update_density_grid(all_reduce=all_reduce)
        self._sync_opt_dev()
        pose = bool(self.cfg.optimize_extrinsics) and "dx01" in ws
        cam_update = pose and self._camera_update_due()
        if cam_update:
            self._sync_cam_dev()
        if self.cfg.graph_step and all_reduce is None:
            self._train_step_graphed(ws, ray_indices, intrinsics, c2w, images, depths, cam_update, depths_cov)
        else:
            jitter = torch.rand(R, device=self.device)
            bg = torch.rand(R, 3, device=self.device) if self.cfg.random_background else None
            self._step_body(ws, ray_indices, intrinsics, c2w, images, depths, jitter, bg, all_reduce, cam_update, depths_cov)
        if pose:
            self._cam_window = 0 if cam_update else self._cam_window + 1
            if cam_update:
                self.cam_step += 1
                self._cam_synced = (self.cam_step, self._cam_synced[1])  # (the commit advanced the device side)
        self.opt_step += 1
        self._dev_synced = self.opt_step  # (the host mirror counts attempts; the device follows the applied steps)
        self.step += 1
        self.params_version += 1
        if self.cfg.adaptive_rays:
            self._adapt_rays(ws, R)

    def _step_body(self, ws, ray_indices, intrinsics, c2w, images, depths, jitter, bg, all_reduce=None,
                   camera_update: bool = True, depths_cov=None) -> None:
        """Every launch of one step, in order (eager, or recorded into a hipGraph by _train_step_graphed)."""
        self.load_rays(ws, ray_indices, intrinsics, c2w, images, depths, depths_cov)
        fused = self._fused_adam_plan() if all_reduce is None else None
        self.forward_backward(ws, jitter, has_depth=depths is not None, background=bg, leaf_flags=all_reduce is None,
                              fused_adam=fused)
        if all_reduce is not None:
            all_reduce(self.grads)
        self.optimizer_step(fused_adam=fused, camera_update=camera_update, all_reduce=all_reduce)
        if self.cfg.adaptive_rays:
            # samples the march found in this step, BEFORE rays were dropped at the packed capacity (the scan's total):
            # a capped measure could never exceed the target, so the batch would only ever grow
            self._measured_acc.add_(ws["totals"][:1])

    _MAX_GRAPHS = 48

    def _train_step_graphed(self, ws, ray_indices, intrinsics, c2w, images, depths, cam_update: bool = True,
                            depths_cov=None) -> None:
        """The step as ONE hipGraph replay (NgpConfig.graph_step).  In front of it, eager: the caller's ray indices are
        copied and the march jitter is drawn into the workspace's fixed buffers (same generator calls as the eager step).
        Captured once per WORKSPACE SIZE (rows: the next power of two above the ray count) and set of input buffers: the
        per-ray launches cover the workspace's rows and read the batch size from the device (self._R_dev), so the adaptive
        ray batch moves without a new capture; the very first step of an engine runs eagerly so that every kernel's code
        object is loaded before anything is recorded."""
        cfg = self.cfg
        R = ws["R"]
        ws["ray_indices"].copy_(ray_indices)
        ws["jitter"].uniform_()
        bg = None
        if cfg.random_background:
            if "background" not in ws:  # (switched on after the workspace was made)
                ws["_per_ray"]["background"] = torch.zeros(ws["R_cap"], 3, dtype=torch.float32, device=self.device)
                self._ray_views(ws, R)
            bg = ws["background"]
            bg.uniform_()
        if not self._kernels_loaded:
            self._step_body(ws, ws["ray_indices"], intrinsics, c2w, images, depths, ws["jitter"], bg, None, cam_update,
                            depths_cov)
            self._kernels_loaded = True
            return
        if cfg.ema_decay > 0.0 and self.params_ema is None:  # (allocated outside the capture)
            self.params_ema = torch.zeros_like(self.params)
            self.params_ema_half = torch.zeros_like(self.params_half)
        if self._R_dev_host != R:
            self._R_dev.fill_(R)
            self._R_dev_host = R
        ws["R_launch"] = ws["R_cap"]
        try:
            self._replay_step(ws, intrinsics, c2w, images, depths, cam_update, depths_cov, bg)
        finally:
            del ws["R_launch"]

    def _replay_step(self, ws, intrinsics, c2w, images, depths, cam_update, depths_cov, bg) -> None:
        cfg = self.cfg
        key = (ws["R_cap"], self._pack_group(ws), bool(cfg.compact_training), int(cfg.march_capacity), tuple(cfg.train_rounds),
               ws["origins"].data_ptr(), intrinsics.data_ptr(), c2w.data_ptr(), images.data_ptr(), tuple(images.shape),
               None if depths is None else depths.data_ptr(), None if depths_cov is None else depths_cov.data_ptr(),
               bool(cfg.optimize_extrinsics), bool(cfg.adaptive_rays),
               bool(cam_update), self._camera_grad_scale() if cam_update else 0.0, self._fused_adam_plan(),
               # every by-value scalar of the step's launches
               (cfg.loss_scale, cfg.lr, cfg.rgb_loss_mult, cfg.depth_loss_mult, cfg.l2_reg, cfg.extrinsic_l2_reg, cfg.ema_decay,
                cfg.cone_angle, cfg.near_distance, tuple(cfg.adam_betas), cfg.adam_eps, bool(cfg.random_background),
                int(cfg.dw_replicas), self.world_size, cfg.extrinsic_lr, cfg.train_min_transmittance),
               0 if self.params_ema is None else self.params_ema.data_ptr())
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= self._MAX_GRAPHS:
                self._graphs.clear()
            t0 = time.perf_counter()
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):  # (no cyclic garbage collection while the stream records)
                self._step_body(ws, ws["ray_indices"], intrinsics, c2w, images, depths, ws["jitter"], bg, None, cam_update,
                                depths_cov)
            self.graph_captures += 1
            self.graph_capture_seconds += time.perf_counter() - t0
            # the graph addresses these buffers: they must stay alive as long as it does
            entry = {"graph": g, "keep": (intrinsics, c2w, images, depths, depths_cov, ws), "pose_inputs": self._pose_inputs}
            self._graphs[key] = entry
        entry["graph"].replay()
        self._pose_inputs = entry["pose_inputs"]
        if cfg.ema_decay > 0.0:
            self._ema_started = True

    def _adapt_rays(self, ws, R: int) -> None:
        """rays_per_batch <- rays * target / measured samples per step (mean since the last adaptation), rounded up to
        128, every `density_update_every` steps -- ONE host read-back per 16 steps, where upstream reads its loss."""
        cfg = self.cfg
        self._measured_n += 1
        if self.step % cfg.density_update_every != 0:
            return
        measured = float(self._measured_acc.item()) / self._measured_n
        self._measured_acc.zero_()
        self._measured_n = 0
        if measured <= 0.0:
            return
        want = int(math.ceil(R * cfg.capacity / measured / 128.0)) * 128
        self.rays_per_batch = max(cfg.min_rays, min(cfg.max_rays, want))

    @torch.no_grad()
    def camera_corrections(self) -> torch.Tensor:
        """[F,3,4] pose corrections (rotation | translation) the extrinsics optimiser has learnt so far
        (counterpart of instant-ngp's nerf.training.get_camera_extrinsics offsets)."""
        _call("nvo_pose_exp_map", _stream(self.device), self.cfg.num_images, _ptr(self.pose_adjustment),
              _ptr(self.corrections), 1)
        return self.corrections.view(self.cfg.num_images, 3, 4).clone()

    @torch.no_grad()
    def density_at(self, positions: torch.Tensor) -> torch.Tensor:
        """Density (exponential activation of the density network's first output, instant-ngp's nerf activation) at
        positions [M, 3] of the engine's normalised frame, evaluated with the inference weights.  Points outside the
        scene box read 0.  Used by mesh extraction; not on the training path."""
        cfg = self.cfg
        lo, hi = cfg.aabb
        stream = _stream(self.device)
        M = positions.shape[0]
        chunk = 1 << 18
        x01 = torch.empty(chunk, 3, device=self.device)
        out = torch.empty(chunk, 16, dtype=torch.float16, device=self.device)
        ctx = torch.empty(self.density_net.ctx_bytes(chunk), dtype=torch.uint8, device=self.device)
        res = torch.empty(M, device=self.device)
        ph = self.inference_params_half()
        for c0 in range(0, M, chunk):
            p = positions[c0:c0 + chunk].to(self.device, torch.float32)
            n = p.shape[0]
            x01.zero_()
            x01[:n] = ((p - lo) / (hi - lo)).clamp_(0.0, 1.0)
            _call("nvo_fwd", self.density_net.handle, stream, chunk, _ptr(x01), self._pp("density", ph), _ptr(out), _ptr(ctx))
            inside = ((p >= lo) & (p <= hi)).all(dim=1)
            res[c0:c0 + n] = torch.where(inside, torch.exp(out[:n, 0].float().clamp(max=15.0)), torch.zeros((), device=self.device))
        return res

    def loss_dict(self) -> dict:
        vals = self.losses.sum(dim=0).tolist()
        if int(self._tail_done.item()) & 0x80000000:  # (nvo_adam_tail::done_counter: the fused optimiser tail refused to commit)
            raise RuntimeError("the optimiser launch's commit failed its check-in (counter left dirty or a stalled launch): "
                               "step counters and weight average are no longer consistent -- restart from a snapshot")
        d = {"rgb_loss": vals[0], "depth_loss": vals[1]}
        if self.cfg.optimize_extrinsics and vals[5] != 0.0:
            d["extrinsic_regularizer"] = vals[5]
        return d

    def samples_last_step(self) -> int:
        ws = self._wss[True]  # (the training workspace: an inference call in between has its own)
        return int(ws["totals"][1].item()) if ws is not None else 0

    @torch.no_grad()
    def render_rays(self, origins, directions, directions_norm, min_transmittance=None):
        """rgb / depth / accumulation of a ray bundle with the inference weights.

        ``min_transmittance`` (default NgpConfig.render_min_transmittance) > 0: two rounds -- the first
        ``render_first_round`` samples of every ray are shaded and composited, then only the rays whose transmittance is
        still above the threshold (and that have not left the scene box) take up their march where it stopped; the rounds of
        a ray are exactly the samples one uninterrupted march finds.  0: one pass over every sample.
        A round may hold more samples than the packed capacity: the march's total is read back (one synchronisation per
        round) and an overflowing round is rendered over halves of the rays -- no ray is dropped."""
        cfg = self.cfg
        R = origins.shape[0]
        ws = self._workspace(R, False)
        ws["origins"].copy_(origins)
        ws["directions"].copy_(directions)
        ws["directions_norm"].copy_(directions_norm.reshape(-1))
        stream = _stream(self.device)
        min_t = float(cfg.render_min_transmittance if min_transmittance is None else min_transmittance)
        first = int(cfg.render_first_round) if min_t > 0.0 else 1024
        self._march(ws, None, stream, None, first, ws["t_next"] if min_t > 0.0 else None)
        found = int(ws["totals"][0].item())
        self.last_render_samples = found  # (callers size their next bundle from it: pyngp.Testbed.render)
        if R > 1 and found > ws["cap"]:
            h = R // 2
            dn = directions_norm.reshape(-1)
            a = self.render_rays(origins[:h].contiguous(), directions[:h].contiguous(), dn[:h].contiguous(), min_t)
            b = self.render_rays(origins[h:].contiguous(), directions[h:].contiguous(), dn[h:].contiguous(), min_t)
            self.last_render_samples = found
            return {k: torch.cat([a[k], b[k]]) for k in a}
        self._shade_and_composite(ws, found, stream, None, ws["carry"] if min_t > 0.0 else None, False)
        if min_t > 0.0:
            # the rays that go on: not yet opaque, still inside the scene box
            alive = (ws["t_next"] >= 0.0) & (ws["carry"] < -math.log(min_t))
            resume = torch.where(alive, ws["t_next"], torch.full_like(ws["t_next"], -1.0))
            if bool(alive.any().item()):
                self.last_render_samples = max(found, self._render_second_round(ws, resume, 0, R, stream))
        return {"rgb": ws["out_rgb"].clamp(0, 1), "depth": ws["out_depth"].clone()[:, None],
                "accumulation": ws["out_accumulation"].clone()[:, None]}

    def _shade_and_composite(self, ws, found: int, stream, carry_in, carry_out, accumulate: bool) -> None:
        # the packed samples sit at the front of the workspace: the shading launches cover them, not the whole capacity
        ws["n_launch"] = min(ws["cap"], max(4096, (found + 4095) // 4096 * 4096))
        self.render_shaded_total += found  # (running count of the samples inference has shaded: diagnostics, tests)
        try:
            self._shade(ws, False, stream)
            la = self._loss_args(ws, False, False, None, carry_in, carry_out, accumulate)
            _call("nvo_ngp_composite_loss", stream, C.byref(la))
        finally:
            del ws["n_launch"]

    def _render_second_round(self, ws, resume, lo: int, hi: int, stream) -> int:
        """The rest of the march for the alive rays lo .. hi - 1 of the bundle (the others sit the launch out), added to the
        first round's outputs; halves of the range when the samples do not fit.  Returns the largest sample count a launch
        of this round held."""
        tr = ws["t_resume"]
        tr.fill_(-1.0)
        tr[lo:hi] = resume[lo:hi]
        self._march(ws, None, stream, tr, 1024 - int(self.cfg.render_first_round), None)
        found = int(ws["totals"][0].item())
        if found > ws["cap"] and hi - lo > 1:
            mid = (lo + hi) // 2
            return max(self._render_second_round(ws, resume, lo, mid, stream),
                       self._render_second_round(ws, resume, mid, hi, stream))
        if found > 0:
            self._shade_and_composite(ws, found, stream, ws["carry"], None, True)
        return found
