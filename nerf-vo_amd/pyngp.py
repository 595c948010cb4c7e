"""``pyngp``-shaped facade over the native occupancy-grid engine (SURVEY.md section 8 row f3 / section 8b "alt
outer" boundary): the subset of NVlabs' testbed bindings that the reference touches, with the same names, argument
meaning and call order, so that its ``InstantNGP`` mapping method
(/root/reference/nerf_vo/mapping/instant_ngp.py:19-117) and ``InstantNGPRenderer`` / ``NeRFSLAMNGPRenderer``
(/root/reference/evaluation/nerf_renderer.py:221-344) run against it unchanged (``import nerf_vo_amd.pyngp as
pyngp``; INTEGRATION.md shows the two-line binding).

What the reference calls, and what answers here:
    pyngp.TestbedMode.Nerf, pyngp.Testbed(mode, 0)            -> Testbed
    pyngp.BoundingBox(min, max), pyngp.LossType.L2             -> plain value types
    create_empty_nerf_dataset(n_images, nerf_scale, nerf_offset, aabb_scale, render_aabb)
    nerf.training.{n_images_for_training, optimize_extrinsics, depth_loss_type}
    nerf.training.update_training_images(frame_ids, poses, images, depths, depths_cov, resolution,
                                         principal_point, focal_length, depth_scale, depth_cov_scale)
    nerf.training.get_camera_extrinsics(frame_idx)             -> 3x4, optimised pose, "NGP" row order (below)
    reload_network_from_file(path), shall_train, frame()
    save_snapshot(path, include_optimizer_state) / load_snapshot(path)
    render_mode = pyngp.Shade | pyngp.Depth, exposure, fov_axis, fov, nerf.{sharpen, render_with_lens_distortion,
    render_min_transmittance}, set_nerf_camera_matrix(3x4), render(width, height, spp, linear) -> float32 [H,W,4]

Pose conventions.  ``update_training_images`` receives camera-to-world matrices in the NeRF / OpenGL convention (y
up, camera looks down -z), exactly what the mapping input carries.  The reference converts poses it READS with
``m[[1, 2, 0, 3]]`` + a y/z column flip (nerf_renderer.py:244-251) and poses it SETS with the column flip +
``m[[2, 0, 1]]`` (:311-316): both are consistent with one "NGP" form, the OpenGL matrix with its rows cycled
(row 0 <- z, row 1 <- x, row 2 <- y), and that is the form ``get_camera_extrinsics`` returns and
``set_nerf_camera_matrix`` accepts.  (instant-ngp's own ``nerf_matrix_to_ngp`` is [UPSTREAM] and absent from the
reference tree; this facade is pinned to the reference's call sites, which is all its callers can observe.)

Differences from the CUDA testbed, all outside what the reference observes: training data may be handed over as
device tensors (no host round trip); snapshots are msgpack containers with this module's own schema (the upstream
.msgpack layout is [UPSTREAM]); ``render`` always uses spp = 1 rays through pixel centres; the mesh of
``compute_and_save_marching_cubes_mesh`` is extracted with marching tetrahedra (meshing.py).  There is no CPU fallback: the engine needs the HIP library and an MI355X.
"""
from __future__ import annotations

import enum
import json
import math
import os
import types

import numpy as np
import torch

from .mapping.cameras import Cameras, CameraType
from .ngp_engine import NgpConfig, NgpEngine

_TO_NGP_ROWS = [2, 0, 1]    # OpenGL c2w rows -> "NGP" row order
_FROM_NGP_ROWS = [1, 2, 0]  # and back


# rays per inference bundle, at most.  Large bundles are what the ray-per-lane march needs (nvo_occ_march_runs takes it
# from 49 152 rays on: a 1200x680 image marched in 26 launches of 65 536 rays costs 4.1 ms, in 14 of 131 072 2.3 ms) and
# they halve the scans; 1200x680 colour + depth on one box: 2^15 rays / 2^21 slots 30.0-31.6 ms, 2^16 / 2^22 25.7-26.2 (still
# wave-per-ray), 2^16 / 2^23 24.0-24.4, 2^17 / 2^23 20.8-21.9, 2^18 / 2^24 20.5-21.2.
_MAX_BUNDLE_RAYS = 1 << 17


class TestbedMode(enum.Enum):
    Nerf = 0
    Sdf = 1
    Image = 2
    Volume = 3


class LossType(enum.Enum):
    L2 = 0
    L1 = 1
    Mape = 2
    Smape = 3
    Huber = 4
    LogL1 = 5
    RelativeL2 = 6


class RenderMode(enum.Enum):
    AO = 0
    Shade = 1
    Normals = 2
    Positions = 3
    Depth = 4


Shade = RenderMode.Shade
Depth = RenderMode.Depth


class BoundingBox:
    def __init__(self, min=None, max=None) -> None:  # noqa: A002 (pyngp's argument names)
        self.min = np.full(3, np.inf) if min is None else np.asarray(min, dtype=np.float64)
        self.max = np.full(3, -np.inf) if max is None else np.asarray(max, dtype=np.float64)


def _as_device_tensor(items, device) -> torch.Tensor:
    """list of numpy arrays / tensors (what the reference passes) or one stacked tensor -> float32 on device."""
    if isinstance(items, torch.Tensor):
        return items.to(device=device, dtype=torch.float32)
    if len(items) and isinstance(items[0], torch.Tensor):
        return torch.stack([t.to(device=device, dtype=torch.float32) for t in items])
    return torch.as_tensor(np.stack([np.asarray(a, dtype=np.float32) for a in items]), device=device)


class _Training:
    def __init__(self, testbed: "Testbed") -> None:
        self._tb = testbed
        self.n_images_for_training = 0
        self.optimize_extrinsics = False  # instant-ngp's default; the reference switches it on
        # instant-ngp's default [UPSTREAM testbed.h Nerf::Training::random_bg_color = true; scripts/run.py switches it off
        # only for --nerf_compatibility]; the reference leaves it alone (nerf_vo/mapping/instant_ngp.py:33-50): every
        # training ray is composited over a random colour, and so is its target (alpha 1 everywhere: unchanged), which
        # pushes the transmittance left at the end of a ray to zero
        self.random_bg_color = True
        self.depth_loss_type = LossType.L2
        self.depth_supervision_lambda = 1.0

    def update_training_images(self, frame_ids, poses, images, depths, depths_cov, resolution, principal_point,
                               focal_length, depth_scale: float = 1.0, depth_cov_scale: float = 1.0) -> None:
        tb = self._tb
        width, height = int(resolution[0]), int(resolution[1])
        tb._ensure_engine(height, width)
        dev = tb.device
        idx = torch.as_tensor(list(frame_ids), dtype=torch.long, device=dev)
        if idx.numel() == 0:
            return
        if int(idx.max()) >= tb._n_images or int(idx.min()) < 0:
            raise RuntimeError(f"update_training_images: frame id outside the dataset of {tb._n_images} images")
        rgba = _as_device_tensor(images, dev)
        if rgba.shape[1:3] != (height, width):
            raise RuntimeError(f"update_training_images: images are {tuple(rgba.shape[1:3])}, resolution says {(height, width)}")
        tb._images[idx] = rgba[..., :3]  # linear RGB; the alpha the reference appends is 1 everywhere
        tb._depths[idx] = _as_device_tensor(depths, dev).reshape(-1, height, width, 1) * float(depth_scale)
        cov = _as_device_tensor(depths_cov, dev).reshape(-1, height, width, 1) * float(depth_cov_scale)
        tb._depths_cov[idx] = cov
        # (a dataset that only ever received ones -- the reference's fallback without DROID-SLAM's covariances,
        # instant_ngp.py:80-85 -- trains on the plain L2 term without the extra gather)
        tb._has_depths_cov = tb._has_depths_cov or bool((cov != 1.0).any().item())
        pose = _as_device_tensor(poses, dev)[:, :3, :4]
        pose = pose.clone()
        pose[:, :, 3] = pose[:, :, 3] * tb._nerf_scale + torch.as_tensor(tb._nerf_offset, dtype=torch.float32, device=dev)
        tb._poses[idx] = pose
        intr = torch.tensor([float(focal_length[0]), float(focal_length[1]), float(principal_point[0]),
                             float(principal_point[1])], dtype=torch.float32, device=dev)
        tb._intrinsics[idx] = intr
        self.n_images_for_training = max(self.n_images_for_training, int(idx.max()) + 1)

    def get_camera_extrinsics(self, frame_idx: int) -> np.ndarray:
        tb = self._tb
        if tb._engine is None:
            raise RuntimeError("get_camera_extrinsics: no training images have been set")
        pose = tb._poses[frame_idx]
        if self.optimize_extrinsics:
            # as the training rays see the camera (nvo_rays_given: direction = R_c d, origin = t + t_c) and as upstream's
            # update_transforms composes it [UPSTREAM: rotation offset times the camera's rotation, position offset ADDED]:
            # [R_c R | t + t_c] -- not the 4x4 product, whose translation R_c t + t_c is off by (R_c - 1) t
            corr = tb._engine.camera_corrections()[frame_idx]
            pose = torch.cat([corr[:3, :3] @ pose[:3, :3], pose[:3, 3:] + corr[:3, 3:]], dim=1)
        return pose.detach().cpu().numpy().astype(np.float64)[_TO_NGP_ROWS]


class Testbed:
    def __init__(self, mode: TestbedMode = TestbedMode.Nerf, device_index: int = 0) -> None:
        if mode != TestbedMode.Nerf:
            raise NotImplementedError("only TestbedMode.Nerf is on the hot path (SURVEY.md section 8)")
        if not torch.cuda.is_available():
            raise RuntimeError("pyngp.Testbed needs an MI355X device; there is no CPU fallback")
        self.device = torch.device(f"cuda:{int(device_index)}")
        self.mode = mode
        self.nerf = types.SimpleNamespace(training=_Training(self), sharpen=0.0, render_with_lens_distortion=False,
                                          render_min_transmittance=0.01)
        self.shall_train = False
        self.render_mode = Shade
        self.exposure = 0.0
        self.fov_axis = 0
        self.fov = 50.625
        self.training_step = 0
        self.loss = 0.0
        self._n_images = 0
        self._nerf_scale, self._nerf_offset, self._aabb_scale = 0.33, np.full(3, 0.5), 1  # instant-ngp defaults
        self.render_aabb = BoundingBox()
        self._network_config: dict = {}
        self._engine: NgpEngine | None = None
        self._camera = np.eye(4)[:3][_TO_NGP_ROWS]
        self._generator = torch.Generator(device=self.device)
        self._draw_scale = None
        self._render_cache = None  # (key, rgba [n,4], z-depth [n]) of the last rendered view
        self._generator.manual_seed(42)

    # ---- dataset / network set-up ------------------------------------------------------------------------
    def create_empty_nerf_dataset(self, n_images: int, nerf_scale: float = 0.33, nerf_offset=(0.5, 0.5, 0.5),
                                  aabb_scale: int = 1, render_aabb: BoundingBox | None = None) -> None:
        if self._engine is not None:
            raise RuntimeError("create_empty_nerf_dataset: the dataset already holds images")
        self._n_images = int(n_images)
        self._nerf_scale = float(nerf_scale)
        self._nerf_offset = np.asarray(nerf_offset, dtype=np.float64)
        self._aabb_scale = int(aabb_scale)
        if render_aabb is not None:
            self.render_aabb = render_aabb
        self.nerf.training.n_images_for_training = 0

    def reload_network_from_file(self, path: str = "") -> None:
        """instant-ngp's configs/nerf/base.json is what the reference loads (instant_ngp.py:16,44) and what the
        engine implements; the file lives in a submodule that may be absent, so a missing path keeps the built-in
        base configuration.  From a present file the optimiser / encoding scalars are honoured."""
        self._network_config = {}
        if path and os.path.exists(path):
            with open(path) as fh:
                self._network_config = json.load(fh)
        if self._engine is not None:
            self._engine.init_params(self._engine.cfg.seed)

    def _config(self) -> NgpConfig:
        cfg = NgpConfig(num_images=self._n_images, aabb_scale=self._aabb_scale,
                        depth_loss_mult=self.nerf.training.depth_supervision_lambda,
                        optimize_extrinsics=bool(self.nerf.training.optimize_extrinsics),
                        random_background=bool(self.nerf.training.random_bg_color))
        opt = self._network_config.get("optimizer", {})
        while "nested" in opt:  # base.json wraps Adam in ExponentialDecay / Ema
            opt = opt["nested"]
        if opt.get("otype", "Adam") == "Adam":
            cfg.lr = float(opt.get("learning_rate", cfg.lr))
            cfg.adam_betas = (float(opt.get("beta1", cfg.adam_betas[0])), float(opt.get("beta2", cfg.adam_betas[1])))
            cfg.adam_eps = float(opt.get("epsilon", cfg.adam_eps))
            cfg.l2_reg = float(opt.get("l2_reg", cfg.l2_reg))
        return cfg

    def _ensure_engine(self, height: int, width: int) -> None:
        if self._engine is not None:
            if (height, width) != self._resolution:
                raise RuntimeError("update_training_images: all training images must share one resolution")
            return
        if self._n_images <= 0:
            raise RuntimeError("create_empty_nerf_dataset must be called first")
        if self.nerf.training.depth_loss_type != LossType.L2:
            raise NotImplementedError("only LossType.L2 depth supervision is built (instant_ngp.py:48)")
        self._engine = NgpEngine(self._config(), self.device)
        self._render_cache = None
        self._resolution = (height, width)
        f32 = dict(dtype=torch.float32, device=self.device)
        n = self._n_images
        self._images = torch.zeros(n, height, width, 3, **f32)
        self._depths = torch.zeros(n, height, width, 1, **f32)
        self._depths_cov = torch.ones(n, height, width, 1, **f32)
        self._has_depths_cov = False
        self._poses = torch.eye(4, **f32)[:3].repeat(n, 1, 1)
        self._intrinsics = torch.zeros(n, 4, **f32)

    # ---- training --------------------------------------------------------------------------------------
    def frame(self) -> bool:
        """One testbed frame = one training step when ``shall_train`` and images are present (the reference calls
        it once before any image exists, instant_ngp.py:50: a no-op)."""
        n = int(self.nerf.training.n_images_for_training)
        if not self.shall_train or self._engine is None or n <= 0:
            return True
        eng = self._engine
        eng.cfg.optimize_extrinsics = bool(self.nerf.training.optimize_extrinsics)
        eng.cfg.random_background = bool(self.nerf.training.random_bg_color)
        eng.n_training_images = n
        h, w = self._resolution
        if self._draw_scale is None or self._draw_scale[0] != (n, h, w):  # (one upload per change, not per frame)
            self._draw_scale = ((n, h, w), torch.tensor([n, h, w], device=self.device, dtype=torch.float32))
        # (Testbed::train adapts the rays per batch to the marched-sample target; NgpEngine.rays_per_batch)
        u = torch.rand((eng.rays_per_batch if eng.cfg.adaptive_rays else eng.cfg.num_rays, 3), device=self.device,
                       generator=self._generator)
        # (image, row, column) = floor(u * (n, h, w)): the int64 conversion of the non-negative products truncates
        # the per-pixel depth variance weights the depth term (all ones until update_training_images supplied one: plain L2)
        eng.train_step(u.mul_(self._draw_scale[1]).long(), self._intrinsics, self._poses, self._images, self._depths,
                       depths_cov=self._depths_cov if self._has_depths_cov else None)
        self.training_step = eng.step
        return True

    # ---- snapshots -------------------------------------------------------------------------------------
    _SNAPSHOT_FORMAT = "nerf_vo_amd.pyngp.v1"

    def save_snapshot(self, path: str, include_optimizer_state: bool = False) -> None:
        import msgpack

        if self._engine is None:
            raise RuntimeError("save_snapshot: nothing has been trained")
        e = self._engine

        def raw(t: torch.Tensor) -> bytes:
            return t.detach().contiguous().cpu().numpy().tobytes()

        n = int(self.nerf.training.n_images_for_training)
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(e.cfg).items()}
        snap = {"format": self._SNAPSHOT_FORMAT, "config": cfg, "step": e.step, "opt_step": e.opt_step,
                "resolution": list(self._resolution), "n_images": self._n_images, "n_images_for_training": n,
                "nerf_scale": self._nerf_scale, "nerf_offset": self._nerf_offset.tolist(),
                "optimize_extrinsics": bool(self.nerf.training.optimize_extrinsics),
                "params": raw(e.params), "density_grid": raw(e.density_grid), "bitfield": raw(e.bitfield),
                "poses": raw(self._poses), "intrinsics": raw(self._intrinsics), "pose_adjustment": raw(e.pose_adjustment)}
        if e.params_ema is not None and e.ema_step > 0:  # what inference reads (moving average of the weights)
            snap["params_ema"] = raw(e.params_ema)
            snap["ema_step"] = e.ema_step
        snap["rays_per_batch"] = e.rays_per_batch
        snap["cam_step"] = e.cam_step  # (the camera optimiser's own step count: its learning-rate schedule resumes)
        if include_optimizer_state:
            snap["optimizer"] = {"exp_avg": raw(e.exp_avg), "exp_avg_sq": raw(e.exp_avg_sq),
                                 "pose_exp_avg": raw(e.pose_exp_avg), "pose_exp_avg_sq": raw(e.pose_exp_avg_sq)}
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "wb") as fh:
            fh.write(msgpack.packb(snap, use_bin_type=True))

    def load_snapshot(self, path: str) -> None:
        import msgpack

        with open(path, "rb") as fh:
            snap = msgpack.unpackb(fh.read(), raw=False)
        if snap.get("format") != self._SNAPSHOT_FORMAT:
            raise RuntimeError(f"load_snapshot: {path} is not a {self._SNAPSHOT_FORMAT} snapshot")
        cfg = dict(snap["config"])
        cfg["adam_betas"] = tuple(cfg["adam_betas"])
        self._n_images = int(snap["n_images"])
        self._nerf_scale, self._nerf_offset = float(snap["nerf_scale"]), np.asarray(snap["nerf_offset"])
        self._aabb_scale = int(cfg["aabb_scale"])
        self.nerf.training.optimize_extrinsics = bool(snap["optimize_extrinsics"])
        self._engine = None
        h, w = snap["resolution"]
        self._ensure_engine(int(h), int(w))
        e = self._engine
        e.cfg = NgpConfig(**cfg)

        def put(dst: torch.Tensor, blob: bytes) -> None:
            src = torch.frombuffer(bytearray(blob), dtype=dst.dtype)
            if src.numel() != dst.numel():
                raise RuntimeError("load_snapshot: buffer size does not match the configuration")
            dst.copy_(src.view(dst.shape))

        e.set_params(torch.frombuffer(bytearray(snap["params"]), dtype=torch.float32))
        if "params_ema" in snap:
            e.params_ema = torch.frombuffer(bytearray(snap["params_ema"]), dtype=torch.float32).to(self.device)
            e.params_ema_half = e.params_ema.to(torch.float16)
            e.ema_step = int(snap["ema_step"])
        e.rays_per_batch = int(snap.get("rays_per_batch", e.cfg.num_rays))
        e.cam_step = int(snap.get("cam_step", 0))
        put(e.density_grid, snap["density_grid"])
        put(e.bitfield, snap["bitfield"])
        put(self._poses, snap["poses"])
        put(self._intrinsics, snap["intrinsics"])
        put(e.pose_adjustment, snap["pose_adjustment"])
        if "optimizer" in snap:
            for name in ("exp_avg", "exp_avg_sq", "pose_exp_avg", "pose_exp_avg_sq"):
                put(getattr(e, name), snap["optimizer"][name])
        e.step, e.opt_step = int(snap["step"]), int(snap["opt_step"])
        e.params_version += 1
        self._render_cache = None
        self.training_step = e.step
        self.nerf.training.n_images_for_training = int(snap["n_images_for_training"])

    # ---- rendering -------------------------------------------------------------------------------------
    def set_nerf_camera_matrix(self, camera_matrix) -> None:
        m = np.asarray(camera_matrix, dtype=np.float64)
        if m.shape != (3, 4):
            raise RuntimeError(f"set_nerf_camera_matrix expects a 3x4 matrix, got {m.shape}")
        self._camera = m.copy()

    def render(self, width: int, height: int, spp: int = 1, linear: bool = True, rays_per_chunk: int = 0) -> np.ndarray:
        """float32 [height, width, 4].  Shade: alpha-premultiplied linear RGB + alpha (the reference divides by
        alpha, nerf_renderer.py:274-277); Depth: z-depth in every channel (it reads channel 0, :296).  Pinhole with
        the focal length of ``fov`` along ``fov_axis``, square pixels and a centred principal point, like the
        testbed's free camera.  (rays_per_chunk = 0: bundles sized so that the samples their rays find fill ~85 % of the
        inference capacity -- from the training batches' samples per ray at first, from the previous bundle's afterwards; a
        bundle that still finds more than fit is rendered in halves by NgpEngine.render_rays, no ray is dropped.)"""
        if self._engine is None:
            raise RuntimeError("render: no network has been trained or loaded")
        if not linear:
            raise NotImplementedError("render(linear=False): the reference always asks for linear output")
        if self.render_mode not in (Shade, Depth):
            raise NotImplementedError(f"render_mode {self.render_mode}: the reference uses Shade and Depth")
        # The reference renders every frame twice, once per mode (evaluation/nerf_renderer.py:259-300): both modes come
        # from ONE pass over the rays, kept until the camera, the image size or the weights change.
        eng = self._engine
        cap = int(eng.cfg.render_capacity or eng.cfg.capacity)
        adaptive = rays_per_chunk <= 0
        if adaptive:  # first bundle: sized from the samples per ray the training batches find
            per_ray = max(16.0, eng.cfg.capacity / max(1, eng.rays_per_batch))
            rays_per_chunk = max(256, min(_MAX_BUNDLE_RAYS, int(0.8 * cap / per_ray) // 256 * 256))
        min_t = float(self.nerf.render_min_transmittance)  # (upstream default 0.01; the reference sets 1e-4)
        key = (self._camera.tobytes(), float(self.fov), int(self.fov_axis), int(width), int(height), int(rays_per_chunk),
               id(eng), eng.params_version, min_t)
        if self._render_cache is None or self._render_cache[0] != key:
            focal = 0.5 * (width if self.fov_axis == 0 else height) / math.tan(0.5 * math.radians(self.fov))
            c2w = torch.tensor(self._camera[_FROM_NGP_ROWS], dtype=torch.float32).unsqueeze(0)
            cams = Cameras(fx=focal, fy=focal, cx=0.5 * width, cy=0.5 * height, height=height, width=width,
                           camera_to_worlds=c2w, camera_type=CameraType.PERSPECTIVE).to(self.device)
            bundle = cams.generate_rays(camera_indices=0, keep_shape=True)
            o = bundle.origins.reshape(-1, 3)
            d = bundle.directions.reshape(-1, 3)
            dn = bundle.metadata["directions_norm"].reshape(-1)
            n = o.shape[0]
            rgba = torch.empty(n, 4, device=self.device)
            z = torch.empty(n, device=self.device)
            lo = 0
            while lo < n:
                hi = min(n, lo + rays_per_chunk)
                nn = dn[lo:hi]
                out = eng.render_rays(o[lo:hi].contiguous(), d[lo:hi].contiguous(), nn.contiguous(), min_t)
                rgba[lo:hi, :3] = out["rgb"]
                rgba[lo:hi, 3:] = out["accumulation"]
                z[lo:hi] = out["depth"][:, 0] / nn  # distance along the ray -> z-depth
                if adaptive:  # the next bundle aims at 85 % of the capacity with this bundle's samples per ray
                    per_ray = max(16.0, eng.last_render_samples / (hi - lo))
                    rays_per_chunk = max(256, min(_MAX_BUNDLE_RAYS, int(0.85 * cap / per_ray) // 256 * 256))
                lo = hi
            self._render_cache = (key, rgba, z)
        _, rgba, z = self._render_cache
        if self.render_mode == Depth:
            return z[:, None].expand(-1, 4).reshape(height, width, 4).cpu().numpy()
        shade = torch.cat([rgba[:, :3] * (2.0 ** self.exposure), rgba[:, 3:]], dim=1)
        return shade.view(height, width, 4).cpu().numpy()

    def compute_and_save_marching_cubes_mesh(self, filename: str, resolution=(256, 256, 256), aabb: BoundingBox | None = None,
                                             thresh: float = 2.5, generate_uvs_for_obj_file: bool = False) -> None:
        """Density iso-surface at ``thresh`` (instant-ngp's default 2.5) over ``aabb`` (dataset / world coordinates, as
        the reference passes it: nerf_renderer.py:296-300; empty or infinite -> the scene box) sampled on a
        ``resolution`` grid, written as .obj or .ply.  The surface is extracted with marching tetrahedra
        (nerf_vo_amd/meshing.py) instead of upstream's marching-cubes tables: same iso-surface, another triangulation.
        Vertices are written in the same dataset coordinates the box is given in."""
        from .meshing import marching_tetrahedra, write_mesh

        if self._engine is None:
            raise RuntimeError("compute_and_save_marching_cubes_mesh: no network has been trained or loaded")
        if generate_uvs_for_obj_file:
            raise NotImplementedError("generate_uvs_for_obj_file: the reference never asks for texture coordinates")
        res = [int(r) for r in (np.asarray(resolution).reshape(-1).tolist() * 3)[:3]]
        e = self._engine
        lo_n, hi_n = e.cfg.aabb  # scene box of the engine's normalised frame
        scale, off = float(self._nerf_scale), np.asarray(self._nerf_offset, dtype=np.float64)
        box = aabb if aabb is not None else BoundingBox()
        lo = np.asarray(box.min, dtype=np.float64)
        hi = np.asarray(box.max, dtype=np.float64)
        world_lo, world_hi = (np.full(3, lo_n) - off) / scale, (np.full(3, hi_n) - off) / scale
        if not (np.isfinite(lo).all() and np.isfinite(hi).all() and (hi > lo).all()):
            lo, hi = world_lo, world_hi
        lo, hi = np.maximum(lo, world_lo), np.minimum(hi, world_hi)
        axes = [torch.linspace(float(lo[k]), float(hi[k]), res[k], device=self.device) for k in range(3)]
        gx, gy, gz = torch.meshgrid(*axes, indexing="ij")
        world = torch.stack([gx, gy, gz], dim=-1).reshape(-1, 3)
        unit = world * scale + torch.as_tensor(off, dtype=torch.float32, device=self.device)
        dens = e.density_at(unit).view(*res)
        verts, faces = marching_tetrahedra(dens, lo, hi, float(thresh))
        os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
        write_mesh(filename, verts, faces)
