"""nerfstudio-shaped model objects over the native engine: ``ExtendedNerfactoModel`` (depth-nerfacto +
the reference's normal-loss hook, /root/reference/nerf_vo/mapping/nerfstudio_utils.py:326-350) and
the SE3 ``CameraOptimizer`` (/root/reference/nerf_vo/mapping/nerfstudio.py:64,208-216).

Only the surface the reference's callers touch is mirrored: ``model.config``,
``model.camera_optimizer(indices)``, ``model.get_outputs_for_camera_ray_bundle(bundle)``,
``get_metrics_dict`` / ``get_loss_dict``, ``train()/eval()``.  All arithmetic is in HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import torch

from .. import _lib
from ..engine import EngineConfig, NerfactoEngine
from ..tinycudann.modules import _ptr, _stream
from .cameras import RayBundle


@dataclass
class CameraOptimizerConfig:
    mode: str = "SE3"
    trans_l2_penalty: float = 1e-2
    rot_l2_penalty: float = 1e-3


@dataclass
class DepthNerfactoModelConfig:
    """nerfacto defaults [UPSTREAM] + depth-nerfacto fields; the reference overrides the loss
    multipliers at /root/reference/nerf_vo/mapping/nerfstudio.py:71-82."""
    near_plane: float = 0.05
    far_plane: float = 1000.0
    num_proposal_samples_per_ray: tuple = (256, 96)
    num_nerf_samples_per_ray: int = 48
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    orientation_loss_mult: float = 0.0001
    pred_normal_loss_mult: float = 0.001
    depth_loss_mult: float = 1e-3
    predict_normals: bool = False
    is_euclidean_depth: bool = False
    depth_sigma: float = 0.01
    should_decay_sigma: bool = False
    eval_num_rays_per_chunk: int = 1 << 15
    mlp_dtype: str = "f16"  # "bf16": fused MLPs on bf16 MFMA, hash tables fp16 + fp32 accumulate (BASELINE configs[4])
    # EngineConfig.deterministic / .dynamic_loss_scale.  None = follow TrainerConfig.mixed_precision (Trainer.setup):
    # mixed_precision=True (/root/reference/nerf_vo/mapping/nerfstudio.py:59) means torch's GradScaler() around the step
    # -- scale 65536, x2 after 2000 clean steps, x0.5 on overflow -- on top of tcnn's 16-bit networks; False = tcnn's own
    # static scale 128.  A model built without a trainer treats None as True (the reference's regime).
    deterministic: bool = False
    dynamic_loss_scale: bool | None = None
    camera_optimizer: CameraOptimizerConfig = field(default_factory=CameraOptimizerConfig)


@dataclass
class ExtendedNerfactoModelConfig(DepthNerfactoModelConfig):
    normal_loss_mult: float = 1e-5

    def setup(self, num_train_data: int, device, world_size: int = 1, max_num_iterations: int = 8192,
              num_rays: int = 4096, seed: int = 1337, rank: int = 0, use_normals: bool = False):
        return ExtendedNerfactoModel(self, num_train_data, device, world_size, max_num_iterations, num_rays, seed, rank,
                                     use_normals)


class CameraOptimizer(torch.nn.Module):
    """pose_adjustment [num_cameras, 6] (zeros) -> exp_map_SE3 -> [n,3,4] corrections."""

    def __init__(self, config: CameraOptimizerConfig, engine: NerfactoEngine):
        super().__init__()
        self.config = config
        self.engine = engine
        self.num_cameras = engine.cfg.num_images

    @property
    def pose_adjustment(self) -> torch.Tensor:
        return self.engine.view("camera_opt.pose_adjustment").view(self.num_cameras, 6)

    def forward(self, indices: torch.Tensor) -> torch.Tensor:
        if self.config.mode == "off":
            return torch.eye(4, device=self.engine.device)[None, :3, :4].repeat(indices.shape[0], 1, 1)
        tangent = self.pose_adjustment[indices.to(self.engine.device).long()].contiguous()
        out = torch.empty(tangent.shape[0], 3, 4, device=tangent.device)
        mode = 1 if self.config.mode == "SO3xR3" else 0
        _lib.check(_lib.lib().nvo_pose_exp_map(_stream(tangent.device), tangent.shape[0], _ptr(tangent), _ptr(out), mode),
                   "nvo_pose_exp_map")
        return out

    def all_corrections(self) -> torch.Tensor:
        return self.forward(torch.arange(self.num_cameras, device=self.engine.device))


class _LazyOutputs(dict):
    """Render outputs whose expensive members are computed when somebody asks for them."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._lazy = {}

    def __missing__(self, key):
        if key in self._lazy:
            self[key] = self._lazy.pop(key)()
            return self[key]
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    # everything that enumerates the outputs sees the lazy members too (nerfstudio-style per-key loops: metrics, image
    # writers, the viewer): they are computed at that point
    def _materialise(self):
        for key in list(self._lazy):
            self[key] = self._lazy.pop(key)()

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)

    def __len__(self):
        return dict.__len__(self) + len(self._lazy)


class ExtendedNerfactoModel:
    def __init__(self, config: ExtendedNerfactoModelConfig, num_train_data: int, device, world_size: int = 1,
                 max_num_iterations: int = 8192, num_rays: int = 4096, seed: int = 1337, rank: int = 0,
                 use_normals: bool = False):
        if config.is_euclidean_depth:
            raise NotImplementedError("is_euclidean_depth=True is not used by the reference (nerfstudio.py:79)")
        self.config = config
        self.device = torch.device(device)
        ecfg = EngineConfig(
            num_images=num_train_data, num_rays=num_rays, near_plane=config.near_plane, far_plane=config.far_plane,
            num_proposal_samples=tuple(config.num_proposal_samples_per_ray),
            num_nerf_samples=config.num_nerf_samples_per_ray, interlevel_loss_mult=config.interlevel_loss_mult,
            distortion_loss_mult=config.distortion_loss_mult, depth_loss_mult=config.depth_loss_mult,
            depth_sigma=config.depth_sigma, normal_loss_mult=float(config.normal_loss_mult),
            max_num_iterations=max_num_iterations, seed=seed, mlp_dtype=config.mlp_dtype,
            expect_normals=bool(use_normals) and float(config.normal_loss_mult) > 0.0,
            deterministic=bool(config.deterministic),
            dynamic_loss_scale=True if config.dynamic_loss_scale is None else bool(config.dynamic_loss_scale),
            optimize_poses=config.camera_optimizer.mode in ("SE3", "SO3xR3"),
            camera_mode=config.camera_optimizer.mode if config.camera_optimizer.mode in ("SE3", "SO3xR3") else "SE3",
            camera_trans_l2_penalty=config.camera_optimizer.trans_l2_penalty,
            camera_rot_l2_penalty=config.camera_optimizer.rot_l2_penalty)
        self.engine = NerfactoEngine(ecfg, self.device, world_size=world_size, rank=rank)
        self.camera_optimizer = CameraOptimizer(config.camera_optimizer, self.engine)
        self.training = True
        # Loss terms whose multiplier is 0 in every shipped configuration (orientation / predicted
        # normals, nerfstudio.py:74-75) contribute exactly zero loss and zero gradient; their heads are
        # not evaluated.  The monosdf normal loss (normal_loss_mult; enhancement modes containing
        # 'normal') runs inside the fused render/loss kernel on the analytic normals of the main field.

    # ---- module protocol ---------------------------------------------------------------------
    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def get_param_groups(self) -> dict:
        e = self.engine
        return {g: [e.params[lo:hi]] for g, (lo, hi) in e.group_ranges.items()}

    def state_dict(self, all_reduce=None) -> dict:
        """``all_reduce``: the step's GradientAllReduce when the optimiser is sharded (multi-GPU): the fp32 master weights
        and Adam moments of the fields group are current on their owner rank only and are all-gathered first -- a
        COLLECTIVE, so every rank must call state_dict() (Trainer.save_checkpoint does)."""
        e = self.engine
        e.sync_sharded_state(all_reduce)
        return {"layout": e.layout(), "mlp_dtype": e.cfg.mlp_dtype,
                "params": e.params.detach().clone(), "exp_avg": e.exp_avg.clone(), "exp_avg_sq": e.exp_avg_sq.clone(),
                "opt_steps": dict(e.opt_steps), "step": e.step,
                "steps_since_proposal_update": e.steps_since_proposal_update,
                # GradScaler state (torch: scaler.state_dict() -> scale, _growth_tracker)
                "scaler": {"scale": e.current_loss_scale(), "growth_tracker": int(e.dev_growth_tracker.item())}}

    def load_state_dict(self, state: dict) -> None:
        e = self.engine
        # the flat buffers only mean something together with their segment table (a build with another padding or segment
        # order would load parameters into the wrong networks, silently when the totals happen to agree)
        have = [tuple(x) for x in state.get("layout", [])]
        want = [tuple(x) for x in e.layout()]
        if have and have != want:
            diff = next((a, b) for a, b in zip(have + [None] * len(want), want + [None] * len(have)) if a != b)
            raise ValueError("checkpoint layout does not match this engine's parameter layout: first difference "
                             f"checkpoint {diff[0]} vs engine {diff[1]}")
        if not have and int(state["params"].numel()) != e.n_params:
            raise ValueError(f"checkpoint without a layout table holds {int(state['params'].numel())} parameters, this engine "
                             f"{e.n_params} (written by an older build: re-train or convert)")
        e.set_params(state["params"])
        e.exp_avg.copy_(state["exp_avg"])
        e.exp_avg_sq.copy_(state["exp_avg_sq"])
        e.opt_steps = dict(state["opt_steps"])
        e.step = int(state["step"])
        e.steps_since_proposal_update = int(state["steps_since_proposal_update"])
        if "scaler" in state and e.cfg.dynamic_loss_scale:
            e.dev_loss_scale.fill_(float(state["scaler"]["scale"]))
            e.dev_growth_tracker.fill_(int(state["scaler"]["growth_tracker"]))

    # ---- inference ---------------------------------------------------------------------------
    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> dict:
        """Chunked eval forward over a full-image bundle ([H,W] shaped); outputs keep that shape.  The chunks
        (``eval_num_rays_per_chunk`` rays each, nerfstudio's Model.get_outputs_for_camera_ray_bundle) run as ONE captured
        graph per image shape (NerfactoEngine.render_image).  ``outputs['normals']`` (predict_normals) costs an extra
        input-gradient pass of the base network per chunk and is produced on FIRST ACCESS: the reference's renderer
        reads 'rgb' and 'depth' only (/root/reference/evaluation/nerf_renderer.py:161-167)."""
        shape = camera_ray_bundle.origins.shape[:-1]
        origins = camera_ray_bundle.origins.reshape(-1, 3)
        directions = camera_ray_bundle.directions.reshape(-1, 3)
        dnorm = camera_ray_bundle.metadata["directions_norm"].reshape(-1)
        chunk = int(self.config.eval_num_rays_per_chunk)
        eng = self.engine
        res = eng.render_image(origins, directions, dnorm, normals=False, chunk=chunk)
        out = _LazyOutputs({k: v.view(*shape, -1) for k, v in res.items()})
        if self.config.predict_normals:
            out._lazy["normals"] = lambda: eng.render_image(origins, directions, dnorm, normals=True, chunk=chunk)[
                "normals"].view(*shape, -1)
        return out

    def get_outputs(self, ray_bundle: RayBundle) -> dict:
        return self.get_outputs_for_camera_ray_bundle(ray_bundle)

    # ---- training-time dictionaries (filled by the last native step) --------------------------
    def get_metrics_dict(self, outputs=None, batch=None, loss_dict: dict | None = None) -> dict:
        ld = self.engine.loss_dict() if loss_dict is None else loss_dict
        cfg = self.config
        metrics = {"distortion": ld["distortion_loss"] / cfg.distortion_loss_mult if cfg.distortion_loss_mult else 0.0,
                   "depth_loss": ld["depth_loss"] / cfg.depth_loss_mult if cfg.depth_loss_mult else 0.0}
        if "normal_loss" in ld and cfg.normal_loss_mult > 0.0:
            metrics["normal_loss"] = ld["normal_loss"] / cfg.normal_loss_mult
        return metrics

    def get_loss_dict(self, outputs=None, batch=None, metrics_dict=None) -> dict:
        return self.engine.loss_dict()


def multiply(pose_a: torch.Tensor, pose_b: torch.Tensor) -> torch.Tensor:
    """nerfstudio utils.poses.multiply for [...,3,4] poses (reference use: nerfstudio.py:208)."""
    r1, t1 = pose_a[..., :3, :3], pose_a[..., :3, 3:]
    r2, t2 = pose_b[..., :3, :3], pose_b[..., :3, 3:]
    return torch.cat([r1 @ r2, t1 + r1 @ t2], dim=-1)
