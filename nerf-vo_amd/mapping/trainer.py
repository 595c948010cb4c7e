"""nerfstudio-shaped Trainer / Pipeline / config objects as the reference's mapper drives them
(/root/reference/nerf_vo/mapping/nerfstudio.py:47-217).  ``train_iteration(step)`` runs ONE native
step of the engine; everything else here is bookkeeping around it."""
from __future__ import annotations

import enum
import threading
import time
from dataclasses import dataclass, field
from pathlib import Path

import torch

from .dataset import DynamicDataManagerConfig
from .model import ExtendedNerfactoModelConfig


class LazyLossDict(dict):
    """What ``train_iteration`` hands back for loss_dict / metrics_dict: a dict whose values are read from the device
    on FIRST ACCESS.  The reference only looks at them on logging steps (/root/reference/nerf_vo/mapping/nerfstudio.py:
    161-168); materialising them every iteration costs a full device sync per step and stalls the graph replay."""

    def __init__(self, fill):
        super().__init__()
        self._fill = fill

    def _load(self):
        if self._fill is not None:
            fill, self._fill = self._fill, None
            super().update(fill())

    def _wrap(name):  # noqa: N805
        def method(self, *a, **k):
            self._load()
            return getattr(dict, name)(self, *a, **k)
        method.__name__ = name
        return method

    for _n in ("__getitem__", "__iter__", "__len__", "__contains__", "__repr__", "__eq__", "keys", "values", "items",
               "get", "copy", "pop", "setdefault", "update", "__setitem__", "__delitem__"):
        locals()[_n] = _wrap(_n)
    del _n, _wrap


class TrainingCallbackLocation(enum.Enum):
    BEFORE_TRAIN_ITERATION = 1
    AFTER_TRAIN_ITERATION = 2
    AFTER_TRAIN = 3


@dataclass
class TrainingCallback:
    where_to_run: list
    func: object
    update_every_num_iters: int | None = None
    args: list = field(default_factory=list)
    kwargs: dict = field(default_factory=dict)

    def run_callback_at_location(self, step: int, location: TrainingCallbackLocation) -> None:
        if location in self.where_to_run:
            if self.update_every_num_iters is None or step % self.update_every_num_iters == 0:
                self.func(*self.args, **self.kwargs, step=step)


@dataclass
class AdamOptimizerConfig:
    lr: float = 1e-2
    eps: float = 1e-15
    betas: tuple = (0.9, 0.999)


@dataclass
class ExponentialDecaySchedulerConfig:
    lr_final: float = 1e-5
    max_steps: int = 8192


@dataclass
class ViewerConfig:
    num_rays_per_chunk: int = 1 << 15
    quit_on_train_completion: bool = True


@dataclass
class LoggingConfig:
    steps_per_log: int = 10


@dataclass
class VanillaPipelineConfig:
    datamanager: DynamicDataManagerConfig = field(default_factory=DynamicDataManagerConfig)
    model: ExtendedNerfactoModelConfig = field(default_factory=ExtendedNerfactoModelConfig)

    def setup(self, device, test_mode="val", world_size=1, local_rank=0, max_num_iterations=8192):
        return VanillaPipeline(self, device, test_mode, world_size, local_rank, max_num_iterations)


class VanillaPipeline:
    def __init__(self, config: VanillaPipelineConfig, device, test_mode="val", world_size=1, local_rank=0,
                 max_num_iterations=8192):
        self.config = config
        self.device = torch.device(device)
        self.world_size = world_size
        self.datamanager = config.datamanager.setup(device=self.device, test_mode=test_mode, world_size=world_size,
                                                    local_rank=local_rank)
        self.model = config.model.setup(num_train_data=config.datamanager.num_frames, device=self.device,
                                        world_size=world_size, max_num_iterations=max_num_iterations,
                                        num_rays=config.datamanager.train_num_rays_per_batch, rank=local_rank,
                                        use_normals=bool(getattr(config.datamanager, "use_normals", False)))
        self.training = True
        self.all_reduce = None  # set by the distributed launcher (nerf_vo_amd.parallel.GradientAllReduce)
        import os

        # hipGraph replay of the step (default); NVO_NO_GRAPH=1 launches every kernel eagerly
        self.use_graph = os.environ.get("NVO_NO_GRAPH", "0") != "1"

    def train(self):
        self.training = True
        self.model.train()

    def eval(self):
        self.training = False
        self.model.eval()

    def get_train_loss_dict(self, step: int):
        """One full native iteration (forward, losses, backward, optimiser)."""
        ds = self.datamanager.train_dataset
        eng = self.model.engine
        eng.step = step
        if self.use_graph:
            eng.train_step_graphed(ds, all_reduce=self.all_reduce)
        else:
            ray_indices, batch = self.datamanager.next_train(step)
            c2w = ds.camera_extrinsics[:, :3, :4].contiguous()
            normals = ds.world_normals01() if (ds.use_normals and eng.cfg.normal_loss_mult > 0.0) else None
            eng.train_step(ray_indices, ds.camera_intrinsics, c2w, ds.frames_color, ds.frames_depth,
                           all_reduce=self.all_reduce, normals=normals)
        totals = eng.loss_totals()  # device snapshot of this step's loss terms; read back only if somebody looks
        loss_dict = LazyLossDict(lambda: eng.loss_dict(totals))
        metrics = LazyLossDict(lambda: self.model.get_metrics_dict(loss_dict=dict(loss_dict)))
        return totals[:7].sum(), loss_dict, metrics


@dataclass
class TrainerConfig:
    project_name: str = "nerf_vo"
    experiment_name: str = "experiment"
    method_name: str = "extended_nerfacto"
    output_dir: Path = Path("outputs")
    relative_model_dir: Path = Path("nerfstudio_models")
    save_only_latest_checkpoint: bool = True
    steps_per_save: int = 1000
    steps_per_eval_batch: int = 500
    steps_per_eval_image: int = 500
    steps_per_eval_all_images: int = 25000
    max_num_iterations: int = 8192
    mixed_precision: bool = True
    pipeline: VanillaPipelineConfig = field(default_factory=VanillaPipelineConfig)
    optimizers: dict = field(default_factory=dict)
    viewer: ViewerConfig = field(default_factory=ViewerConfig)
    logging: LoggingConfig = field(default_factory=LoggingConfig)
    vis: str = "viewer"
    timestamp: str = "{timestamp}"
    load_dir: Path | None = None

    def set_timestamp(self) -> None:
        if self.timestamp == "{timestamp}":
            self.timestamp = time.strftime("%Y-%m-%d_%H%M%S")

    def print_to_terminal(self) -> None:
        print(f"[nerf_vo_amd] TrainerConfig(method={self.method_name}, max_num_iterations={self.max_num_iterations}, "
              f"rays={self.pipeline.datamanager.train_num_rays_per_batch})")

    def get_base_dir(self) -> Path:
        return Path(self.output_dir) / self.experiment_name / self.method_name / self.timestamp

    def get_checkpoint_dir(self) -> Path:
        return self.get_base_dir() / self.relative_model_dir

    def save_config(self) -> None:
        import yaml

        base = self.get_base_dir()
        base.mkdir(parents=True, exist_ok=True)
        (base / "config.yml").write_text(yaml.dump(self), "utf8")

    def is_viewer_enabled(self) -> bool:
        return False  # the viser viewer is out of scope

    def is_viewer_beta_enabled(self) -> bool:
        return False

    def setup(self, local_rank: int = 0, world_size: int = 1, device=None):
        return Trainer(self, local_rank, world_size, device)


class Trainer:
    def __init__(self, config: TrainerConfig, local_rank: int = 0, world_size: int = 1, device=None):
        self.config = config
        self.local_rank = local_rank
        self.world_size = world_size
        self.device = torch.device(device if device is not None else f"cuda:{local_rank}")
        self.train_lock = threading.Lock()
        self.callbacks: list[TrainingCallback] = []
        self.viewer_state = None
        self.checkpoint_dir = config.get_checkpoint_dir()
        self.pipeline: VanillaPipeline | None = None

    def setup(self, test_mode="val") -> None:
        # mixed_precision -> GradScaler (/root/reference/nerf_vo/mapping/nerfstudio.py:59; nerfstudio's Trainer builds
        # GradScaler(enabled=mixed_precision)): the engine's dynamic loss scale, unless the model config pins it
        if self.config.pipeline.model.dynamic_loss_scale is None:
            self.config.pipeline.model.dynamic_loss_scale = bool(self.config.mixed_precision)
        self.pipeline = self.config.pipeline.setup(device=self.device, test_mode=test_mode,
                                                   world_size=self.world_size, local_rank=self.local_rank,
                                                   max_num_iterations=self.config.max_num_iterations)
        self._apply_optimizer_config()
        if self.config.load_dir is not None:
            self._load_checkpoint(Path(self.config.load_dir))

    def _apply_optimizer_config(self) -> None:
        ecfg = self.pipeline.model.engine.cfg
        opt = self.config.optimizers
        if "fields" in opt:
            ecfg.lr_fields = opt["fields"]["optimizer"].lr
            ecfg.adam_eps = opt["fields"]["optimizer"].eps
        if "proposal_networks" in opt:
            ecfg.lr_proposal = opt["proposal_networks"]["optimizer"].lr
        if "camera_opt" in opt:
            ecfg.lr_camera = opt["camera_opt"]["optimizer"].lr
            sched = opt["camera_opt"].get("scheduler")
            if sched is not None:
                ecfg.lr_camera_final = sched.lr_final
                ecfg.max_num_iterations = sched.max_steps

    def _init_viewer_state(self) -> None:
        self.viewer_state = None

    def _update_viewer_state(self, step: int) -> None:
        return None

    def train_iteration(self, step: int):
        # loss: 0-dim device tensor (as in nerfstudio); the dictionaries are read from the device on first access
        return self.pipeline.get_train_loss_dict(step)

    def save_checkpoint(self, step: int) -> None:
        self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
        path = self.checkpoint_dir / f"step-{step:09d}.ckpt"
        # (sharded optimiser: state_dict() all-gathers the fields group's fp32 state -- every rank calls it, rank 0 writes)
        state = self.pipeline.model.state_dict(all_reduce=self.pipeline.all_reduce)
        if self.local_rank != 0 and self.world_size > 1:
            return
        torch.save({"step": step, "pipeline": state}, path)
        if self.config.save_only_latest_checkpoint:
            for f in self.checkpoint_dir.glob("*.ckpt"):
                if f != path:
                    f.unlink()

    def _load_checkpoint(self, load_dir: Path) -> None:
        ckpts = sorted(load_dir.glob("step-*.ckpt"))
        if not ckpts:
            raise FileNotFoundError(f"no checkpoint under {load_dir}")
        state = torch.load(ckpts[-1], map_location=self.device)
        self.pipeline.model.load_state_dict(state["pipeline"])


class _Profiler:
    @staticmethod
    def flush_profiler(logging_config) -> None:
        return None


profiler = _Profiler()
