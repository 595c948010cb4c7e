"""Keyframe buffer and data manager of the mapping stage.

Behavioural mirror of ``DynamicDataset`` / ``DynamicDataManager``
(/root/reference/nerf_vo/mapping/nerfstudio_utils.py:30-323): pre-allocated per-keyframe buffers that
are updated IN PLACE while a ``Cameras`` object keeps views of them; two ingest modes (per-slot
overwrite when every pose comes with a frame, append + sliding-window pose/depth refresh otherwise);
world normalisation by the first pose; NCHW -> NHWC.  Differences, all deliberate:
  * the normal images are rotated to world space once at ingest and cached, instead of being
    re-solved for every pixel of every active frame on each training step;
  * ``next_train`` hands the native engine pixel INDICES; gathering and ray generation are HIP
    kernels (nvo_gather_pixels / nvo_raygen), not torch indexing.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path

import torch

from .cameras import Cameras, CameraType

# world alignment used by the reference: first camera looks along +y with z up
_WORLD_ALIGN = [[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, -1.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]]


def opencv_to_opengl(c2w: torch.Tensor) -> torch.Tensor:
    """Negate the camera y and z axes (what the enhancement stage does before mapping,
    /root/reference/nerf_vo/enhancement/enhancement_module.py:117-118)."""
    out = c2w.clone()
    out[..., :3, 1:3] *= -1
    return out


class DynamicDataset(torch.utils.data.Dataset):
    def __init__(self, num_frames: int, frame_height: int, frame_width: int,
                 device: torch.device = torch.device("cuda:0"), use_normals: bool = True,
                 dir_prediction: str | None = None) -> None:
        super().__init__()
        self.device = torch.device(device)
        self.use_normals = use_normals
        self.num_frames = num_frames
        self.num_active_frames = 0
        self.frame_height = frame_height
        self.frame_width = frame_width
        self.normalization_matrix = None
        self.scene_box_aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], device=self.device)

        f32 = dict(dtype=torch.float32, device=self.device)
        self.camera_intrinsics = torch.zeros((num_frames, 4), **f32)
        self.camera_extrinsics = torch.eye(4, **f32).repeat(num_frames, 1, 1)
        self.frames_color = torch.zeros((num_frames, frame_height, frame_width, 3), **f32)
        self.frames_depth = torch.zeros((num_frames, frame_height, frame_width, 1), **f32)
        if use_normals:
            self.frames_normal = torch.zeros((num_frames, frame_height, frame_width, 3), **f32)
            # world-space, [0,1]-mapped normals, refreshed when a frame's normal or pose changes
            self._normal_world01 = torch.zeros((num_frames, frame_height, frame_width, 3), **f32)

        if dir_prediction is not None:
            saved = torch.load(f"{dir_prediction}/dataset.pt", map_location=self.device)
            n = saved["camera_extrinsics"].shape[0]
            self.num_active_frames = n
            self.camera_intrinsics[: saved["camera_intrinsics"].shape[0]] = saved["camera_intrinsics"]
            self.camera_extrinsics[:n] = saved["camera_extrinsics"]
            self.frames_color[:n] = saved["frames_color"]
            self.frames_depth[:n] = saved["frames_depth"]
            if use_normals and "frames_normal" in saved:
                self.frames_normal[:n] = saved["frames_normal"]
                self._refresh_world_normals(torch.arange(n, device=self.device))

        # views, not copies: in-place buffer updates stay visible to the ray generator
        self.cameras = Cameras(
            camera_to_worlds=self.camera_extrinsics[:, :3], fx=self.camera_intrinsics[:, 0],
            fy=self.camera_intrinsics[:, 1], cx=self.camera_intrinsics[:, 2], cy=self.camera_intrinsics[:, 3],
            width=frame_width, height=frame_height, camera_type=CameraType.PERSPECTIVE)
        self.metadata = {}

    # ---- torch Dataset protocol ---------------------------------------------------------------
    def __len__(self) -> int:
        return self.num_active_frames if self.num_active_frames > 0 else self.num_frames

    def __getitem__(self, frame_index: int) -> dict:
        return self.get_frame(frame_index)

    def get_frame(self, frame_index: int) -> dict:
        data = {"image_idx": frame_index, "image": self.frames_color[frame_index],
                "depth_image": self.frames_depth[frame_index]}
        if self.use_normals:
            data["normal_image"] = self._normal_world01[frame_index]
        return data

    def get_dataset(self) -> dict:
        n = self.num_active_frames
        data = {"image_idx": torch.arange(0, n, dtype=torch.long, device=self.device),
                "image": self.frames_color[:n], "depth_image": self.frames_depth[:n]}
        if self.use_normals:
            data["normal_image"] = self._normal_world01[:n]
        return data

    def world_normals01(self) -> torch.Tensor:
        """The whole (fixed-address) world-space normal buffer in the (n+1)/2 colour space; rows
        >= num_active_frames are never sampled."""
        return self._normal_world01

    # ---- ingest -------------------------------------------------------------------------------
    def update(self, input: dict) -> None:
        self.insert_update(self.prepare_update(input))

    def prepare_update(self, input: dict) -> dict:
        key_idx = input["keyframe_indices"]
        assert int(key_idx.max()) < self.num_frames, "keyframe index beyond the pre-allocated buffer"
        n_new = input["frames_color"].shape[0]
        if input["camera_extrinsics"].shape[0] == n_new:
            # every pose comes with its frame (dense tracker): address slots directly
            indices = key_idx
            num_active = int(key_idx.max()) + 1
        else:
            # sparse tracker: new frames are appended, poses/depths of the whole window refreshed
            indices = torch.arange(self.num_active_frames, self.num_active_frames + n_new)
            num_active = self.num_active_frames + n_new

        extr = input["camera_extrinsics"].detach().to(torch.float32)
        if self.normalization_matrix is None:
            align = torch.tensor(_WORLD_ALIGN, dtype=extr.dtype, device=extr.device)
            self.normalization_matrix = torch.linalg.solve(extr[0], align)  # inv(E0) @ M
        # N @ E_k for every pose of the packet, written as the reference writes it (nerfstudio_utils.py:197-199: the double
        # permute(2, 1, 0) is a batched left-multiply)
        extr = (self.normalization_matrix.to(extr.device) @ extr.permute(2, 1, 0)).permute(2, 1, 0)

        out = {
            "indices": indices, "keyframe_indices": key_idx, "num_active_frames": num_active,
            "camera_intrinsics": input["camera_intrinsics"].detach().clone(),
            "camera_extrinsics": extr,
            "frames_color": input["frames_color"].detach().permute(0, 2, 3, 1),
            "frames_depth": input["frames_depth"].detach().permute(0, 2, 3, 1),
        }
        if self.use_normals:
            out["frames_normal"] = input["frames_normal"].detach().permute(0, 2, 3, 1)
        return out

    def insert_update(self, input: dict) -> None:
        # bumped on every ingest: a step prefix launched ahead of time (engine pipelining) is re-done when it saw
        # an older state of the buffer
        self.version = getattr(self, "version", 0) + 1
        dev = self.device
        idx, key = input["indices"], input["keyframe_indices"]

        def put(buf: torch.Tensor, ids: torch.Tensor, src: torch.Tensor) -> None:
            """buf[ids] = src as the reference writes it (nerfstudio_utils.py:217-227), without the index tensor when ``ids``
            is a run of consecutive slots -- what every tracker delivers (a new frame appended, a sliding window of
            poses / depths refreshed): a slice copy instead of an H2D copy of the indices + an index_put over up to
            26 x 480 x 640 elements per buffer.  Same values either way."""
            ids_l = ids.tolist()
            src = src.to(dev)
            if ids_l and ids_l == list(range(ids_l[0], ids_l[0] + len(ids_l))):
                buf[ids_l[0]:ids_l[0] + len(ids_l)].copy_(src)
            else:
                buf[ids.to(dev)] = src

        put(self.camera_intrinsics, idx, input["camera_intrinsics"])
        put(self.camera_extrinsics, key, input["camera_extrinsics"])
        put(self.frames_color, idx, input["frames_color"])
        put(self.frames_depth, key, input["frames_depth"])
        if self.use_normals:
            put(self.frames_normal, idx, input["frames_normal"])
            self._refresh_world_normals(torch.unique(torch.cat([idx.to(dev), key.to(dev)])))
        self.num_active_frames = input["num_active_frames"]

    def _refresh_world_normals(self, frames: torch.Tensor) -> None:
        """(R^-1 n + 1) / 2 for the given frames -- the per-step solve of the reference
        (nerfstudio_utils.py:145-153) hoisted to ingest time."""
        # The reference's own expression on SLICE VIEWS of the buffers, one solve per run of consecutive frames (ingest
        # touches one or two such runs): torch.linalg.solve takes a different path for a strided view of the 4x4 pose
        # buffer than for a gathered contiguous copy and the results differ in the last bit -- with views the fixtures
        # generated by the reference's get_dataset() are reproduced bit for bit (tests/test_dataset_golden_cpu.py).
        ids = sorted(set(int(i) for i in frames.tolist()))
        h, w = self.frame_height, self.frame_width
        lo = 0
        while lo < len(ids):
            hi = lo
            while hi + 1 < len(ids) and ids[hi + 1] == ids[hi] + 1:
                hi += 1
            a, b = ids[lo], ids[hi] + 1
            f = b - a
            self._normal_world01[a:b] = (torch.linalg.solve(
                self.camera_extrinsics[a:b, :3, :3],
                self.frames_normal[a:b].permute(0, 3, 1, 2).reshape(f, 3, h * w)).reshape(f, 3, h, w).permute(0, 2, 3, 1) + 1) / 2
            lo = hi + 1

    # ---- snapshot ------------------------------------------------------------------------------
    def save_dataset(self, dir_prediction: str) -> None:
        n = self.num_active_frames
        data = {"camera_intrinsics": self.camera_intrinsics, "camera_extrinsics": self.camera_extrinsics[:n],
                "frames_color": self.frames_color[:n], "frames_depth": self.frames_depth[:n]}
        if self.use_normals:
            data["frames_normal"] = self.frames_normal[:n]
        Path(dir_prediction).mkdir(parents=True, exist_ok=True)
        torch.save(data, f"{dir_prediction}/dataset.pt")


@dataclass
class DynamicDataManagerConfig:
    train_num_rays_per_batch: int = 4096
    eval_num_rays_per_batch: int = 4096
    camera_optimizer: object = None
    num_frames: int = 128
    frame_height: int = 480
    frame_width: int = 640
    use_normals: bool = True
    dir_prediction: str | None = None

    def setup(self, device, test_mode="test", world_size=1, local_rank=0):
        return DynamicDataManager(self, device=device, test_mode=test_mode, world_size=world_size,
                                  local_rank=local_rank)


class DynamicDataManager:
    """``next_train`` returns (ray_indices [R,3] int64, batch dict) -- pixel sampling exactly like
    nerfstudio's PixelSampler.sample_method: floor(rand(R,3) * [n_active, H, W])."""

    def __init__(self, config: DynamicDataManagerConfig, device=torch.device("cuda:0"), test_mode="test",
                 world_size: int = 1, local_rank: int = 0):
        self.config = config
        self.device = torch.device(device)
        self.world_size = world_size
        self.local_rank = local_rank
        self.test_mode = test_mode
        self.train_dataset = DynamicDataset(
            num_frames=config.num_frames, frame_height=config.frame_height, frame_width=config.frame_width,
            device=self.device, use_normals=config.use_normals, dir_prediction=config.dir_prediction)
        self.eval_dataset = None
        # rank-offset stream so that every rank draws its own rays (weak scaling)
        self.generator = torch.Generator(device=self.device)
        self.generator.manual_seed(42 + 1000003 * local_rank)
        self.setup_train()

    def setup_train(self):
        self.train_count = 0

    def setup_eval(self):
        pass

    def sample_pixels(self, num_rays: int) -> torch.Tensor:
        ds = self.train_dataset
        scale = torch.tensor([ds.num_active_frames, ds.frame_height, ds.frame_width], device=self.device)
        u = torch.rand((num_rays, 3), device=self.device, generator=self.generator)
        return torch.floor(u * scale).long()

    def next_train(self, step: int):
        self.train_count += 1
        ray_indices = self.sample_pixels(self.config.train_num_rays_per_batch)
        return ray_indices, {"indices": ray_indices}

    def next_eval(self, step: int):
        return self.next_train(step)

    def next_eval_image(self, step: int):
        import random

        frame_index = random.randint(0, self.train_dataset.num_active_frames - 1)
        bundle = self.train_dataset.cameras.generate_rays(camera_indices=frame_index, keep_shape=True)
        return frame_index, bundle, self.train_dataset[frame_index]

    def get_train_rays_per_batch(self) -> int:
        return self.config.train_num_rays_per_batch

    def get_eval_rays_per_batch(self) -> int:
        return self.config.eval_num_rays_per_batch

    def get_datapath(self) -> Path:
        return Path()

    def get_param_groups(self) -> dict:
        return {}
