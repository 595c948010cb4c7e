"""Synthetic "Replica-shaped" keyframe sequence: an analytic textured room
seen from a smooth orbit, rendered by closed-form ray/box intersection so that colour, depth and
normals are exact.  Used by bench.py, the smoke test and the PSNR harness because no dataset can be
downloaded here.  Shapes / intrinsics follow the reference's Replica setup
(/root/reference/datasets/replica.json + /root/reference/nerf_vo/data/data_utils.py:24-34 scaling;
192 keyframes per /root/reference/configs/nerf_vo_replica.yaml:15).

Poses are produced in the OpenCV convention the tracker emits (x right, y down, z forward;
/root/reference/nerf_vo/tracking/dpvo.py:90-92); the keyframe-ingest code applies the
OpenCV->OpenGL flip and the world normalisation exactly like the reference does.
"""
from __future__ import annotations

import math

import torch

REPLICA_NATIVE = dict(w=1200, h=680, fx=600.0, fy=600.0, cx=599.5, cy=339.5)


def replica_intrinsics(height: int, width: int) -> tuple[float, float, float, float]:
    sx, sy = width / REPLICA_NATIVE["w"], height / REPLICA_NATIVE["h"]
    return (REPLICA_NATIVE["fx"] * sx, REPLICA_NATIVE["fy"] * sy, REPLICA_NATIVE["cx"] * sx, REPLICA_NATIVE["cy"] * sy)


def orbit_poses_opencv(n: int, radius: float = 0.9, height_amp: float = 0.25, device="cpu") -> torch.Tensor:
    """[n,4,4] camera-to-world matrices (OpenCV axes) on a closed orbit inside the room, looking at a
    slowly moving target near the centre."""
    t = torch.arange(n, dtype=torch.float32, device=device) / n * 2 * math.pi
    eye = torch.stack([radius * torch.cos(t), radius * torch.sin(t), height_amp * torch.sin(2 * t)], dim=1)
    target = torch.stack([0.3 * torch.cos(t + 2.0), 0.3 * torch.sin(t + 2.0), 0.1 * torch.cos(t)], dim=1) * -1.0
    fwd = torch.nn.functional.normalize(target - eye, dim=1)
    up = torch.tensor([0.0, 0.0, 1.0], device=device).expand_as(fwd)
    right = torch.nn.functional.normalize(torch.cross(fwd, up, dim=1), dim=1)
    down = torch.cross(fwd, right, dim=1)
    c2w = torch.eye(4, device=device).repeat(n, 1, 1)
    c2w[:, :3, 0] = right
    c2w[:, :3, 1] = down
    c2w[:, :3, 2] = fwd
    c2w[:, :3, 3] = eye
    return c2w


def _texture(p: torch.Tensor, normal_axis: torch.Tensor) -> torch.Tensor:
    """Smooth + checker texture of the hit point, different per wall."""
    f = 6.0
    base = 0.5 + 0.5 * torch.sin(p * f + normal_axis[..., None].float() * 1.3)
    # the checker is evaluated in the two in-wall coordinates only: the coordinate along the wall normal is +-1 up
    # to rounding, and floor(4 * (+-1 +- eps)) would flip per pixel (salt noise no view-consistent model can fit)
    in_wall = torch.ones_like(p).scatter_(-1, normal_axis[..., None], 0.0)
    checker = ((torch.floor(p * 4) * in_wall).sum(dim=-1) % 2)[..., None]
    return (0.75 * base + 0.25 * checker).clamp(0.0, 1.0)


@torch.no_grad()
def render_room(c2w_cv: torch.Tensor, height: int, width: int, intr, half_extent: float = 2.0):
    """Ray-cast the axis-aligned room [-h,h]^3 from inside.  c2w_cv [n,4,4] OpenCV.  Returns
    color [n,3,H,W], z-depth [n,1,H,W], camera-frame normals [n,3,H,W] (all float32, on c2w's device)."""
    dev = c2w_cv.device
    fx, fy, cx, cy = intr
    ys, xs = torch.meshgrid(torch.arange(height, device=dev, dtype=torch.float32) + 0.5,
                            torch.arange(width, device=dev, dtype=torch.float32) + 0.5, indexing="ij")
    d_cam = torch.stack([(xs - cx) / fx, (ys - cy) / fy, torch.ones_like(xs)], dim=-1)  # z-depth parametrised
    colors, depths, normals = [], [], []
    for i in range(c2w_cv.shape[0]):
        rot, o = c2w_cv[i, :3, :3], c2w_cv[i, :3, 3]
        d = d_cam @ rot.T
        inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
        t_far = torch.maximum((half_extent - o) * inv, (-half_extent - o) * inv)
        t_hit, axis = t_far.min(dim=-1)
        p = o + d * t_hit[..., None]
        colors.append(_texture(p / half_extent, axis).permute(2, 0, 1))
        depths.append(t_hit[None])  # d_cam has z = 1, so t is the z-depth
        n_world = torch.zeros_like(p)
        sign = -torch.sign(torch.gather(d, -1, axis[..., None]))[..., 0]
        n_world.scatter_(-1, axis[..., None], sign[..., None])
        normals.append((n_world @ rot).permute(2, 0, 1))
    return torch.stack(colors), torch.stack(depths), torch.stack(normals)


@torch.no_grad()
def make_sequence(num_keyframes: int = 192, height: int = 480, width: int = 640, device="cpu",
                  depth_clip: float = 5.0, scene_scale: float = 1.0):
    """Dict in the schema the mapping stage receives from enhancement, for the
    whole sequence at once: camera_intrinsics [n,4], camera_extrinsics [n,4,4] (OpenCV c2w),
    frames_color [n,3,H,W] in [0,1], frames_depth [n,1,H,W] clipped to [0,depth_clip],
    frames_normal [n,3,H,W]."""
    intr = replica_intrinsics(height, width)
    c2w = orbit_poses_opencv(num_keyframes, device=device)
    c2w[:, :3, 3] *= scene_scale
    color, depth, normal = render_room(c2w, height, width, intr, half_extent=2.0 * scene_scale)
    return {
        "camera_intrinsics": torch.tensor(intr, dtype=torch.float32, device=device).repeat(num_keyframes, 1),
        "camera_extrinsics": c2w,
        "frames_color": color,
        "frames_depth": depth.clamp(0.0, depth_clip),
        "frames_normal": normal,
    }


class SyntheticEvaluationDataset:
    """The analytic room behind the dataset interface the reference's evaluation code reads
    (/root/reference/evaluation/renderer.py:66-76,81,239-258; evaluator.py:95-98): ground-truth poses in the
    standard (OpenCV) convention, ``camera_intrinsics`` as a dict with ``depth_scale``, ``evaluation_frames``
    (every frame that is not a keyframe stride), and ``frames_color`` / ``frames_depth`` per mode."""

    def __init__(self, num_frames: int = 96, height: int = 120, width: int = 160, evaluation_stride: int = 5,
                 depth_scale: float = 6553.5, device="cpu", scene_scale: float = 1.0):
        self.device = torch.device(device)  # where the ground-truth frames are ray-cast (values are the same)
        fx, fy, cx, cy = replica_intrinsics(height, width)
        self.camera_intrinsics = {"fx": fx, "fy": fy, "cx": cx, "cy": cy, "height": height, "width": width,
                                  "depth_scale": depth_scale}
        self.num_frames = num_frames
        # scene_scale shrinks room and orbit together (the occupancy-grid back-end takes poses as they come: the room has
        # to lie inside its scene box [-1.5, 2.5]^3)
        self.scene_scale = float(scene_scale)
        poses = orbit_poses_opencv(num_frames)
        poses[:, :3, 3] *= self.scene_scale
        self.camera_extrinsics = poses.double().numpy()
        self.evaluation_frames = list(range(2, num_frames, evaluation_stride))

    def render(self, pose_cv) -> tuple:
        """(uint8 [H,W,3] colour, float [H,W] z-depth) of the room from a standard-convention c2w pose."""
        ci = self.camera_intrinsics
        pose = torch.as_tensor(pose_cv, dtype=torch.float32)[None].to(self.device)
        color, depth, _ = render_room(pose, ci["height"], ci["width"], (ci["fx"], ci["fy"], ci["cx"], ci["cy"]),
                                      half_extent=2.0 * self.scene_scale)
        return ((color[0].permute(1, 2, 0).cpu().numpy() * 255).astype("uint8"), depth[0, 0].double().cpu().numpy())

    def _indices(self, mode: str, keyframes) -> list:
        if mode == "keyframes":
            return list(keyframes)
        if mode == "evaluation_frames":
            return list(self.evaluation_frames)
        return list(range(self.num_frames))

    def frames_color(self, mode: str = "evaluation_frames", keyframes=None) -> list:
        return [self.render(self.camera_extrinsics[i])[0] for i in self._indices(mode, keyframes)]

    def frames_depth(self, mode: str = "evaluation_frames", keyframes=None) -> list:
        return [self.render(self.camera_extrinsics[i])[1] for i in self._indices(mode, keyframes)]
