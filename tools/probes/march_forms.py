"""Wave-per-ray against ray-per-lane march on inference-sized launches (NVO_OCC_MARCH_LANES=1 selects the second form):
a 1200x680 view of the bench scene after 600 steps, every ray marched to the end (1024 samples at most), in launches of
--rays rays.  python tools/probes/march_forms.py --rays 65536"""
import argparse
import ctypes as C
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=65536)
    a = ap.parse_args()
    import nerf_vo_amd.pyngp as pyngp
    from nerf_vo_amd import _lib
    from nerf_vo_amd.engine import _call
    from nerf_vo_amd.mapping.cameras import Cameras, CameraType
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.synthetic import make_sequence
    from nerf_vo_amd.tinycudann.modules import _ptr, _stream

    dev = torch.device("cuda:0")
    H, W, K = 272, 480, 48
    seq = make_sequence(K, H, W, device=dev, scene_scale=0.2)
    poses = seq["camera_extrinsics"].clone()
    poses[:, :3, 3] += 0.5
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf, 0)
    tb.create_empty_nerf_dataset(n_images=K, nerf_scale=1.0, nerf_offset=np.zeros(3), aabb_scale=4)
    tb.reload_network_from_file("")
    tb.shall_train = True
    color = seq["frames_color"].permute(0, 2, 3, 1)
    color = torch.cat([color, torch.ones_like(color[..., :1])], dim=3)
    depth = seq["frames_depth"].permute(0, 2, 3, 1)
    tb.nerf.training.update_training_images(
        frame_ids=list(range(K)), poses=opencv_to_opengl(poses)[:, :3], images=color.contiguous(), depths=depth.contiguous(),
        depths_cov=torch.ones_like(depth), resolution=np.array([W, H]),
        principal_point=seq["camera_intrinsics"][0, 2:].cpu().numpy(), focal_length=seq["camera_intrinsics"][0, :2].cpu().numpy())
    for _ in range(600):
        tb.frame()
    eng = tb._engine
    fx = float(seq["camera_intrinsics"][0, 0]) * 1200.0 / W
    c2w = opencv_to_opengl(poses)[3:4, :3, :4].contiguous()
    cams = Cameras(fx=fx, fy=fx, cx=600.0, cy=340.0, height=680, width=1200, camera_to_worlds=c2w,
                   camera_type=CameraType.PERSPECTIVE).to(dev)
    b = cams.generate_rays(camera_indices=0, keep_shape=True)
    o, d = b.origins.reshape(-1, 3).contiguous(), b.directions.reshape(-1, 3).contiguous()
    N, R = o.shape[0], a.rays
    lib = _lib.lib()
    scratch = torch.empty(int(lib.nvo_occ_march_scratch_bytes(R)), dtype=torch.uint8, device=dev)
    counts = torch.zeros(N, dtype=torch.int32, device=dev)
    stream = _stream(dev)
    cfg = eng.cfg

    def frame():
        for lo in range(0, N, R):
            n = min(R, N - lo)
            _call("nvo_occ_march_runs", stream, n, C.c_void_p(o.data_ptr() + 12 * lo), C.c_void_p(d.data_ptr() + 12 * lo),
                  _ptr(eng.bitfield), cfg.n_levels, cfg.cone_angle, cfg.near_distance, None,
                  C.c_void_p(counts.data_ptr() + 4 * lo), _ptr(scratch), scratch.numel(), None, 1024, None, None, 0)

    frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        frame()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"{N} rays in launches of {R}: {ms:.2f} ms per image, {int(counts.sum())} samples found, "
          f"{float(counts.float().mean()):.1f} per ray (max {int(counts.max())})")


if __name__ == "__main__":
    main()
