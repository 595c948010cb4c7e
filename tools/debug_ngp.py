import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry; entry.build()
from nerf_vo_amd.mapping.dataset import opencv_to_opengl
from nerf_vo_amd.ngp_engine import NgpConfig, NgpEngine
from nerf_vo_amd.synthetic import make_sequence
device = torch.device("cuda:0")
n, H, W = 8, 60, 80
seq = make_sequence(n, H, W, device=device, scene_scale=0.2)
c2w = opencv_to_opengl(seq["camera_extrinsics"]); c2w[:, :3, 3] += 0.5; c2w = c2w[:, :3, :4].contiguous()
images = seq["frames_color"].permute(0, 2, 3, 1).contiguous(); depths = seq["frames_depth"].permute(0, 2, 3, 1).contiguous()
eng = NgpEngine(NgpConfig(num_images=n, num_rays=512, capacity=1 << 18), device)
scale = torch.tensor([n, H, W], device=device)
p0 = eng.params.clone()
for it in range(400):
    idx = torch.floor(torch.rand(512, 3, device=device) * scale).long()
    eng.train_step(idx, seq["camera_intrinsics"], c2w, images, depths)
    if it % 20 == 0:
        ws = eng._ws
        print(it, eng.loss_dict(), "samples", int(ws["offsets"][-1]), "dropped rays", int((ws["counts"] == 0).sum()),
              "skip", int(eng.skip_flag), "|dp|", float((eng.params - p0).abs().max()), "|g|", float(eng.grads.abs().max()),
              "occ", float(np.unpackbits(eng.bitfield.cpu().numpy()).mean()), "acc", float(ws["out_accumulation"].mean()), "depth", float(ws["out_depth"].mean()), "gt", float(ws["gt_depth"].mean()))
