import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import mapping_loop as ml
# run the loop, then inspect the last step's per-sample gradients of the main field
import argparse
orig = ml.run
res = None
import nerf_vo_amd.engine as E
captured = {}
_old = E.NerfactoEngine.train_step_graphed
def hook(self, *a, **k):
    captured["eng"] = self
    return _old(self, *a, **k)
E.NerfactoEngine.train_step_graphed = hook
r = ml.run(profile_steps=0, render_frames=0)
eng = captured["eng"]
ws = eng._workspace(eng.cfg.num_rays, True)
torch.cuda.synchronize()
drgb = ws["drgb"].float()[:, :3]
dpre = ws["dout2"].float()  # [N,16]: d_base_out incl. dpre in col 0
dead_s = (drgb.abs().sum(1) == 0) & (dpre.abs().sum(1) == 0)
N = dead_s.numel()
tiles = dead_s.view(-1, 16).all(dim=1)
print("loss scale", eng.current_loss_scale(), "dead samples", float(dead_s.float().mean()), "dead 16-tiles", float(tiles.float().mean()),
      "drgb zero", float((drgb.abs().sum(1) == 0).float().mean()), "dpre rows zero", float((dpre.abs().sum(1) == 0).float().mean()))
w = ws["weights2"].view(-1, 48)
print("weights: mean samples per ray with w>1e-3:", float((w > 1e-3).float().sum(1).mean()), " w>1e-5:", float((w > 1e-5).float().sum(1).mean()))
