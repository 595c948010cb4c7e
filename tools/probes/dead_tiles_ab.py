"""Same-state A/B of EngineConfig.sparse_backward (the live-tile lists of the main field's MLP backwards) and of the hash
grid's live-row list (NVO_GRID_LIVE_ROWS): run the mapping loop once,
then time blocks of graph-replayed steps on the trained field with the step graphs re-captured under either setting,
alternating, in one process.  Prints ms/step per block and the loss scale (which sets how much of dL/d(rgb) underflows)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mapping_loop as ml  # noqa: E402
import nerf_vo_amd.engine as E  # noqa: E402

captured = {}
_old = E.NerfactoEngine.train_step_graphed


def hook(self, dataset, *a, **k):
    captured["eng"], captured["ds"] = self, dataset
    return _old(self, dataset, *a, **k)


E.NerfactoEngine.train_step_graphed = hook
ml.run(profile_steps=0, render_frames=0, window_starts=())
eng, ds = captured["eng"], captured["ds"]
E.NerfactoEngine.train_step_graphed = _old


GRAPHS = {}
SCALE = float(os.environ.get("NVO_AB_LOSS_SCALE", "0"))  # > 0: pin the GradScaler's scale for the blocks (64 = the underflow regime)


def block(sparse: bool, rows: bool, steps: int = 300) -> float:
    if SCALE > 0:
        eng.dev_loss_scale.fill_(SCALE)
        eng.dev_growth_tracker.zero_()
    eng.cfg.sparse_backward = "on" if sparse else "off"  # (EngineConfig.sparse_backward: which kind of step graph runs)
    if rows:
        os.environ.pop("NVO_GRID_LIVE_ROWS", None)
    else:
        os.environ["NVO_GRID_LIVE_ROWS"] = "0"  # (read when a graph is captured: the MLP backwards keep their tile lists)
    # one set of step graphs per setting, all kept alive (dropping the graphs of a setting and capturing again crashed a
    # later replay -- tools/probes/recapture_repro.py; not a flow the mapper has)
    eng._graphs = GRAPHS.setdefault((sparse, rows), {})
    for _ in range(12):  # warm-up: eager step + captures of the variants the schedule uses here
        eng.train_step_graphed(ds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step_graphed(ds)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


# (sparse steps, grid live-row list)
ORDER = [(c[0] == "1", c[1:2] != "0") for c in os.environ.get("NVO_AB_ORDER", "11,10,00").split(",")]
for rep in range(int(os.environ.get("NVO_AB_REPS", "3"))):
    for sparse, rows in ORDER:
        ms = block(sparse, rows)
        ws = eng._workspace(eng.cfg.num_rays, True)
        live = ws["tile_live"]
        print(f"rep {rep} sparse steps {'on ' if sparse else 'off'} grid live-row list {'on ' if rows else 'off'}: {ms:.4f} ms/step  loss scale {eng.current_loss_scale():g}  "
              f"tiles with rgb gradient {float((live & 1).bool().float().mean()):.3f}  with any {float((live != 0).float().mean()):.3f}", flush=True)
