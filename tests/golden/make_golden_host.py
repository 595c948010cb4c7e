#!/usr/bin/env python3
"""Golden vectors for host logic either side of the hot path, produced BY THE REFERENCE (imports
/root/reference; pure Python, runs on CPU):
  * MappingModule.step call cadence (nerf_vo/mapping/mapping_module.py:35-55) for a scripted queue schedule
  * scale_camera_intrinsics (nerf_vo/data/data_utils.py:24-34) on datasets/replica.json for the two BASELINE
    resolutions
Writes tests/golden/host_golden.json.   python tests/golden/make_golden_host.py
"""
import argparse
import json
import os
import sys
import threading

sys.path.insert(0, "/root/reference")
from nerf_vo.data.data_utils import scale_camera_intrinsics  # noqa: E402
from nerf_vo.mapping.mapping_module import MappingModule  # noqa: E402


class _FakeMethod:
    is_initialized = True

    def __init__(self, shut_down_after):
        self.calls = []
        self.is_shut_down = False
        self.shut_down_after = shut_down_after

    def __call__(self, input):
        self.calls.append(input is not None)
        if len(self.calls) >= self.shut_down_after:
            self.is_shut_down = True


def cadence(schedule, mapping_iterations, num_keyframes, shut_down_after):
    mod = object.__new__(MappingModule)
    mod.args = argparse.Namespace(mapping_iterations=mapping_iterations, num_keyframes=num_keyframes)
    mod.method = _FakeMethod(shut_down_after)
    mod.step_counter = 0
    mod.shutdown = False
    mod.is_receving_data = True
    mod.last_received_data = 0
    mod.shared_variables = {"status_lock": threading.Lock(), "status": {}}
    rows = []
    for item in schedule:
        inp = None if item == 0 else {"last_frame": item == 2}
        n_before = len(mod.method.calls)
        _, skip = mod.step(inp)
        rows.append([item, int(len(mod.method.calls) > n_before), int(skip), int(mod.shutdown)])
    return rows


def build():
    out = {"cadence": []}
    # 0 = empty queue tick, 1 = keyframe batch, 2 = batch carrying last_frame
    schedules = [
        ([1] + [0] * 12 + [1] + [0] * 3 + [1] + [0] * 20 + [2] + [0] * 15, 64, 8, 10 ** 9),
        ([0] * 5 + [1, 1, 0, 0, 0, 0, 0, 0, 2] + [0] * 10, 30, 10, 18),
        ([1] + [0] * 9, 8192, 192, 10 ** 9),
    ]
    for sched, its, kf, stop in schedules:
        out["cadence"].append({"schedule": sched, "mapping_iterations": its, "num_keyframes": kf,
                               "shut_down_after": stop, "rows": cadence(sched, its, kf, stop)})
    native = json.load(open("/root/reference/datasets/replica.json"))["replica"]
    out["intrinsics"] = []
    for h, w in ((480, 640), (360, 640), (680, 1200)):
        cam = {"height": native["h"], "width": native["w"], "fx": native["fx"], "fy": native["fy"],
               "cx": native["cx"], "cy": native["cy"]}
        r = scale_camera_intrinsics(dict(cam), height=h, width=w)
        out["intrinsics"].append({"height": h, "width": w, "fx": r["fx"], "fy": r["fy"], "cx": r["cx"], "cy": r["cy"]})
    # PSNR: evaluation/evaluation_utils.py cannot be imported (needs cv2), but calculate_psnr (:289-306) is plain
    # numpy: its definition is executed from the reference file AT GENERATION TIME (nothing is copied into the
    # repository) and applied per colour channel, which is what calculate_psnr_color (:309-318) does with
    # cv2.split + the mean of the three values.
    import ast

    import numpy as np
    src = open("/root/reference/evaluation/evaluation_utils.py").read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "calculate_psnr")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "evaluation_utils.py", "exec"), ns)
    rng = np.random.default_rng(5)
    out["psnr"] = []
    for shape, spread in (((24, 32, 3), 30), ((17, 9, 3), 3), ((8, 8, 3), 200)):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        b = np.clip(a.astype(np.int32) + rng.integers(-spread, spread + 1, shape), 0, 255).astype(np.uint8)
        with np.errstate(over="ignore"):
            per_channel = [float(ns["calculate_psnr"](a[..., c], b[..., c])) for c in range(3)]
        out["psnr"].append({"a": a.tolist(), "b": b.tolist(), "per_channel": per_channel,
                            "color": sum(per_channel) / 3.0})
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_golden.json")
    json.dump(build(), open(path, "w"), indent=1)
    print("wrote", path)
