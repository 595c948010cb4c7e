"""Build recipe for libnerfvo_hip.so (gfx950 only).

`python nerf-vo_amd/build.py` (or `__graft_entry__.build()`) compiles every csrc/*.hip with
hipcc --offload-arch=gfx950 and links ONE shared library in-tree at nerf-vo_amd/lib/, so that the
built artefact travels with the repo snapshot to the GPU box.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_DIR = PKG_DIR / "lib"
OBJ_DIR = LIB_DIR / "obj"
LIB_PATH = LIB_DIR / "libnerfvo_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-Wno-unused-result", "-ffp-contract=off"]
CXXFLAGS += os.environ.get("NVO_EXTRA_CXXFLAGS", "").split()  # e.g. -DNVO_MLP_PHASE (per-phase cycle counters, debugging)


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _headers() -> list[Path]:
    return sorted(CSRC.glob("*.h")) + sorted((PKG_DIR.parent / "include").glob("*.h"))


def _stamp(src: Path) -> str:
    h = hashlib.sha256()
    h.update(src.read_bytes())
    for hdr in _headers():
        h.update(hdr.read_bytes())
    h.update(" ".join(CXXFLAGS).encode())
    return h.hexdigest()


def _compile(src: Path) -> Path:
    obj = OBJ_DIR / (src.stem + ".o")
    stamp_file = OBJ_DIR / (src.stem + ".stamp")
    stamp = _stamp(src)
    if obj.exists() and stamp_file.exists() and stamp_file.read_text() == stamp:
        return obj
    tmp = obj.with_name(f"{obj.name}.{os.getpid()}.tmp")  # concurrent builders (multi-process tests) never see a torn file
    cmd = [HIPCC, *CXXFLAGS, "-c", str(src), "-o", str(tmp)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        tmp.unlink(missing_ok=True)
        raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
    if res.stderr.strip():
        sys.stderr.write(res.stderr)
    os.replace(tmp, obj)
    stamp_file.write_text(stamp)
    return obj


def build(force: bool = False, verbose: bool = True) -> Path:
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    if force:
        for f in OBJ_DIR.glob("*.stamp"):
            f.unlink()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    newest_obj = max(o.stat().st_mtime for o in objs)
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < newest_obj:
        tmp = LIB_PATH.with_name(f"{LIB_PATH.name}.{os.getpid()}.tmp")
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(tmp)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            tmp.unlink(missing_ok=True)
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
        os.replace(tmp, LIB_PATH)  # atomic: a process that already mapped the old library keeps its inode
    if verbose:
        # stderr: bench.py must print exactly one JSON line on stdout
        print(f"[nerf-vo_amd] built {LIB_PATH} ({LIB_PATH.stat().st_size / 1e6:.1f} MB) from {len(srcs)} sources",
              file=sys.stderr)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
