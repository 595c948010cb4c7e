"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/nerfvo_hip.h declares,
and its host-side module objects agree with the oracle's level table (no GPU compute here)."""
import ctypes as C
import json
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _header_functions():
    text = (ROOT / "include" / "nerfvo_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nvo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nerf_vo_amd import _lib

    lib = _lib.lib()
    declared = _header_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/nerfvo_hip.h but not exported"
    # the ctypes table and the header must describe the same set
    assert set(declared) == set(_lib.exported_symbols())


def test_struct_mirrors_match_the_header(tmp_path):
    """Every ctypes.Structure of nerf_vo_amd/_lib.py against the C struct it mirrors: include/nerfvo_hip.h is compiled
    as plain C (gcc -- what a cgo / JNI / ctypes binding of a maintainer would do), and sizeof + every
    field's offsetof must agree.  A field added on one side only would otherwise show up as a wrong kernel argument."""
    import shutil
    import subprocess

    from nerf_vo_amd import _lib

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    mirrors = {}
    for name in dir(_lib):
        cls = getattr(_lib, name)
        if isinstance(cls, type) and issubclass(cls, C.Structure) and cls is not C.Structure:
            m = re.search(r"(nvo_\w+)", cls.__doc__ or "")
            assert m, f"{name}: the docstring names the C struct it mirrors"
            mirrors[name] = (cls, m.group(1))
    text = (ROOT / "include" / "nerfvo_hip.h").read_text()
    declared = set(re.findall(r"^}\s*(nvo_\w+);", text, flags=re.M))
    assert declared == {c for _, c in mirrors.values()}, "every argument struct of the header has a mirror"
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "nerfvo_hip.h"', "int main(void) {"]
    for name, (cls, cname) in sorted(mirrors.items()):
        src.append(f'  printf("{name} %zu", sizeof({cname}));')
        src += [f'  printf(" %zu", offsetof({cname}, {f[0]}));' for f in cls._fields_]
        src.append('  printf("\\n");')
    src += ["  return 0;", "}"]
    (tmp_path / "layout.c").write_text("\n".join(src))
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(tmp_path / "layout.c"), "-o",
                    str(tmp_path / "layout")], check=True, capture_output=True, text=True)
    out = subprocess.run([str(tmp_path / "layout")], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.splitlines():
        name, *nums = line.split()
        cls = mirrors[name][0]
        expect = [C.sizeof(cls)] + [getattr(cls, f[0]).offset for f in cls._fields_]
        assert [int(v) for v in nums] == expect, f"{name} vs {mirrors[name][1]}: [sizeof, offsets...] {nums} != {expect}"
        seen += 1
    assert seen == len(mirrors) >= 12


def _create_encoding(cfg):
    from nerf_vo_amd import _lib

    lib = _lib.lib()
    h = C.c_void_p()
    rc = lib.nvo_create_encoding(3, json.dumps(cfg).encode(), C.byref(h))
    return lib, h, rc


GRID_CONFIGS = {
    "main": dict(n_levels=16, log2_hashmap_size=19, base_resolution=16, max_res=2048),
    "prop0": dict(n_levels=5, log2_hashmap_size=17, base_resolution=16, max_res=128),
    "prop1": dict(n_levels=5, log2_hashmap_size=17, base_resolution=16, max_res=256),
}


def _pls(c):
    return float(np.exp((np.log(c["max_res"]) - np.log(c["base_resolution"])) / (c["n_levels"] - 1)))


@pytest.mark.parametrize("name", list(GRID_CONFIGS))
def test_grid_level_table_matches_oracle(name):
    from oracle import grid as G

    c = GRID_CONFIGS[name]
    cfg = {"otype": "HashGrid", "n_levels": c["n_levels"], "n_features_per_level": 2,
           "log2_hashmap_size": c["log2_hashmap_size"], "base_resolution": c["base_resolution"],
           "per_level_scale": _pls(c)}
    lib, h, rc = _create_encoding(cfg)
    assert rc == 0, lib.nvo_last_error()
    spec = G.make_grid_spec(c["n_levels"], 2, c["log2_hashmap_size"], c["base_resolution"], _pls(c))
    lv = np.zeros((c["n_levels"], 4), np.uint32)
    sc = np.zeros(c["n_levels"], np.float32)
    assert lib.nvo_grid_describe(h, lv.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)) == 0
    assert (lv == spec.levels).all()
    assert (sc.view(np.uint32) == spec.scales.view(np.uint32)).all(), "scales must agree bit for bit"
    assert lib.nvo_n_params(h) == spec.n_params
    assert lib.nvo_n_output_dims(h) == 2 * c["n_levels"]
    lib.nvo_destroy(h)


def test_param_counts_match_survey():
    """main grid 12 196 240 params, proposal grids 766 528 / 860 160."""
    from oracle import grid as G

    counts = {k: G.make_grid_spec(c["n_levels"], 2, c["log2_hashmap_size"], c["base_resolution"], _pls(c)).n_params
              for k, c in GRID_CONFIGS.items()}
    assert counts == {"main": 12196240, "prop0": 766528, "prop1": 860160}


def test_errors_are_reported_not_swallowed():
    lib, h, rc = _create_encoding({"otype": "Frequency", "n_frequencies": 4})
    assert rc != 0
    assert b"unsupported" in lib.nvo_last_error()
    from nerf_vo_amd import _lib

    h = C.c_void_p()
    rc = lib.nvo_create_network(7, 3, json.dumps({"otype": "FullyFusedMLP", "n_neurons": 128,
                                                  "n_hidden_layers": 2}).encode(), C.byref(h))
    assert rc != 0 and b"no gfx950 kernel instance" in lib.nvo_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "create_network")


def test_network_param_layout():
    from nerf_vo_amd import _lib
    from oracle import mlp as M

    lib = _lib.lib()
    for (n_in, n_out, width, n_hidden) in [(32, 16, 64, 1), (63, 3, 64, 2), (27, 64, 64, 3), (10, 1, 16, 1)]:
        h = C.c_void_p()
        cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
               "n_neurons": width, "n_hidden_layers": n_hidden}
        assert lib.nvo_create_network(n_in, n_out, json.dumps(cfg).encode(), C.byref(h)) == 0, lib.nvo_last_error()
        assert lib.nvo_n_params(h) == M.mlp_n_params(n_in, n_out, width, n_hidden)
        assert lib.nvo_padded_output_dims(h) == M.pad16(n_out)
        init = np.zeros(lib.nvo_n_params(h), np.float32)
        assert lib.nvo_initial_params(h, 1337, init.ctypes.data_as(C.c_void_p)) == 0
        assert np.isfinite(init).all() and init.std() > 0
        lib.nvo_destroy(h)


def test_cpu_tensor_is_rejected_loudly():
    import torch

    import nerf_vo_amd.tinycudann as tcnn

    enc = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.rand(4, 3))
