"""GPU: HIP kernels (through the C-ABI) against the COMMITTED golden fixtures tests/golden/oracle_golden.npz.
Nothing here reads /root/reference or imports the oracle package for expected values."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GRIDS = {"main": (16, 19, 16, 2048), "prop0": (5, 17, 16, 128), "prop1": (5, 17, 16, 256)}


def _pls(b, m, L):
    return float(np.exp((np.log(m) - np.log(b)) / (L - 1)))


def _points(n, seed):
    x = np.random.default_rng(seed).random((n, 3), dtype=np.float32)
    x[0], x[1], x[2] = 0.0, 1.0, 0.5
    return x


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(os.path.join(HERE, "golden", "oracle_golden.npz")))


@pytest.mark.parametrize("name", list(GRIDS))
def test_level_table_and_indices_bit_exact(device, golden, name):
    import nerf_vo_amd.tinycudann as tcnn
    from nerf_vo_amd import _lib

    L, T, b, m = GRIDS[name]
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": L, "n_features_per_level": 2, "log2_hashmap_size": T,
                            "base_resolution": b, "per_level_scale": _pls(b, m, L)})
    lib = _lib.lib()
    lv = np.zeros((L, 4), np.uint32)
    sc = np.zeros(L, np.float32)
    lib.nvo_grid_describe(enc.native_tcnn_module.handle, lv.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p))
    assert (lv == golden[f"g1_{name}_levels"]).all()
    assert (sc.view(np.uint32) == golden[f"g1_{name}_scales"].view(np.uint32)).all()
    x = torch.from_numpy(_points(64, 11)).to(device)
    idx = torch.zeros((L, 64, 8), dtype=torch.int32, device=device)
    _lib.check(lib.nvo_grid_indices(enc.native_tcnn_module.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                    64, C.c_void_p(x.data_ptr()), C.c_void_p(idx.data_ptr())), "grid_indices")
    assert (idx.cpu().numpy().view(np.uint32) == golden[f"g2_{name}_indices"]).all()


def test_encoded_features_match_golden(device, golden):
    import nerf_vo_amd.tinycudann as tcnn

    L, T, b, m = GRIDS["prop0"]
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": L, "n_features_per_level": 2, "log2_hashmap_size": T,
                            "base_resolution": b, "per_level_scale": _pls(b, m, L)}).to(device)
    table = np.random.default_rng(5).uniform(-1, 1, (enc.params.numel() // 2, 2))
    with torch.no_grad():
        enc.params.copy_(torch.from_numpy(table.reshape(-1)).float().to(device))
    with torch.no_grad():
        y = enc(torch.from_numpy(_points(64, 12)).to(device)).float().cpu().numpy()
    # the golden features use the un-rounded float64 table; the kernel reads its fp16 copy: |err| <= ~1e-3
    np.testing.assert_allclose(y, golden["g3_features"], atol=2e-3, rtol=2e-3)


def test_sh_and_se3_match_golden(device, golden):
    import nerf_vo_amd.tinycudann as tcnn
    from nerf_vo_amd import _lib

    enc = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4}).to(device)
    d = torch.from_numpy(golden["g5_dirs"]).float().to(device)
    with torch.no_grad():
        y = enc((d + 1) / 2).float().cpu().numpy()
    np.testing.assert_allclose(y, golden["g5_sh4"], atol=2e-3, rtol=2e-3)

    tang = torch.from_numpy(golden["g9_tangent"]).float().to(device).contiguous()
    out = torch.empty(tang.shape[0], 3, 4, device=device)
    _lib.check(_lib.lib().nvo_se3_exp_map(C.c_void_p(torch.cuda.current_stream().cuda_stream), tang.shape[0],
                                          C.c_void_p(tang.data_ptr()), C.c_void_p(out.data_ptr())), "se3")
    np.testing.assert_allclose(out.cpu().numpy(), golden["g9_exp"], atol=2e-6, rtol=1e-5)


def test_lindisp_bins_match_golden(device, golden):
    from nerf_vo_amd import _lib

    jit = torch.from_numpy(golden["g6_jitter"]).float().reshape(-1).to(device).contiguous()
    sb = torch.empty(8, 257, device=device)
    tb = torch.empty(8, 257, device=device)
    _lib.check(_lib.lib().nvo_sample_lindisp(C.c_void_p(torch.cuda.current_stream().cuda_stream), 8, 256, 0.05, 1000.0,
                                             C.c_void_p(jit.data_ptr()), C.c_void_p(sb.data_ptr()),
                                             C.c_void_p(tb.data_ptr())), "lindisp")
    np.testing.assert_allclose(sb.cpu().numpy(), golden["g6_lindisp_sbins"], atol=1e-6)
    np.testing.assert_allclose(tb.cpu().numpy(), golden["g6_lindisp_tbins"], rtol=2e-3)
