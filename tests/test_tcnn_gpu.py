"""GPU parity tests: HIP kernels behind the tcnn-compatible surface vs the CPU oracle.

Tolerances (stated per north_star: fp32 tolerance for outputs, bit-exact for hash indices):
  * hash / dense corner indices: bit-exact (uint32).
  * encoded features: kernel accumulates in fp32 and rounds once to fp16 -> |err| <= 1 fp16 ulp of
    the value + tiny fp32 accumulation slack, tested as atol 2e-3*scale.
  * MLP outputs: fp16 operands, fp32 accumulate, fp16 hidden storage; the oracle rounds at the same
    points in float64 -> rtol 1e-2, atol 1e-2*scale (a few fp16 ulps amplified through <= 4 layers).
  * gradients: fp32 accumulation of fp16 products over the batch -> rtol 2e-2 / atol 2e-2*scale.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MAIN = dict(n_levels=16, log2_hashmap_size=19, base_resolution=16, max_res=2048)
PROP0 = dict(n_levels=5, log2_hashmap_size=17, base_resolution=16, max_res=128)
PROP1 = dict(n_levels=5, log2_hashmap_size=17, base_resolution=16, max_res=256)


def _pls(c):
    return float(np.exp((np.log(c["max_res"]) - np.log(c["base_resolution"])) / (c["n_levels"] - 1)))


def _enc_cfg(c):
    return {"otype": "HashGrid", "n_levels": c["n_levels"], "n_features_per_level": 2,
            "log2_hashmap_size": c["log2_hashmap_size"], "base_resolution": c["base_resolution"],
            "per_level_scale": _pls(c)}


def _spec(c):
    from oracle import grid as G

    return G.make_grid_spec(c["n_levels"], 2, c["log2_hashmap_size"], c["base_resolution"], _pls(c))


def _points(n, seed, edge_cases=True):
    rng = np.random.default_rng(seed)
    x = rng.random((n, 3), dtype=np.float32)
    if edge_cases and n >= 16:
        x[0] = 0.0
        x[1] = 1.0
        x[2] = (0.5, 0.5, 0.5)
        x[3] = (1.0, 0.0, 1.0)
        x[4] = np.float32(1.0) - np.float32(2.0 ** -24)
        x[5] = np.float32(2.0 ** -24)
        x[6] = (0.25, 0.75, 0.125)
    return x


# bf16 keeps 8 significant bits where fp16 keeps 11: one rounding step is 2^(11-8) = 8x wider, and every tolerance of a
# test that runs in both formats is multiplied by this ONE factor in bf16 mode
BF16_K = 8.0


MARGINS = []  # one record per _assert_close call of the session


def _assert_close(got, ref, rtol, atol_scale, what, max_outlier_frac=0.0, max_outlier=0.05):
    """|got-ref| <= atol_scale*max|ref| + rtol*|ref| elementwise.  max_outlier_frac > 0 is only used
    for gradients that pass through ReLU kinks: a hidden unit whose pre-activation is within fp16
    rounding of 0 can take a different branch in the kernel and in the float64 oracle, which changes
    a handful of (sample-local) gradient entries discontinuously; such an entry may be off by at most
    max_outlier * max|ref| (0.05; the bf16 tests allow 0.2 -- 8x wider rounding band around every kink)."""
    got = got.double().cpu()
    ref = ref.double().cpu()
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs()
    bound = atol_scale * scale + rtol * ref.abs()
    bad = err > bound
    # how much of each stated tolerance the comparison actually used (conftest writes the session's table:
    # profiles/*_parity_margins.json)
    MARGINS.append({"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "what": what, "elements": int(bad.numel()),
                    "rtol": rtol, "atol_scale": atol_scale, "worst_error_over_bound": float((err / bound.clamp_min(1e-300)).max()),
                    "rel_l1": float(err.sum() / max(ref.abs().sum().item(), 1e-300)),
                    "outlier_frac_allowed": max_outlier_frac, "outlier_frac": float(bad.sum()) / max(1, bad.numel()),
                    "max_error_over_scale": float(err.max()) / scale})
    if max_outlier_frac > 0 and bad.sum().item() <= max_outlier_frac * bad.numel():
        assert err.max().item() <= max_outlier * scale, f"{what}: outlier too large {err.max().item():.3e} (scale {scale:.3e})"
        return
    assert not bad.any(), (
        f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; max err {err.max().item():.4e} "
        f"(scale {scale:.4e}); first bad idx {bad.nonzero()[0].tolist()} got {got[bad][0].item():.6e} "
        f"ref {ref[bad][0].item():.6e}")


def _assert_close_chain(got, ref, what, rel_l1, cap):
    """Hash-grid gradients that arrive through a 16-bit gradient chain (loss kernels -> fused-MLP backward -> 16-bit
    dL/d(encoded) -> scatter).  No per-entry RELATIVE bound exists for them: dL/d(encoded) = W^T dZ is a sum whose terms
    are rounded to 16 bits one by one, so an entry fed by a few samples whose terms cancel carries an error set by the
    terms, not by the result; and a hidden unit whose pre-activation lies within rounding of 0 takes the other ReLU
    branch in the kernel than in the float64 oracle, which moves the few entries its sample touches by a finite amount.
    What the format does guarantee, and what this asserts -- both without any outlier allowance:
        (a) aggregate accuracy:  sum |got - ref|  <=  rel_l1 x sum |ref|
        (b) no entry is far off: max |got - ref|  <=  cap x max |ref|
    with rel_l1 / cap set to 2-2.5x what the suite measures (profiles/r5_parity_margins.md)."""
    got, ref = got.double().cpu().reshape(-1), ref.double().cpu().reshape(-1)
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs()
    l1 = float(err.sum() / max(ref.abs().sum().item(), 1e-300))
    worst = float(err.max()) / scale
    MARGINS.append({"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "what": what, "elements": int(err.numel()),
                    "rtol": f"relative L1 <= {rel_l1:g}", "atol_scale": f"max error <= {cap:g} x max|ref|",
                    "worst_error_over_bound": max(l1 / rel_l1, worst / cap), "rel_l1": l1, "outlier_frac_allowed": 0.0,
                    "outlier_frac": 0.0, "max_error_over_scale": worst})
    assert l1 <= rel_l1, f"{what}: relative L1 error {l1:.3e} > {rel_l1:g}"
    assert worst <= cap, f"{what}: max error {worst:.3e} x max|ref| > {cap:g} (idx {int(err.argmax())})"


@pytest.mark.parametrize("cfg", [MAIN, PROP0, PROP1], ids=["main", "prop0", "prop1"])
def test_grid_indices_bit_exact(device, cfg):
    import nerf_vo_amd.tinycudann as tcnn
    from nerf_vo_amd import _lib
    from oracle import grid as G

    enc = tcnn.Encoding(3, _enc_cfg(cfg))
    n = 4096
    x = _points(n, 1)
    xd = torch.from_numpy(x).to(device)
    idx = torch.zeros((cfg["n_levels"], n, 8), dtype=torch.int32, device=device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.lib().nvo_grid_indices(enc.native_tcnn_module.handle, stream, n, C.c_void_p(xd.data_ptr()),
                                           C.c_void_p(idx.data_ptr())), "grid_indices")
    torch.cuda.synchronize()
    ref, _ = G.grid_indices_c(_spec(cfg), x)
    got = idx.cpu().numpy().view(np.uint32)
    assert (got == ref).all(), f"{int((got != ref).sum())} corner indices differ"


def _set_bwd_mode(enc, bwd_mode):
    """The three backward forms of the hash grid (+ the run-merging variant of the second, which the proposal grids run):
    0 global atomics (the readable reference form) | 1 slice owner | 5 slice owner with run-merged dense levels |
    7 streamed: tile-local pair records accumulated as two 32-bit fixed-point sums per 64-bit word, 8192-entry bins, the
    coarse levels slice-owner with run merging (what the main grid runs)."""
    m = enc.native_tcnn_module
    m.set_option("grid_bwd_runs", int(bwd_mode in (5, 7)))
    m.set_option("grid_bwd_mode", {5: 1, 7: 3}.get(bwd_mode, bwd_mode))


BWD_MODES = [0, 1, 5, 7]
BWD_MODE_IDS = ["atomic", "lds", "lds-runs", "streamed-packed32"]


@pytest.mark.parametrize("cfg", [MAIN, PROP0], ids=["main", "prop0"])
@pytest.mark.parametrize("bwd_mode", BWD_MODES, ids=BWD_MODE_IDS)
def test_grid_encoding_fwd_bwd(device, cfg, bwd_mode):
    import nerf_vo_amd.tinycudann as tcnn
    from oracle import grid as G

    spec = _spec(cfg)
    enc = tcnn.Encoding(3, _enc_cfg(cfg)).to(device)
    _set_bwd_mode(enc, bwd_mode)
    assert enc.n_output_dims == spec.n_output_dims and enc.params.numel() == spec.n_params
    g = torch.Generator().manual_seed(5)
    params = (torch.rand(spec.n_params, generator=g) * 2 - 1)
    with torch.no_grad():
        enc.params.copy_(params.to(device))
    n = 1000  # ragged on purpose: padded to 1024 inside the module
    x = torch.from_numpy(_points(n, 2)).to(device).requires_grad_(True)
    y = enc(x)
    assert y.shape == (n, spec.n_output_dims) and y.dtype == torch.float16
    dy = torch.randn(n, spec.n_output_dims, generator=g).to(device)
    (y.float() * dy).sum().backward()
    torch.cuda.synchronize()

    table = params.to(torch.float16).double().view(-1, 2).requires_grad_(True)
    xr = x.detach().double().cpu().requires_grad_(True)
    yr = G.grid_encode(spec, xr, table)
    dy16 = (dy.cpu().float() * 128.0).to(torch.float16).double() / 128.0  # kernel sees fp16(dy*loss_scale)
    (yr * dy16).sum().backward()

    _assert_close(y, yr, rtol=4e-4, atol_scale=4e-4, what="encoded features")
    _assert_close(enc.params.grad, table.grad.reshape(-1), rtol=1e-3, atol_scale=1e-4, what="dL/dparams")
    _assert_close(x.grad, xr.grad, rtol=1.1e-3, atol_scale=1.1e-4, what="dL/dx")


@pytest.mark.parametrize("cfg", [MAIN, PROP0], ids=["main", "prop0"])
@pytest.mark.parametrize("bwd_mode", [1, 3], ids=["lds", "streamed"])
def test_grid_bwd_32bit_accumulators(device, cfg, bwd_mode):
    """grid_acc_bits=32: int32 accumulators with the overflow-proof L1-derived scale (resolution L1 / 2^29 per
    level and feature).  Same oracle comparison as the 64-bit form with the absolute tolerance widened to that
    resolution x sqrt(contributions): atol 2e-4 * max|ref|; the result is reproducible bit for bit."""
    import nerf_vo_amd.tinycudann as tcnn
    from oracle import grid as G

    spec = _spec(cfg)
    enc = tcnn.Encoding(3, _enc_cfg(cfg)).to(device)
    enc.native_tcnn_module.set_option("grid_acc_bits", 32)
    enc.native_tcnn_module.set_option("grid_bwd_mode", bwd_mode)
    g = torch.Generator().manual_seed(5)
    params = (torch.rand(spec.n_params, generator=g) * 2 - 1)
    with torch.no_grad():
        enc.params.copy_(params.to(device))
    n = 4096
    x = torch.from_numpy(_points(n, 2)).to(device)
    dy = torch.randn(n, spec.n_output_dims, generator=g).to(device) * torch.logspace(-3, 1, spec.n_output_dims)[None].to(device)
    grads = []
    for _ in range(2):
        enc.params.grad = None
        (enc(x).float() * dy).sum().backward()
        grads.append(enc.params.grad.clone())
    torch.cuda.synchronize()
    if bwd_mode == 1:
        assert torch.equal(grads[0], grads[1]) or (grads[0] - grads[1]).abs().max() <= 1e-6 * grads[0].abs().max()
    table = params.to(torch.float16).double().view(-1, 2).requires_grad_(True)
    yr = G.grid_encode(spec, x.double().cpu(), table)
    dy16 = (dy.cpu().float() * 128.0).to(torch.float16).double() / 128.0
    (yr * dy16).sum().backward()
    ref = table.grad.reshape(-1)
    # per level: the resolution scales with that level's own L1, so compare level by level
    for l in range(spec.n_levels):
        lo, hi = 2 * int(spec.levels[l, 0]), 2 * int(spec.levels[l, 0] + spec.levels[l, 1])
        _assert_close(grads[0][lo:hi], ref[lo:hi], rtol=4.7e-4, atol_scale=9.4e-5, what=f"dL/dparams level {l} (32-bit)")


@pytest.mark.parametrize("share", [60, 120, 250])
def test_grid_bwd_item_table_shares_do_not_change_the_gradient(device, share):
    """grid_bwd_dense_share only redistributes the chunks of the slice-owner item table between dense and hashed slices
    (what the engine sets to 120 where most proposal samples are live): the gradient must stay the one of the even
    table -- same integer accumulators per slice, chunks combined with float atomics (rtol 1e-4, atol 1e-6 x max)."""
    import nerf_vo_amd.tinycudann as tcnn

    n = 4096 * 24
    g = torch.Generator().manual_seed(13)
    x = torch.rand(n, 3, generator=g).to(device)
    dy = torch.randn(n, 2 * PROP0["n_levels"], generator=g).to(device)
    grads = []
    for pct in (100, share):
        enc = tcnn.Encoding(3, _enc_cfg(PROP0)).to(device)
        m = enc.native_tcnn_module
        m.set_option("grid_acc_bits", 32)
        m.set_option("grid_bwd_runs", 1)
        m.set_option("grid_bwd_batch", n)
        m.set_option("grid_bwd_dense_share", pct)
        m.set_option("grid_bwd_mode", 1)
        with torch.no_grad():
            enc.params.copy_(torch.linspace(-1, 1, enc.params.numel(), device=device))
        (enc(x).float() * dy).sum().backward()
        grads.append(enc.params.grad.clone())
    _assert_close(grads[1], grads[0], rtol=3e-5, atol_scale=3e-7, what=f"dense share {share} % vs even item table")
    assert float(grads[0].abs().max()) > 0


@pytest.mark.parametrize("bad_value", [float("inf"), float("-inf"), float("nan")], ids=["inf", "-inf", "nan"])
@pytest.mark.parametrize("cfg", [MAIN, PROP0], ids=["main", "prop0"])
def test_grid_bwd_propagates_nonfinite(device, cfg, bad_value):
    """A non-finite dy (fp16 overflow of the loss-scaled gradient) must surface as a non-finite parameter gradient in
    EVERY scatter form, so that the optimiser's GradScaler-style check skips the step -- integer LDS accumulators
    cannot carry inf / NaN by themselves (regression: they turned it into a finite garbage update)."""
    import nerf_vo_amd.tinycudann as tcnn

    enc = tcnn.Encoding(3, _enc_cfg(cfg)).to(device)
    spec = _spec(cfg)
    n = 20000
    g = torch.Generator().manual_seed(13)
    x = torch.rand(n, 3, generator=g).to(device)
    for level in (0, spec.n_levels - 1):
        dy = torch.randn(n, 2 * spec.n_levels, generator=g).to(device)
        dy[12345, 2 * level + 1] = bad_value
        for mode, bits in ((0, 64), (1, 64), (1, 32), (3, 64), (3, 32)):
            enc.native_tcnn_module.set_option("grid_bwd_mode", mode)
            enc.native_tcnn_module.set_option("grid_acc_bits", bits)
            enc.params.grad = None
            (enc(x).float() * dy).sum().backward()
            grad = enc.params.grad
            lo, cnt = int(spec.levels[level, 0]), int(spec.levels[level, 1])
            assert not bool(torch.isfinite(grad[2 * lo:2 * (lo + cnt)]).all()), \
                f"mode {mode} ({bits}-bit): non-finite dy at level {level} vanished from the gradient"
            # the other levels are untouched by the poison
            other = torch.cat([grad[:2 * lo], grad[2 * (lo + cnt):]])
            assert bool(torch.isfinite(other).all())
    enc.native_tcnn_module.set_option("grid_acc_bits", 64)


@pytest.mark.parametrize("kind", ["encoding", "network_with_input_encoding"])
@pytest.mark.parametrize("cfg", [MAIN, PROP1], ids=["main", "prop1"])
def test_stored_input_gradients_match_gather(device, cfg, kind):
    """Option prepare_input_gradients (tcnn's forward flag): the forward stores d(encoded)/d(position) as fp16 in cell
    units and the input backward streams it.  Same outputs bit for bit, same parameter gradients; the input gradient
    equals the gather form up to the fp16 rounding of the stored derivative (2^-11 per level term; the gather form
    itself is checked against the oracle in test_grid_encoding_fwd_bwd): atol 2e-3 * max|dx|, rtol 2e-3."""
    import nerf_vo_amd.tinycudann as tcnn

    if kind == "encoding":
        m = tcnn.Encoding(3, _enc_cfg(cfg)).to(device)
        n_out = 2 * cfg["n_levels"]
    else:
        m = tcnn.NetworkWithInputEncoding(3, 16, _enc_cfg(cfg), {"otype": "FullyFusedMLP", "activation": "ReLU",
                                                                 "output_activation": "None", "n_neurons": 64,
                                                                 "n_hidden_layers": 1}).to(device)
        n_out = 16
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        m.params.copy_(((torch.rand(m.params.numel(), generator=g) - 0.5) * 0.8).to(device))
    n = 30000 + 37  # ragged
    x0 = torch.rand(n, 3, generator=g)
    x0[:4] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [1.0, 0.0, 0.25]])
    dy = torch.randn(n, n_out, generator=g).to(device)
    res = []
    for stored in (0, 1, 0):
        m.native_tcnn_module.set_option("prepare_input_gradients", stored)
        x = x0.to(device).requires_grad_(True)
        m.params.grad = None
        y = m(x)
        (y.float() * dy).sum().backward()
        res.append((y.detach().clone(), x.grad.clone(), m.params.grad.clone()))
    (y0, dx0, dp0), (y1, dx1, dp1), (y2, dx2, _) = res
    assert torch.equal(y0, y1), "the option must not change the forward output"
    # (parameter gradients: same kernels; the default slice-owner scatter is reproducible to fp32 rounding only)
    _assert_close(dp1, dp0, rtol=1e-5, atol_scale=2e-6, what="dL/dparams with the option on")
    assert torch.equal(dx0, dx2), "switching the option off again must restore the gather form exactly"
    _assert_close(dx1, dx0, rtol=4.8e-4, atol_scale=4.8e-4, what="dL/dx from stored dy/dx vs gather")
    assert float(dx0.abs().max()) > 0


def test_grid_bwd_lds_matches_atomic_large(device):
    """Both scatter forms at the full main-field batch (196 608 samples): linearity + agreement."""
    import nerf_vo_amd.tinycudann as tcnn

    enc = tcnn.Encoding(3, _enc_cfg(MAIN)).to(device)
    n = 4096 * 48
    g = torch.Generator().manual_seed(7)
    x = torch.rand(n, 3, generator=g).to(device)
    dy = torch.randn(n, 32, generator=g).to(device)
    grads = {}
    for mode in BWD_MODES:
        _set_bwd_mode(enc, mode)
        enc.params.grad = None
        y = enc(x)
        (y.float() * dy).sum().backward()
        grads[mode] = enc.params.grad.clone()
    torch.cuda.synchronize()
    # packed 32-bit fixed-point accumulators (one 64-bit LDS atomic per record, scale 2^29 / L1 of the bin): each add
    # rounds to L1(bin) / 2^29 -- finer than the 16-17 mantissa bits the records carry -- and the sums are integers
    _assert_close(grads[7], grads[0], rtol=4.1e-4, atol_scale=4.1e-6, what="streamed (tile-local, packed 32-bit) vs atomic")
    _set_bwd_mode(enc, 7)
    enc.params.grad = None
    (enc(x).float() * dy).sum().backward()
    hashed7 = 2 * (4096 + 12168 + 29792 + 79512 + 205384)
    assert torch.equal(enc.params.grad[hashed7:], grads[7][hashed7:]), "packed form is not bitwise reproducible"
    # run-merged coarse levels (uniform random points are the worst case: no two consecutive samples share a cell)
    _assert_close(grads[5], grads[0], rtol=1e-4, atol_scale=1e-6, what="lds + run-merged dense levels vs atomic dL/dparams")
    _assert_close(grads[1], grads[0], rtol=1e-4, atol_scale=1e-6, what="lds vs atomic dL/dparams")
    # every sample distributes a total weight of 1 per level/feature: sum of grads == sum of dy16
    dy16 = (dy * 128).half().double() / 128
    assert abs(grads[1].double().sum().item() - dy16.sum().item()) <= 1e-2 * dy16.abs().sum().item() ** 0.5 + 1.0


def test_pair_records_that_straddle_two_bins(device):
    """The record pass of the packed streamed backward carries both x corners of a (y, z) pair in ONE record because they
    share a bin -- except where they do not: (a) hashed levels finer than the bin size (the occupancy-grid back-end's
    grid reaches resolution 8192: px = 8191 -> px ^ (px + 1) = 0x3FFF, beyond the 8192-entry bin), (b) a dense level's
    bin boundary (index % 8192 == 8191).  Such pairs go out as two single-corner records; this test aims samples AT those
    places and compares with the global-atomic form (rtol 1e-3, atol 1e-5 x max, as test_grid_bwd_lds_matches_atomic_large)."""
    import nerf_vo_amd.tinycudann as tcnn

    ngp = dict(n_levels=16, log2_hashmap_size=19, base_resolution=16, max_res=8192)
    g = torch.Generator().manual_seed(11)
    for cfg, what in ((ngp, "resolution 8192"), (MAIN, "dense bin boundary")):
        enc = tcnn.Encoding(3, _enc_cfg(cfg)).to(device)
        spec = _spec(cfg)
        n = 8192
        x = torch.rand(n, 3, generator=g)
        if cfg is ngp:
            # finest level: cell x index 8191 (scale * x + 0.5 in [8191, 8192))
            scale = float(spec.scales[-1])
            assert abs(scale - 8191.0) < 0.1
            x[: n // 2, 0] = (8191.0 + torch.rand(n // 2, generator=g) * 0.48 + 0.01 - 0.5) / scale  # upper half-cell: x <= 1
            assert float(x[:, 0].max()) <= 1.0
        else:
            # level 4 (dense, res + 1 = 59): cells whose first corner has index % 8192 == 8191
            lvl = 4
            off, size, res, hashed = (int(v) for v in spec.levels[lvl])
            assert not hashed
            stride = res + 1
            cells = []
            for idx in range(8191, size - stride * stride, 8192):
                cx, cy, cz = idx % stride, (idx // stride) % stride, idx // (stride * stride)
                if cx < res and cy < res and cz < res:
                    cells.append((cx, cy, cz))
            assert len(cells) >= 8
            scale = float(spec.scales[lvl])
            c = torch.tensor(cells, dtype=torch.float32)[torch.randint(0, len(cells), (n // 2,), generator=g)]
            x[: n // 2] = (c + torch.rand(n // 2, 3, generator=g) * 0.98 + 0.01 - 0.5) / scale
        x = x.clamp(0.0, 1.0).to(device)
        dy = torch.randn(n, 2 * cfg["n_levels"], generator=g).to(device)
        grads = []
        for mode in (0, 7):
            _set_bwd_mode(enc, mode)
            enc.params.grad = None
            (enc(x).float() * dy).sum().backward()
            grads.append(enc.params.grad.clone())
        _assert_close(grads[1], grads[0], rtol=1e-3 if cfg is ngp else 4.2e-4, atol_scale=1e-5 if cfg is ngp else 4.2e-6,
                      what=f"packed pair records vs atomics ({what})")
        dy16 = (dy * 128).half().double() / 128
        assert abs(grads[1].double().sum().item() - dy16.sum().item()) <= 1e-2 * dy16.abs().sum().item() ** 0.5 + 1.0


def test_spherical_harmonics(device):
    import nerf_vo_amd.tinycudann as tcnn
    from oracle import sh as S

    for degree in (1, 2, 3, 4):
        enc = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": degree}).to(device)
        g = torch.Generator().manual_seed(degree)
        d = torch.nn.functional.normalize(torch.randn(1000, 3, generator=g), dim=-1)
        d01 = ((d + 1) / 2).to(device).requires_grad_(True)
        y = enc(d01)
        assert y.shape == (1000, degree * degree)
        dy = torch.randn(1000, degree * degree, generator=g).to(device)
        (y.float() * dy).sum().backward()
        dr = d01.detach().double().cpu().requires_grad_(True)
        yr = S.sh_encode(dr, degree)
        dy16 = (dy.cpu() * 128).half().double() / 128
        _assert_close(y, yr, rtol=4.7e-4, atol_scale=4.7e-4, what=f"SH degree {degree}")
        if degree == 1:  # constant encoding: the gradient is exactly zero
            assert (d01.grad == 0).all()
            continue
        (yr * dy16).sum().backward()
        _assert_close(d01.grad, dr.grad, rtol=4.4e-7, atol_scale=2.2e-7, what=f"SH degree {degree} dL/dd")


MLP_SHAPES = [
    # n_in, n_out, width, n_hidden, act, out_act
    (32, 16, 64, 1, "ReLU", "None"),      # nerfacto base MLP
    (63, 3, 64, 2, "ReLU", "Sigmoid"),    # colour MLP
    (27, 64, 64, 3, "ReLU", "None"),      # predicted normals MLP
    (10, 1, 16, 1, "ReLU", "None"),       # proposal density MLP (standalone)
    (16, 16, 16, 1, "ReLU", "None"),
    (27, 3, 32, 2, "ReLU", "Sigmoid"),    # tcnn n_neurons = 32 (not on the NeRF-VO path)
    (16, 16, 32, 1, "ReLU", "None"),
]


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("shape", MLP_SHAPES, ids=[f"{s[0]}-{s[2]}x{s[3]}-{s[1]}" for s in MLP_SHAPES])
def test_network_fwd_bwd(device, shape, dtype):
    """Fused MLP vs the float64 oracle that rounds at the kernel's 16-bit storage points.  f16: fp16 operands
    (11 significant bits) -> rtol 1e-2 / 2e-2.  bf16 (v_mfma_f32_16x16x16_bf16, BASELINE configs[4]): 8 significant
    bits, so a value that sits near a rounding boundary moves by 2^-8 relative when the fp32 accumulation order
    differs from the oracle's exact sum -> tolerances x8 (rtol 8e-2 / atol 4e-2 of the tensor's scale)."""
    import nerf_vo_amd.tinycudann as tcnn
    from oracle import mlp as M
    from oracle.quant import activation_format, q16

    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    k = 1.0 if dtype == "f16" else BF16_K
    n_in, n_out, width, n_hidden, act, out_act = shape
    cfg = {"otype": "FullyFusedMLP", "activation": act, "output_activation": out_act, "n_neurons": width,
           "n_hidden_layers": n_hidden}
    net = tcnn.Network(n_in, n_out, cfg, dtype=tdt).to(device)
    assert net.params.numel() == M.mlp_n_params(n_in, n_out, width, n_hidden)
    g = torch.Generator().manual_seed(11)
    p = torch.randn(net.params.numel(), generator=g) * (1.5 / np.sqrt(width))
    with torch.no_grad():
        net.params.copy_(p.to(device))
    n = 3000  # ragged: padded to 3072
    x = torch.randn(n, n_in, generator=g).to(device).requires_grad_(True)
    y = net(x)
    assert y.shape == (n, n_out) and y.dtype == tdt
    dy = torch.randn(n, n_out, generator=g).to(device)
    (y.float() * dy).sum().backward()
    torch.cuda.synchronize()

    with activation_format(dtype):
        pr = p.to(tdt).double().requires_grad_(True)
        ws = M.split_weights(pr, n_in, n_out, width, n_hidden)
        xr = x.detach().double().cpu().requires_grad_(True)
        yr = M.mlp_forward(xr, ws, act, out_act, pad_value=1.0)[:, :n_out]
        dy16 = q16(dy.cpu().double() * 128) / 128  # the kernel sees dy * loss_scale in its 16-bit format
        (yr * dy16).sum().backward()

    _assert_close(y, yr, rtol=1e-3 * k, atol_scale=5e-4 * k, what="MLP output")
    _assert_close(x.grad, xr.grad, rtol=1.9e-3 * k, atol_scale=9.3e-4 * k, what="MLP dL/dinput", max_outlier_frac=1e-3 if k > 1 else 0.0)
    _assert_close(net.params.grad, pr.grad, rtol=3.1e-3 * k, atol_scale=1.55e-3 * k, what="MLP dL/dparams")


@pytest.mark.parametrize("deterministic", [False, True], ids=["atomics", "deterministic"])
def test_network_backward_flags_an_overflow_inside_the_chain(device, deterministic):
    """Option "nonfinite_flag_ptr" on a fused MLP: the backward ORs the word with 1 when a weight-gradient total it
    flushes is not finite.  dL/doutput = 400 * 128 = 51200 is FINITE in fp16 (a clean root), but with weights of 0.5 the
    hidden dZ = sum over 16 outputs of 0.5 * 51200 = 409600 is not: the overflow exists only inside the 16-bit chain
    and must raise the flag through dW = dZ^T H (what replaced the optimiser's scan of the non-grid gradient ranges)."""
    import nerf_vo_amd.tinycudann as tcnn

    cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1}
    net = tcnn.Network(32, 16, cfg).to(device)
    flags = torch.zeros(4, dtype=torch.int32, device=device)
    net.native_tcnn_module.set_option("nonfinite_flag_ptr", flags.data_ptr() + 4)
    net.native_tcnn_module.set_option("deterministic", int(deterministic))
    with torch.no_grad():
        net.params.fill_(0.5)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2048, 32, generator=g).to(device).requires_grad_(True)
    for dy, expect in ((1e-3, 0), (400.0, 1), (1e-3, 1)):  # (the word is OR-ed, never cleared by the kernel)
        net.params.grad = None
        x.grad = None
        (net(x).float() * dy).sum().backward()
        torch.cuda.synchronize()
        assert flags.tolist() == [0, expect, 0, 0], (dy, flags.tolist())
        assert bool(torch.isfinite(net.params.grad).all()) == (dy < 1.0)
    flags.zero_()
    net.native_tcnn_module.set_option("nonfinite_flag_ptr", 0)  # detached again: nothing is written
    net.params.grad = None
    (net(x).float() * 400.0).sum().backward()
    torch.cuda.synchronize()
    assert flags.tolist() == [0, 0, 0, 0]


def test_network_identity_layout(device):
    """A = I style check with an ASYMMETRIC weight: catches row/col swaps in the MFMA fragment maps."""
    import nerf_vo_amd.tinycudann as tcnn
    from oracle import mlp as M

    cfg = {"otype": "FullyFusedMLP", "activation": "None", "output_activation": "None", "n_neurons": 64,
           "n_hidden_layers": 1}
    net = tcnn.Network(32, 16, cfg).to(device)
    w0 = torch.zeros(64, 32)
    w0[:32, :32] = torch.eye(32)
    w1 = (torch.arange(16 * 64).float().view(16, 64) % 61 - 30) / 32.0  # asymmetric, exactly fp16
    with torch.no_grad():
        net.params.copy_(torch.cat([w0.flatten(), w1.flatten()]).to(device))
    x = ((torch.arange(128 * 32).float().view(128, 32) % 17) - 8) / 8.0
    y = net(x.to(device))
    ref = (x @ w0.t()) @ w1.t()
    assert torch.equal(y.float().cpu(), ref.half().float()), (y.float().cpu() - ref).abs().max()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("cfg,width", [(MAIN, 64), (PROP0, 16)], ids=["main-64", "prop0-16"])
def test_network_with_input_encoding(device, cfg, width, dtype):
    """bf16: network weights / encoded features / gradients in bfloat16, hash table fp16 with fp32 interpolation and
    fp32 (fixed-point) gradient accumulation -- BASELINE configs[4]; tolerances x8 (8 vs 11 significant bits)."""
    import nerf_vo_amd.tinycudann as tcnn
    from oracle import grid as G
    from oracle import mlp as M
    from oracle.quant import activation_format, q16

    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    k = 1.0 if dtype == "f16" else BF16_K
    spec = _spec(cfg)
    n_out = 16 if width == 64 else 1
    net_cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": width,
               "n_hidden_layers": 1}
    model = tcnn.NetworkWithInputEncoding(3, n_out, _enc_cfg(cfg), net_cfg, dtype=tdt).to(device)
    n_net = M.mlp_n_params(spec.n_output_dims, n_out, width, 1)
    assert model.params.numel() == n_net + spec.n_params
    g = torch.Generator().manual_seed(3)
    p = torch.cat([torch.randn(n_net, generator=g) * (1.5 / np.sqrt(width)),
                   torch.rand(spec.n_params, generator=g) * 2 - 1])
    with torch.no_grad():
        model.params.copy_(p.to(device))
    n = 2000
    x = torch.from_numpy(_points(n, 9)).to(device).requires_grad_(True)
    y = model(x)
    assert y.dtype == tdt
    dy = torch.randn(n, n_out, generator=g).to(device)
    (y.float() * dy).sum().backward()
    torch.cuda.synchronize()

    with activation_format(dtype):
        # working copy: network weights in the network's format, hash table ALWAYS fp16
        pr = torch.cat([p[:n_net].to(tdt).double(), p[n_net:].to(torch.float16).double()]).requires_grad_(True)
        ws = M.split_weights(pr[:n_net], spec.n_output_dims, n_out, width, 1)
        table = pr[n_net:].view(-1, 2)
        xr = x.detach().double().cpu().requires_grad_(True)
        enc = G.grid_encode(spec, xr, table, quantize_output=True)
        yr = M.mlp_forward(enc, ws, "ReLU", "None", pad_value=0.0)[:, :n_out]
        dy16 = q16(dy.cpu().double() * 128) / 128
        (yr * dy16).sum().backward()

    _assert_close(y, yr, rtol=1.2e-3 * k, atol_scale=6e-4 * k, what="NWIE output")
    _assert_close(model.params.grad[:n_net], pr.grad[:n_net], rtol=4e-3 * k, atol_scale=2e-3 * k, what="NWIE dW")
    # the encoding gradient passes through a 16-bit d(encoded) buffer (_assert_close_chain: aggregate + cap)
    chain = {(64, "f16"): (7e-4, 9e-3), (64, "bf16"): (5.5e-3, 9e-2), (16, "f16"): (6e-4, 6.5e-4), (16, "bf16"): (5e-3, 4e-3)}
    _assert_close_chain(model.params.grad[n_net:], pr.grad[n_net:], "NWIE dgrid", *chain[(width, dtype)])
    _assert_close(x.grad, xr.grad, rtol=3e-2 * k, atol_scale=1e-2 * k, what="NWIE dL/dx")


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("cfg,width,compact", [(MAIN, 64, False), (PROP0, 16, True), (PROP1, 16, False)],
                         ids=["main-64", "prop0-16-compact", "prop1-16"])
def test_fused_encoding_forward_is_bit_identical(device, cfg, width, compact, dtype):
    """option fuse_encoding: the hash grid evaluated inside the MLP kernel's operand load (NVO_IO_GRID_FUSED) must give
    the SAME bits as the two-kernel form -- output, the encoded features it leaves in ctx for the backward, and hence
    every gradient."""
    import nerf_vo_amd.tinycudann as tcnn

    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    n_out = 16 if width == 64 else 1
    net_cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": width,
               "n_hidden_layers": 1}
    g = torch.Generator().manual_seed(21)
    n = 16 * 301  # several tiles per wave and a ragged last wave
    x = torch.from_numpy(_points(n, 4)).to(device)
    x[0] = 0.0
    x[1] = 1.0  # both domain faces (dense-index wrap)
    dy = torch.randn(n, n_out, generator=g).to(device)
    res = []
    params = None
    for fused in (0, 1):
        model = tcnn.NetworkWithInputEncoding(3, n_out, _enc_cfg(cfg), net_cfg, dtype=tdt).to(device)
        if params is None:
            params = torch.cat([torch.randn(model.params.numel(), generator=g) * 0.3]).to(device)
        with torch.no_grad():
            model.params.copy_(params)
        model.native_tcnn_module.set_option("fuse_encoding", fused)
        if compact:
            model.native_tcnn_module.set_option("recompute_hidden", 1)
        xx = x.clone().requires_grad_(True)
        y = model(xx)
        (y.float() * dy).sum().backward()
        torch.cuda.synchronize()
        res.append((y.detach().clone(), model.params.grad.clone(), xx.grad.clone()))
    assert torch.equal(res[0][0].view(torch.int16), res[1][0].view(torch.int16)), "fused forward output differs"
    assert torch.equal(res[0][2], res[1][2]), "dL/dx differs (the encoded features left in ctx differ)"
    # the weight gradient is flushed with float atomics (order-dependent in the last bits); the grid gradient of the
    # default slice-owner scatter is deterministic for single-chunk slices -- both must agree to rounding
    _assert_close(res[1][1], res[0][1], rtol=5e-5, atol_scale=5e-7, what="dL/dparams, fused vs two-kernel forward")


@pytest.mark.parametrize("kind", ["encoding", "network", "network_with_input_encoding"])
def test_empty_and_ragged_batches(device, kind):
    """Batches the 128-row granularity does not divide, and none at all: an EMPTY batch is a no-op with an empty output
    and zero gradients (torch hands NULL pointers for empty tensors -- nvo_fwd / nvo_bwd take them for batch 0), a
    ragged batch gives the rows of the padded one bit for bit (forward) and the same input gradient rows."""
    import nerf_vo_amd.tinycudann as tcnn

    net_cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
               "n_hidden_layers": 1}
    if kind == "encoding":
        m = tcnn.Encoding(3, _enc_cfg(PROP0)).to(device)
    elif kind == "network":
        m = tcnn.Network(32, 16, net_cfg).to(device)
    else:
        m = tcnn.NetworkWithInputEncoding(3, 16, _enc_cfg(PROP0), net_cfg).to(device)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        m.params.copy_((torch.randn(m.params.numel(), generator=g) * 0.3).to(device))
    x0 = torch.zeros(0, m.n_input_dims, device=device, requires_grad=True)
    y0 = m(x0)
    assert tuple(y0.shape) == (0, m.n_output_dims) and y0.dtype == m.dtype
    y0.float().sum().backward()
    torch.cuda.synchronize()
    assert tuple(x0.grad.shape) == (0, m.n_input_dims)
    assert m.params.grad is not None and float(m.params.grad.abs().max()) == 0.0
    m.params.grad = None
    full = torch.rand(256, m.n_input_dims, generator=g).to(device)
    dy = torch.randn(256, m.n_output_dims, generator=g).to(device)
    xf = full.clone().requires_grad_(True)
    yf = m(xf)
    (yf.float() * dy).sum().backward()
    m.params.grad = None
    for B in (1, 127, 129):
        xb = full[:B].clone().requires_grad_(True)
        yb = m(xb)
        assert tuple(yb.shape) == (B, m.n_output_dims)
        assert torch.equal(yb.view(torch.int16), yf[:B].view(torch.int16)), f"batch {B}: rows differ from the padded batch"
        (yb.float() * dy[:B]).sum().backward()
        torch.cuda.synchronize()
        assert torch.equal(xb.grad, xf.grad[:B]), f"batch {B}: input gradient rows differ"
        assert bool(torch.isfinite(m.params.grad).all())
        m.params.grad = None


@pytest.mark.parametrize("coherent", [True, False], ids=["ray-ordered", "random"])
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("cfg,width", [(MAIN, 64), (PROP0, 16), (PROP1, 16)], ids=["main", "prop0", "prop1"])
def test_forward_over_runs_is_bit_identical(device, coherent, dtype, cfg, width):
    """option grid_fwd_runs (what the inference path switches on): a thread walks four consecutive samples of a level and
    gathers only where the cell changes -- the SAME bits as the thread-per-sample forward, on ray-ordered samples (runs of
    equal cells on the coarse levels, none on the fine ones) and on random ones (no runs at all), both domain faces
    included.  The proposal grids take the form with the two coarsest levels in LDS (k_grid_fwd_small_runs)."""
    import nerf_vo_amd.tinycudann as tcnn

    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    net_cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": width,
               "n_hidden_layers": 1}
    g = torch.Generator().manual_seed(33)
    rays, per_ray = 96, 48
    n = rays * per_ray  # 4608 = 36 x 128
    if coherent:
        o = torch.rand(rays, 1, 3, generator=g) * 0.5 + 0.25
        d = torch.nn.functional.normalize(torch.randn(rays, 1, 3, generator=g), dim=-1)
        t = (torch.arange(per_ray).float()[None, :, None] * 2.0e-3) * (1.0 + torch.rand(rays, 1, 1, generator=g))
        x = (o + d * t).clamp(0.0, 1.0).reshape(n, 3)
    else:
        x = torch.from_numpy(_points(n, 4))
    x[0] = 0.0
    x[1] = 1.0  # both domain faces (dense-index wrap)
    x = x.contiguous().to(device)
    res = []
    params = None
    for runs in (0, 1):
        model = tcnn.NetworkWithInputEncoding(3, 16 if width == 64 else 1, _enc_cfg(cfg), net_cfg, dtype=tdt).to(device)
        if params is None:
            params = (torch.randn(model.params.numel(), generator=g) * 0.3).to(device)
        with torch.no_grad():
            model.params.copy_(params)
            model.native_tcnn_module.set_option("grid_fwd_runs", runs)
            res.append(model(x).clone())
    torch.cuda.synchronize()
    assert bool((res[0].float().abs() > 0).any())
    assert torch.equal(res[0].view(torch.int16), res[1].view(torch.int16)), "forward over runs differs from the per-sample forward"


def _raw_nwie(device, cfg, compact, acc_bits=32, compact_live=0, runs=0):
    """A proposal-shaped NetworkWithInputEncoding driven through the raw C-ABI the way the engine drives it."""
    import json

    from nerf_vo_amd.tinycudann.modules import _create

    net_cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 16,
               "n_hidden_layers": 1}
    m = _create("nvo_create_network_with_input_encoding", 3, 1, json.dumps(_enc_cfg(cfg)).encode(), json.dumps(net_cfg).encode())
    m.set_option("grid_bwd_mode", 1)
    m.set_option("grid_acc_bits", acc_bits)
    m.set_option("grid_compact_live", compact_live)
    m.set_option("grid_bwd_runs", runs)
    if compact:
        m.set_option("compact_output", 1)
        m.set_option("recompute_hidden", 1)
    return m


@pytest.mark.parametrize("zero_frac", [0.0, 0.9, 1.0], ids=["all-live", "90pct-zero", "all-zero"])
def test_zero_gradient_samples_are_skipped_exactly(device, zero_frac):
    """Proposal-network backward with most dL/dout EXACTLY zero, clustered in runs along the rays as in training: the
    compact-output kernel skips all-zero 16-sample tiles and the slice-owner scatter scans the list of live samples
    (option grid_compact_live).  Reference: the same network with full-width [B][16] output rows (no tile skipping
    in that instantiation) and the scan over all samples.  dL/dx must agree to the last bit, the gradients up to the
    order of the float atomics that flush multi-chunk slices and the MLP weight gradient."""
    from nerf_vo_amd.engine import _call, _ptr, _stream

    g = torch.Generator().manual_seed(17)
    N = 16 * 1024
    x = torch.from_numpy(_points(N, 6)).to(device)
    st = _stream(device)
    dout = torch.randn(N, generator=g)
    keep = (torch.rand(N // 64, generator=g) >= zero_frac).repeat_interleave(64)  # runs of 64 samples live or dead
    dout = (dout * keep).to(device)
    res = {}
    for tag, compact, live in (("ref", False, 0), ("skip", True, 1)):
        m = _raw_nwie(device, PROP0, compact, compact_live=live)
        if "params" not in res:
            res["params"] = (torch.randn(m.n_params, generator=g) * 0.3).to(device)
        ph = res["params"].half()
        ctx = torch.empty(m.ctx_bytes(N), dtype=torch.uint8, device=device)
        out = torch.empty(N if compact else (N, 16), dtype=torch.float16, device=device)
        _call("nvo_fwd", m.handle, st, N, _ptr(x), _ptr(ph), _ptr(out), _ptr(ctx))
        if compact:
            dy = (dout * 128).half()
        else:
            dy = torch.zeros(N, 16, dtype=torch.float16, device=device)
            dy[:, 0] = (dout * 128).half()
        dx = torch.full((N, 3), 7.0, device=device)
        dp = torch.full((m.n_params,), 7.0, device=device)
        _call("nvo_bwd", m.handle, st, N, _ptr(x), _ptr(ph), _ptr(out), _ptr(dy), _ptr(ctx), _ptr(dx), _ptr(dp))
        torch.cuda.synchronize()
        res[tag] = (out.float().view(N, -1)[:, 0].clone(), dx.clone(), dp.clone())
    n_net = 16 * 16 + 16 * 16
    assert torch.equal(res["ref"][0], res["skip"][0])
    assert torch.equal(res["ref"][1], res["skip"][1]), "dL/dx differs"
    # (multi-chunk slices and the MLP weight gradient are flushed with float atomics: equal up to summation order.  The
    # grid gradient accumulates in int32 with the scale 2^29 / L1(dy): with the live list the L1 norms are summed inside
    # k_live_samples, without it by k_dy_l1 -- another summation order, a scale that differs in its last bits, hence
    # every addend rounded to the integer grid independently in the two runs: quantum L1 / 2^29 per addend.)
    _assert_close(res["skip"][2][n_net:], res["ref"][2][n_net:], rtol=1e-5, atol_scale=2e-7, what="grid gradient, live list")
    _assert_close(res["skip"][2][:n_net], res["ref"][2][:n_net], rtol=1e-4, atol_scale=1e-6, what="dW with skipped tiles")  # (float-atomic sums in two orders: 0.04-0.4 of this bound from run to run)
    if zero_frac == 1.0:
        assert float(res["skip"][2].abs().max()) == 0.0 and float(res["skip"][1].abs().max()) == 0.0


def _ray_points(n_rays, n_samples, seed):
    """Samples along rays through the unit cube, ray-major (the order the samplers emit)."""
    rng = np.random.default_rng(seed)
    o = rng.random((n_rays, 1, 3), dtype=np.float32) * 0.2 + 0.4
    d = rng.standard_normal((n_rays, 1, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = np.sort(rng.random((n_rays, n_samples, 1), dtype=np.float32), axis=1) * 0.55
    return np.clip(o + t * d, 0.0, 1.0).reshape(-1, 3).astype(np.float32)


@pytest.mark.parametrize("live", [0, 1], ids=["all-samples", "live-list"])
@pytest.mark.parametrize("acc_bits", [64, 32])
@pytest.mark.parametrize("cfg", [MAIN, PROP0], ids=["main", "prop0"])
def test_run_merged_dense_levels_match_slice_owner(device, cfg, acc_bits, live):
    """Option grid_bwd_runs on ray-ordered samples (the case it is built for: a coarse cell holds a run of consecutive
    samples), level-major dL/dy as the fused networks hand it over (16-byte vector loads), some samples with an exactly
    zero gradient.  Reference: the same slice-owner items scanning sample by sample.  Both sum fp32 products; the
    run-merged form adds the products of a run in fp32 registers before the one conversion to fixed point, so the
    agreement is to fp32 rounding of the run sums (64-bit accumulators) or to the int32 quantum L1 / 2^29 per
    addend (32-bit accumulators)."""
    from nerf_vo_amd.engine import _call, _ptr, _stream

    g = torch.Generator().manual_seed(23)
    n_rays, S = 256, 48
    N = n_rays * S
    x = torch.from_numpy(_ray_points(n_rays, S, 11)).to(device)
    st = _stream(device)
    dout = torch.randn(N, generator=g) * (torch.rand(N, generator=g) > 0.25)
    dout = dout.to(device)
    res = {}
    params = None
    for tag, runs in (("owner", 0), ("runs", 1)):
        m = _raw_nwie(device, cfg, True, acc_bits=acc_bits, runs=runs, compact_live=live)
        if runs:  # a batch hint that does NOT match the launch (the chunking is a tuning input, never a contract)
            m.set_option("grid_bwd_batch", 3 * N + 16)
        if params is None:
            params = (torch.randn(m.n_params, generator=g) * 0.3).to(device)
        ph = params.half()
        ctx = torch.empty(m.ctx_bytes(N), dtype=torch.uint8, device=device)
        out = torch.empty(N, dtype=torch.float16, device=device)
        _call("nvo_fwd", m.handle, st, N, _ptr(x), _ptr(ph), _ptr(out), _ptr(ctx))
        dy = (dout * 128).half()
        dx = torch.zeros((N, 3), device=device)
        dp = torch.full((m.n_params,), 7.0, device=device)
        _call("nvo_bwd", m.handle, st, N, _ptr(x), _ptr(ph), _ptr(out), _ptr(dy), _ptr(ctx), _ptr(dx), _ptr(dp))
        torch.cuda.synchronize()
        res[tag] = dp.clone()
    n_net = 16 * 16 + 16 * 16
    assert float(res["owner"][n_net:].abs().max()) > 0
    tol = dict(rtol=5e-6, atol_scale=5e-7) if acc_bits == 64 else dict(rtol=5e-4, atol_scale=5e-5)
    _assert_close(res["runs"][n_net:], res["owner"][n_net:], what="grid gradient, run-merged", **tol)


def test_double_backward_raises_like_upstream(device):
    """``bwd_bwd_input`` exists and RAISES (tcnn's FullyFusedMLP has no second-order backward
    either).  A first-order torch.autograd.grad -- what nerfacto's analytic normals use -- works; differentiating through
    it (create_graph=True, then backward) must fail loudly instead of silently returning a constant's gradient."""
    import nerf_vo_amd.tinycudann as tcnn

    net = tcnn.NetworkWithInputEncoding(
        3, 16, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 12, "base_resolution": 4,
                "per_level_scale": 1.5},
        {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1}).to(device)
    x = torch.rand(256, 3, device=device, requires_grad=True)
    y = net(x).float()
    (g,) = torch.autograd.grad(y[:, 0].sum(), x, retain_graph=True)  # first order: fine
    assert torch.isfinite(g).all() and float(g.abs().sum()) > 0
    (g2,) = torch.autograd.grad(y[:, 0].sum(), x, create_graph=True)
    with pytest.raises(NotImplementedError, match="bwd_bwd_input"):
        g2.sum().backward()
    with pytest.raises(NotImplementedError, match="bwd_bwd_input"):
        net.native_tcnn_module.bwd_bwd_input()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("cfg", [PROP0, PROP1, MAIN], ids=["prop0", "prop1", "main"])
def test_grid_forward_forms_are_bit_identical(device, cfg, dtype):
    """The level-major forward has several forms of ONE arithmetic (module option grid_fwd_small_form).  Proposal grids:
    0 the first thread-per-(sample, level) kernel, 1 coarse levels from LDS, 3 software-pipelined, 4 instruction-lean (the
    default).  Main grid: 0 the first kernel, anything else the instruction-lean one (the default).  Same fp32
    interpolation in the same order, one rounding: the outputs must agree bit for bit, on ray-coherent samples, on cells
    at the domain faces (the dense levels' wrap takes the lean forms' generic branch), on positions outside [0, 1]
    (memory-safe garbage in every form, the SAME garbage), and for batches that end inside a pass / a tile."""
    import nerf_vo_amd.tinycudann as tcnn

    wide = cfg["n_levels"] == 16
    net = tcnn.NetworkWithInputEncoding(3, 16 if wide else 1, _enc_cfg(cfg), {
        "otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64 if wide else 16,
        "n_hidden_layers": 1}).to(device)
    m = net.native_tcnn_module
    m.set_option("bf16", int(dtype == "bf16"))
    with torch.no_grad():
        net.params.uniform_(-1, 1)
    g = torch.Generator(device="cpu").manual_seed(5)
    for n, S in ((4096 * 96, 96), (1024 * 256 + 640, 256), (131 * 128, 96)):
        R = (n + S - 1) // S
        o = (torch.rand(R, 1, 3, generator=g) - 0.5) * 1.2
        d = torch.nn.functional.normalize(torch.randn(R, 1, 3, generator=g), dim=-1)
        t = 1.0 / torch.linspace(1.0 / 0.05, 1.0 / 30.0, S).view(1, S, 1)
        p = o + d * t
        mag = p.abs().amax(dim=-1, keepdim=True).clamp_min(1e-9)
        p = torch.where(mag > 1, (2 - 1 / mag) * (p / mag), p)
        x = ((p + 2) / 4).reshape(-1, 3)[:n].contiguous()
        x[:7] = torch.from_numpy(_points(16, 3)[:7])          # faces, corners, one ulp inside them
        x[7:71] = torch.rand(64, 3, generator=g).round()       # every lane of a wave on a face / edge / corner
        x[71] = torch.tensor([1.5, -0.25, 0.5])                # outside the unit cube
        x[72] = torch.tensor([-3.0, 7.0, 2.0])
        x = x.to(device)
        outs = {}
        for form in (0, 1, 3, 4):
            m.set_option("grid_fwd_small_form", form)
            with torch.no_grad():
                y = net(x)
            torch.cuda.synchronize()
            outs[form] = y.view(torch.int16).cpu()
        # the run-walking inference forms (option grid_fwd_runs: four consecutive samples per thread, gathers only where
        # the cell changes): the first kernels (form 1) and the instruction-lean one of the small grids (form 4)
        m.set_option("grid_fwd_runs", 1)
        for form in (1, 4):
            m.set_option("grid_fwd_small_form", form)
            with torch.no_grad():
                y = net(x)
            torch.cuda.synchronize()
            outs[("runs", form)] = y.view(torch.int16).cpu()
        m.set_option("grid_fwd_runs", 0)
        m.set_option("grid_fwd_small_form", -1)
        for form in (0, 3, 4, ("runs", 1), ("runs", 4)):
            diff = int((outs[form] != outs[1]).sum())
            assert diff == 0, f"form {form} differs from form 1 in {diff} of {outs[1].numel()} outputs (n={n})"
