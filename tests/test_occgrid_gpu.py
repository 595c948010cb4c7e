"""GPU: cascaded occupancy grid (Morton bitfield, DDA marcher, EMA/bitfield/max-pool update) vs the C
oracle -- BIT-EXACT: occupancy hits, per-ray sample counts, the t/dt of every sample, bitfield bytes."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _scene_grid(n_levels, seed):
    """A blobby density field so that rays cross empty and occupied regions in every cascade."""
    from oracle import occgrid as O

    rng = np.random.default_rng(seed)
    grid = (rng.random((n_levels, O.CELLS), dtype=np.float32) ** 6) * 0.08
    grid[:, ::7] = 0.0
    grid[0, :1000] = -1.0  # never-visible cells stay negative through the EMA
    return grid


@pytest.mark.parametrize("n_levels,cone", [(3, 1.0 / 256.0), (1, 0.0), (5, 0.004)])
def test_march_bit_exact(device, n_levels, cone):
    from nerf_vo_amd import _lib
    from oracle import occgrid as O

    lib = _lib.lib()
    grid = _scene_grid(n_levels, 1)
    bf = O.grid_to_bitfield(grid, n_levels)
    rng = np.random.default_rng(2)
    R = 512
    o = (rng.random((R, 3), dtype=np.float32) - 0.5) * 0.9 + 0.5
    o[:8] = rng.random((8, 3), dtype=np.float32) * 6 - 2.5  # some origins outside every cascade
    d = rng.normal(size=(R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[8] = (1.0, 0.0, 0.0)  # axis-aligned ray: infinite inverse direction components
    jit = rng.random(R).astype(np.float32)
    cap = R * 1024
    od, dd, jd = (torch.from_numpy(a).to(device) for a in (o, d, jit))
    bfd = torch.from_numpy(bf).to(device)
    counts = torch.zeros(R, dtype=torch.int32, device=device)
    offsets = torch.zeros(R + 1, dtype=torch.int32, device=device)
    ridx = torch.full((cap,), -1, dtype=torch.int32, device=device)
    t = torch.zeros(cap, device=device)
    dt = torch.zeros(cap, device=device)
    scratch = torch.empty(int(lib.nvo_occ_march_scratch_bytes(R)), dtype=torch.uint8, device=device)
    _lib.check(lib.nvo_occ_march(_stream(), R, _p(od), _p(dd), _p(bfd), n_levels, cone, 0.0, _p(jd), cap, _p(counts),
                                 _p(offsets), _p(ridx), _p(t), _p(dt), _p(scratch), scratch.numel()), "occ_march")
    torch.cuda.synchronize()
    rc, rt, rdt = O.march_rays(o, d, bf, n_levels, cone, 0.0, jit)
    got_c = counts.cpu().numpy().astype(np.uint32)
    assert (got_c == rc).all(), f"{int((got_c != rc).sum())} rays with different sample counts"
    assert rc.sum() > 1000 and ((rc == 0).any() or n_levels > 3)
    off = offsets.cpu().numpy().astype(np.int64)
    assert (off[:-1] == np.concatenate([[0], np.cumsum(rc)[:-1]])).all() and off[-1] == rc.sum()
    tt, dtt, rr = t.cpu().numpy(), dt.cpu().numpy(), ridx.cpu().numpy()
    for r in range(R):
        n = int(rc[r])
        sl = slice(off[r], off[r] + n)
        assert (rr[sl] == r).all()
        assert (tt[sl].view(np.uint32) == rt[r, :n].view(np.uint32)).all(), f"ray {r}: t differs"
        assert (dtt[sl].view(np.uint32) == rdt[r, :n].view(np.uint32)).all(), f"ray {r}: dt differs"


@pytest.mark.parametrize("n_levels,cone", [(3, 1.0 / 256.0), (1, 0.0)])
def test_ray_per_lane_march_matches_wave_per_ray(device, n_levels, cone):
    """A launch of >= 49 152 rays (an inference bundle) takes the ray-per-lane march, a training batch the wave-per-ray one
    (pinned to the C oracle by test_march_bit_exact): the same 65 536 rays through both -- one launch against four of
    16 384 -- must give the same counts and the same (t, dt) of every sample bit for bit, in one pass and in rounds (48
    samples, then the rest from where the first round stopped, appended to the ray's run: t_resume / t_next / run_offset),
    rays outside every cascade, an axis-aligned ray and rays that sit a round out included."""
    from nerf_vo_amd import _lib
    from oracle import occgrid as O

    lib = _lib.lib()
    bf = O.grid_to_bitfield(_scene_grid(n_levels, 1), n_levels)
    rng = np.random.default_rng(5)
    R, chunk = 65536, 16384
    o = (rng.random((R, 3), dtype=np.float32) - 0.5) * 0.9 + 0.5
    o[:64] = rng.random((64, 3), dtype=np.float32) * 6 - 2.5
    d = rng.normal(size=(R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[64] = (0.0, 1.0, 0.0)
    jit = rng.random(R).astype(np.float32)
    od, dd, jd = (torch.from_numpy(a).to(device) for a in (o, d, jit))
    bfd = torch.from_numpy(bf).to(device)
    nscr = int(lib.nvo_occ_march_scratch_bytes(R))

    def march(rays_per_launch, rounds):
        scratch = torch.zeros(nscr, dtype=torch.uint8, device=device)
        counts = [torch.zeros(R, dtype=torch.int32, device=device) for _ in rounds]
        t_next = torch.full((R,), -2.0, device=device)
        t_resume = None
        base = 0
        for k, budget in enumerate(rounds):
            last = k + 1 == len(rounds)
            for lo in range(0, R, rays_per_launch):
                n = min(rays_per_launch, R - lo)
                at = lambda t, w: None if t is None else C.c_void_p(t.data_ptr() + w * lo)  # noqa: E731
                _lib.check(lib.nvo_occ_march_runs(
                    _stream(), n, at(od, 12), at(dd, 12), _p(bfd), n_levels, cone, 0.0, at(jd, 4), at(counts[k], 4),
                    C.c_void_p(scratch.data_ptr() + 8 * 1024 * lo), nscr - 8 * 1024 * lo, at(t_resume, 4), budget,
                    None if last else at(t_next, 4), None, base), "occ_march_runs")
            if not last:
                t_resume = t_next.clone()
                t_resume[::5] = -1.0  # (every fifth ray sits the next round out)
            base += budget
        torch.cuda.synchronize()
        return [c.cpu().numpy() for c in counts], scratch.cpu().numpy().view(np.uint32).reshape(R, 1024, 2), t_next.cpu().numpy()

    for rounds in ([1024], [48, 976]):
        c_lane, runs_lane, next_lane = march(R, rounds)
        c_wave, runs_wave, next_wave = march(chunk, rounds)
        total = np.zeros(R, np.int64)
        for k in range(len(rounds)):
            assert (c_lane[k] == c_wave[k]).all(), f"rounds {rounds}, round {k}: {int((c_lane[k] != c_wave[k]).sum())} counts differ"
            total += c_lane[k]
        assert total.sum() > 100 * R / 4 and (total == 0).any()
        if len(rounds) > 1:
            assert (next_lane.view(np.uint32) == next_wave.view(np.uint32)).all(), "t_next differs"
            assert (c_lane[1][::5] == 0).all() and (c_lane[1] > 0).any()
            # (a later round appends at run_offset: the run is contiguous only where the first round filled its budget)
            used = (np.arange(1024)[None, :] < c_lane[0][:, None]) | \
                   ((np.arange(1024)[None, :] >= rounds[0]) & (np.arange(1024)[None, :] < rounds[0] + c_lane[1][:, None]))
        else:
            used = np.arange(1024)[None, :] < total[:, None]
        assert (runs_lane[used] == runs_wave[used]).all(), f"rounds {rounds}: (t, dt) differ"


def test_capacity_drops_whole_rays(device):
    from nerf_vo_amd import _lib
    from oracle import occgrid as O

    lib = _lib.lib()
    bf = np.full((1, O.CELLS // 8), 0xFF, np.uint8)  # everything occupied
    R = 64
    o = np.full((R, 3), 0.5, np.float32)
    d = np.tile(np.array([[0.6, 0.64, 0.48]], np.float32), (R, 1))
    cap = 1000
    tens = [torch.from_numpy(a).to(device) for a in (o, d, bf)]
    counts = torch.zeros(R, dtype=torch.int32, device=device)
    offsets = torch.zeros(R + 1, dtype=torch.int32, device=device)
    ridx = torch.full((cap,), -1, dtype=torch.int32, device=device)
    t = torch.zeros(cap, device=device)
    dt = torch.zeros(cap, device=device)
    scratch = torch.empty(int(lib.nvo_occ_march_scratch_bytes(R)), dtype=torch.uint8, device=device)
    _lib.check(lib.nvo_occ_march(_stream(), R, _p(tens[0]), _p(tens[1]), _p(tens[2]), 1, 0.0, 0.0, None, cap,
                                 _p(counts), _p(offsets), _p(ridx), _p(t), _p(dt), _p(scratch), scratch.numel()), "occ_march")
    # an undersized staging area is refused, not overrun
    assert lib.nvo_occ_march(_stream(), R, _p(tens[0]), _p(tens[1]), _p(tens[2]), 1, 0.0, 0.0, None, cap, _p(counts),
                             _p(offsets), _p(ridx), _p(t), _p(dt), _p(scratch), scratch.numel() - 8) != 0
    torch.cuda.synchronize()
    c = counts.cpu().numpy()
    per_ray = c[0]
    assert per_ray > 100 and (c[: cap // per_ray] == per_ray).all() and (c[cap // per_ray:] == 0).all()
    assert (ridx.cpu().numpy()[: (cap // per_ray) * per_ray] >= 0).all()


def test_update_bitfield_and_maxpool_bit_exact(device):
    from nerf_vo_amd import _lib
    from oracle import occgrid as O

    lib = _lib.lib()
    for seed, thr in ((3, 0.01), (4, 0.5)):  # second case: mean < threshold decides
        n_levels = 3
        grid = _scene_grid(n_levels, seed)
        fresh = (np.random.default_rng(seed + 10).random((n_levels, O.CELLS), dtype=np.float32) ** 5) * 0.1
        ref_grid = O.ema_update(grid, fresh, 0.95)
        ref_bf = O.grid_to_bitfield(ref_grid, n_levels, thr)
        gd = torch.from_numpy(grid.copy()).to(device)
        fd = torch.from_numpy(fresh).to(device)
        bfd = torch.zeros(n_levels, O.CELLS // 8, dtype=torch.uint8, device=device)
        scratch = torch.zeros(8, dtype=torch.uint8, device=device)
        _lib.check(lib.nvo_occ_update(_stream(), n_levels, _p(gd), _p(fd), 0.95, thr, _p(bfd), _p(scratch)), "occ_update")
        torch.cuda.synchronize()
        assert (gd.cpu().numpy().view(np.uint32) == ref_grid.view(np.uint32)).all()
        got = bfd.cpu().numpy()
        assert (got == ref_bf).all(), f"{int((got != ref_bf).sum())} bitfield bytes differ"
        assert np.unpackbits(got[1]).mean() > np.unpackbits(O.grid_to_bitfield(ref_grid[:1], 1, thr)).mean() * 0.1


def test_cell_positions_invert_the_index(device):
    from nerf_vo_amd import _lib
    from oracle import occgrid as O

    lib = _lib.lib()
    for level in (0, 2):
        pos = torch.zeros(O.CELLS, 3, device=device)
        _lib.check(lib.nvo_occ_cell_positions(_stream(), level, None, _p(pos)), "cell_positions")
        torch.cuda.synchronize()
        p = pos.cpu().numpy()
        sel = np.random.default_rng(0).integers(0, O.CELLS, 2000)
        u = (p[sel] - 0.5) / (2.0 ** level) + 0.5
        ijk = np.floor(u * 128).astype(np.uint32)
        assert (O.morton3d_numpy(ijk[:, 0], ijk[:, 1], ijk[:, 2]) == sel).all()


@pytest.mark.parametrize("n_levels,thresh", [(3, -0.01), (3, 0.01), (1, 0.01), (5, 0.01)])
def test_refresh_samples_bit_exact(device, n_levels, thresh):
    """nvo_occ_sample_cells (scattered density-grid refresh past the warm-up) == the numpy restatement:
    cascade, candidate-cell sequence with the rejection against the grid, point inside the cell -- uint32 / float32 bit
    for bit; the uniform pass (thresh < 0) skips only untrained (negative) cells, the occupied pass lands on cells above
    the threshold whenever one of its ten candidates is."""
    from nerf_vo_amd import _lib
    from oracle import occgrid as O

    lib = _lib.lib()
    grid = _scene_grid(n_levels, 5)
    h = 0.5 * (1 << (n_levels - 1))
    lo, hi = 0.5 - h, 0.5 + h
    g_dev = torch.from_numpy(grid.reshape(-1)).to(device)
    n_total, first, n, step, seed, sid = 3 * (1 << 17), 70001, 50000, 4321, 1337, 1
    x01 = torch.full((n, 3), -7.0, device=device)
    cells = torch.zeros(n, dtype=torch.int32, device=device)
    rc = lib.nvo_occ_sample_cells(_stream(), n, first, n_total, step, seed, sid, n_levels, _p(g_dev), thresh, lo, hi,
                                  _p(x01), _p(cells))
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    want_cells, want_x = O.refresh_samples(n, first, n_total, step, seed, sid, n_levels, grid, thresh, lo, hi)
    got_cells = cells.cpu().numpy().view(np.uint32)
    assert (got_cells == want_cells).all()
    assert (x01.cpu().numpy().view(np.uint32) == want_x.view(np.uint32)).all()
    vals = grid.reshape(-1)[got_cells]
    if thresh < 0:
        assert (vals >= 0).mean() > 0.999  # (1000 untrained cells of 2 M: ten candidates in a row is out of reach)
    else:
        assert (vals > thresh).mean() > 0.9 and (grid > thresh).mean() < 0.5
    # every cascade is drawn, points stay inside the scene box and inside their cell
    lev = got_cells >> 21
    assert set(np.unique(lev)) == set(range(n_levels))
    xs = x01.cpu().numpy()
    assert xs.min() >= 0.0 and xs.max() <= 1.0
    cx, cy, cz = O.morton3d_invert_numpy(got_cells & (O.CELLS - 1))
    p = xs.astype(np.float64) * (hi - lo) + lo
    for a, c in enumerate((cx, cy, cz)):
        u = ((p[:, a] - 0.5) / np.ldexp(1.0, lev.astype(np.int32)) + 0.5) * O.GRID
        assert (np.floor(u + 1e-4) >= c).all() and (u <= c + 1 + 1e-4).all()
    # bad arguments are refused before any launch
    assert lib.nvo_occ_sample_cells(_stream(), n, n_total - 10, n_total, step, seed, sid, n_levels, _p(g_dev), thresh, lo, hi,
                                    _p(x01), _p(cells)) != 0


def test_thickness_splat_keeps_the_cell_maximum(device):
    """nvo_ngp_thickness_splat: fresh[cell] = max over the cell's samples of the thickness nvo_ngp_thickness computes for
    that sample in the cell's cascade -- bit for bit (same expression), cells without a sample stay 0, NaN is dropped."""
    from nerf_vo_amd import _lib
    from oracle import occgrid as O

    lib = _lib.lib()
    n, n_levels = 40000, 3
    g = torch.Generator().manual_seed(3)
    out = (torch.randn(n, 16, generator=g) * 3).half()
    out[17, 0] = float("nan")
    cells_np = np.random.default_rng(8).integers(0, 5000, n).astype(np.uint32) * 1237 % (n_levels * O.CELLS)
    cells_np[0:13000] = cells_np[13000:26000]  # duplicates
    cells = torch.from_numpy(cells_np.view(np.int32)).to(device)
    out_d = out.to(device)
    fresh = torch.zeros(n_levels * O.CELLS, device=device)
    assert lib.nvo_ngp_thickness_splat(_stream(), n, _p(out_d), 16, _p(cells), _p(fresh)) == 0, _lib.last_error()
    per_level = []
    for level in range(n_levels):
        tmp = torch.empty(n, device=device)
        assert lib.nvo_ngp_thickness(_stream(), n, _p(out_d), 16, level, _p(tmp)) == 0
        per_level.append(tmp.cpu().numpy())
    torch.cuda.synchronize()
    per = np.stack(per_level)[cells_np >> 21, np.arange(n)]
    want = np.zeros(n_levels * O.CELLS, np.float32)
    ok = ~np.isnan(per)
    np.maximum.at(want, cells_np[ok], per[ok])
    got = fresh.cpu().numpy()
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    assert (got > 0).sum() == len(np.unique(cells_np[ok]))


@pytest.mark.parametrize("margin", [0.0, 1.0])
def test_mark_untrained_cells_matches_the_oracle(device, margin):
    """nvo_occ_mark_untrained (SURVEY.md 2.4 K16 mark_untrained_density_grid): the seen / unseen verdict of every sampled
    cell equals the numpy restatement's; unseen cells become -1, seen cells that were marked come back as 0, every other
    value is kept; with more cameras no cell loses its view."""
    from nerf_vo_amd import _lib
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.synthetic import make_sequence
    from oracle import occgrid as O

    lib = _lib.lib()
    n_levels, F, H, W = 3, 6, 60, 80
    seq = make_sequence(F, H, W, device=device, scene_scale=0.2)
    c2w = opencv_to_opengl(seq["camera_extrinsics"])
    c2w[:, :3, 3] += 0.5
    c2w = c2w[:, :3, :4].contiguous()
    K = seq["camera_intrinsics"].contiguous()
    rng = np.random.default_rng(2)
    grid = rng.random(n_levels * O.CELLS, dtype=np.float32) * 0.05
    grid[::5] = -1.0
    g = torch.from_numpy(grid).to(device)
    assert lib.nvo_occ_mark_untrained(_stream(), n_levels, _p(g), 3, _p(K), _p(c2w), H, W, margin) == 0, _lib.last_error()
    torch.cuda.synchronize()
    got = g.cpu().numpy()
    cells = rng.integers(0, n_levels * O.CELLS, 150000).astype(np.uint32)
    seen = O.cells_seen(cells, n_levels, K.cpu().numpy()[:3], c2w.cpu().numpy()[:3], H, W, margin)
    if margin > 0:  # a superset of upstream's trainable cells
        strict = O.cells_seen(cells, n_levels, K.cpu().numpy()[:3], c2w.cpu().numpy()[:3], H, W, 0.0)
        assert (seen | ~strict).all() and seen.sum() > strict.sum()
    assert 0.02 < seen.mean() < 0.98
    was_marked = grid[cells] < 0
    want = np.where(seen, np.where(was_marked, np.float32(0.0), grid[cells]), np.float32(-1.0))
    assert (got[cells].view(np.uint32) == want.view(np.uint32)).all(), int((got[cells] != want).sum())
    # the coarser the cascade, the larger the share of cells outside every frustum; more cameras only add views
    g6 = torch.from_numpy(grid).to(device)
    assert lib.nvo_occ_mark_untrained(_stream(), n_levels, _p(g6), F, _p(K), _p(c2w), H, W, margin) == 0
    got6 = g6.cpu().numpy()
    assert ((got6 >= 0) | (got < 0)).all() and (got6 >= 0).sum() > (got >= 0).sum()
    # marking again with the same cameras changes nothing
    g6b = g6.clone()
    assert lib.nvo_occ_mark_untrained(_stream(), n_levels, _p(g6b), F, _p(K), _p(c2w), H, W, margin) == 0
    assert torch.equal(g6b, g6)


@pytest.mark.parametrize("first", [1, 37, 64, 200])
def test_march_in_rounds_concatenates_to_the_single_march(device, first):
    """nvo_occ_march_resume: a ray marched in rounds (at most `first` samples, then the rest from the candidate t_next handed
    out) yields exactly the (t, dt) sequence of one uninterrupted march -- bit for bit; rays that left the box or whose
    resume value is negative produce nothing."""
    from nerf_vo_amd import _lib
    from oracle import occgrid as O

    lib = _lib.lib()
    n_levels, cone, R, cap = 3, 1.0 / 256.0, 384, 1 << 19
    grid = _scene_grid(n_levels, 1)
    bf = torch.from_numpy(O.grid_to_bitfield(grid, n_levels)).to(device)
    rng = np.random.default_rng(21)
    o = torch.from_numpy(((rng.random((R, 3), dtype=np.float32) - 0.5) * 0.8 + 0.5)).to(device)
    d = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((R, 3)).astype(np.float32)), dim=-1).to(device)
    jit = torch.from_numpy(rng.random(R, dtype=np.float32)).to(device)
    scratch = torch.empty(int(lib.nvo_occ_march_scratch_bytes(R)), dtype=torch.uint8, device=device)

    def march(t_resume, max_new, want_next):
        counts = torch.zeros(R, dtype=torch.int32, device=device)
        offsets = torch.zeros(R + 1, dtype=torch.int32, device=device)
        ray_idx = torch.full((cap,), -1, dtype=torch.int32, device=device)
        t = torch.zeros(cap, device=device)
        dt = torch.zeros(cap, device=device)
        nxt = torch.full((R,), 7.0, device=device) if want_next else None
        rc = lib.nvo_occ_march_resume(_stream(), R, _p(o), _p(d), _p(bf), n_levels, cone, 0.1, _p(jit), cap, _p(counts),
                                      _p(offsets), _p(ray_idx), _p(t), _p(dt), _p(scratch), scratch.numel(), _p(t_resume),
                                      max_new, _p(nxt))
        assert rc == 0, _lib.last_error()
        torch.cuda.synchronize()
        c, off = counts.cpu().numpy(), offsets.cpu().numpy()
        tt, dd = t.cpu().numpy(), dt.cpu().numpy()
        return c, [np.stack([tt[off[r]:off[r] + c[r]], dd[off[r]:off[r] + c[r]]]) for r in range(R)], \
            (nxt.cpu().numpy() if want_next else None)

    c_all, runs_all, _ = march(None, 1024, False)
    assert c_all.sum() > 20000 and c_all.max() > first
    c1, runs1, nxt = march(None, first, True)
    assert (c1 == np.minimum(c_all, first)).all()
    assert ((nxt >= 0) | (c_all <= first)).all()          # a ray with samples left hands out where it goes on
    resume = torch.from_numpy(nxt).to(device)
    c2, runs2, nxt2 = march(resume, 1024, True)
    assert (c1 + c2 == c_all).all() and (nxt2 < 0).all()
    for r in range(R):
        both = np.concatenate([runs1[r], runs2[r]], axis=1)
        assert (both.view(np.uint32) == runs_all[r].view(np.uint32)).all(), r
    assert (c2[nxt < 0] == 0).all()


@pytest.mark.parametrize("R,R_used,cap", [(65536, 65536, 1 << 22), (65536, 13312, 1 << 18), (16384, 12345, 1 << 18), (4096, 100, 1 << 15),
                                          (64, 64, 1 << 10), (1, 1, 64), (256, 0, 1 << 10)])
def test_pack_in_one_launch_matches_the_three_launch_pack(device, R, R_used, cap):
    """nvo_occ_pack_fused (scan by decoupled look-back between 16-ray workgroups + copy + network input of every copied
    sample) against nvo_occ_pack + nvo_ngp_positions: counts, offsets, totals, ray_idx, t, dt and x01 bit for bit, with
    the ray count on the device, rays dropped at the capacity, a run offset, and 40 launches in a row on one state block
    (every launch runs in a new epoch of it) -- up to 4096 workgroups waiting on each other's totals."""
    from nerf_vo_amd import _lib
    from oracle import ngp as ON

    lib = _lib.lib()
    g = torch.Generator().manual_seed(R + R_used)
    n_scr = int(lib.nvo_occ_march_scratch_bytes(R))
    scratch = torch.rand(n_scr // 4, generator=g).to(device)  # (t, dt) pairs of every run slot
    origins = (torch.rand(R, 3, generator=g) - 0.5).to(device)
    directions = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(device)
    state = torch.zeros(int(lib.nvo_occ_pack_state_bytes()), dtype=torch.uint8, device=device)
    r_dev = torch.tensor([R_used], dtype=torch.int32, device=device)
    i32 = dict(dtype=torch.int32, device=device)
    for it in range(40):
        run_offset = 0 if it % 2 == 0 else 17
        hi = max(2, min(1024 - run_offset, (3 * cap) // max(R_used, 1)))  # about a third of the launches overflow the capacity
        counts = torch.randint(0, hi, (R,), generator=g, dtype=torch.int32)
        counts[torch.rand(R, generator=g) < 0.2] = 0
        counts = counts.to(device)
        out = {}
        for fused in (False, True):
            c_out, offs, tot = torch.full((R,), -7, **i32), torch.full((R + 1,), -7, **i32), torch.full((2,), -7, **i32)
            ridx = torch.full((cap,), -1, **i32)
            t, dt = torch.zeros(cap, device=device), torch.zeros(cap, device=device)
            x01 = torch.zeros(cap, 3, device=device)
            if fused:
                _lib.check(lib.nvo_occ_pack_fused(_stream(), R, _p(counts), cap, _p(c_out), _p(offs), _p(tot), _p(scratch), n_scr,
                                                  _p(ridx), _p(t), _p(dt), _p(r_dev), run_offset, _p(state), _p(origins),
                                                  _p(directions), -1.5, 2.5, _p(x01), 16 if it % 4 < 2 else 64), "nvo_occ_pack_fused")
            else:
                _lib.check(lib.nvo_occ_pack(_stream(), R, _p(counts), cap, _p(c_out), _p(offs), _p(tot), _p(scratch), n_scr,
                                            _p(ridx), _p(t), _p(dt), _p(r_dev), run_offset), "nvo_occ_pack")
                _lib.check(lib.nvo_ngp_positions(_stream(), cap, _p(ridx), _p(t), _p(origins), _p(directions), -1.5, 2.5, _p(x01)),
                           "nvo_ngp_positions")
            torch.cuda.synchronize()
            out[fused] = (c_out[:R_used], offs[:R_used + 1], tot, ridx, t, dt, x01)
        for a, b, name in zip(out[False], out[True], ("counts", "offsets", "totals", "ray_idx", "t", "dt", "x01")):
            assert torch.equal(a, b), (name, it)
        kept, offsets, total = ON.compact_offsets(counts[:R_used].cpu().numpy(), cap)
        assert (out[True][0].cpu().numpy() == kept).all() and (out[True][1].cpu().numpy() == offsets).all()
        assert out[True][2].cpu().tolist() == [total, min(total, cap)]
