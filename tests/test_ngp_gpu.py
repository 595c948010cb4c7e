"""GPU parity of the occupancy-grid ("instant-ngp") back-end vs the CPU oracle on identical rays:
bit-exact packed samples, rendered colour/depth, losses, all gradients; plus the density-grid update
and a short training run."""
import numpy as np
import pytest
import torch

from test_tcnn_gpu import _assert_close, _assert_close_chain

pytestmark = pytest.mark.gpu


def _engine(device, **kw):
    from nerf_vo_amd.ngp_engine import NgpConfig, NgpEngine

    kw.setdefault("march_capacity", 1 << 17)
    cfg = NgpConfig(num_images=4, capacity=1 << 15, **kw)
    eng = NgpEngine(cfg, device)
    g = torch.Generator().manual_seed(9)
    flat = torch.zeros(eng.n_params)
    nd = eng.n_density_mlp
    flat[:nd] = (torch.rand(nd, generator=g) * 2 - 1) * 0.25
    n_grid = eng.density_net.n_params - nd
    flat[nd:nd + n_grid] = (torch.rand(n_grid, generator=g) * 2 - 1) * 0.8
    flat[eng.density_net.n_params:] = (torch.rand(eng.n_rgb, generator=g) * 2 - 1) * 0.3
    eng.set_params(flat)
    return eng


def _oracle(eng):
    from oracle.ngp import NgpOracle

    orc = NgpOracle(aabb_scale=eng.cfg.aabb_scale, cone_angle=eng.cfg.cone_angle, near=eng.cfg.near_distance,
                    depth_mult=eng.cfg.depth_loss_mult)
    ph = eng.params_half.double().cpu()
    nd = eng.n_density_mlp
    nden = eng.density_net.n_params
    orc.params = {"density_mlp": ph[:nd].clone().requires_grad_(True),
                  "grid": ph[nd:nden].clone().view(-1, 2).requires_grad_(True),
                  "rgb_mlp": ph[nden:].clone().requires_grad_(True)}
    return orc


@pytest.mark.parametrize("cov,compact,rounds", [("none", True, 0), ("ones", True, 0), ("varying", True, 0), ("none", False, 0),
                                                ("none", True, 128)])
def test_ngp_step_matches_oracle(device, cov, compact, rounds):
    """``cov``: the per-ray variance of the depth target (nvo_ngp_loss_args::gt_depth_cov) -- absent, all ones (must be
    the absent case bit for bit) or spread over three decades with zeros, negatives and inf mixed in (rays whose depth
    term is dropped).  ``compact``: the batch that is trained on holds the samples in front of T < 1e-4 only
    (NgpConfig.compact_training, upstream's training-batch compaction): marched samples bit for bit, kept counts against the
    numpy restatement on the kernel's own density outputs, packing rule, then the step against the oracle composited over
    the kept samples.  ``rounds`` > 0: the pass that finds the kept counts runs in three rounds (NgpConfig.train_rounds),
    its counts must be the single pass's."""
    from oracle import occgrid as O
    from oracle import ngp as ON

    eng = _engine(device, compact_training=compact, train_rounds=())
    if compact:  # denser medium: a good share of the rays must end in front of their last sample
        nd = eng.n_density_mlp
        boosted = eng.params.clone()
        boosted[:nd] *= 3.0
        eng.set_params(boosted)
    # a structured occupancy grid (independent of the network) so that rays see gaps and hits
    rng = np.random.default_rng(3)
    grid = (rng.random((eng.cfg.n_levels, O.CELLS), dtype=np.float32) ** 8) * 0.05
    eng.density_grid.copy_(torch.from_numpy(grid.reshape(-1)).to(device))
    from nerf_vo_amd.engine import _call
    from nerf_vo_amd.tinycudann.modules import _ptr, _stream
    _call("nvo_occ_update", _stream(device), eng.cfg.n_levels, _ptr(eng.density_grid), None, 0.95, 0.01,
          _ptr(eng.bitfield), _ptr(eng._scratch8))
    bf = eng.bitfield.cpu().numpy().reshape(eng.cfg.n_levels, -1)
    assert (bf == O.grid_to_bitfield(grid, eng.cfg.n_levels)).all()

    R = 96
    g = torch.Generator().manual_seed(4)
    origins = (torch.rand(R, 3, generator=g) - 0.5) * 0.6 + 0.5
    directions = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    dnorm = 1.0 + 0.2 * torch.rand(R, generator=g)
    jitter = torch.rand(R, generator=g)
    gt_rgb, gt_depth = torch.rand(R, 3, generator=g), torch.rand(R, generator=g) * 0.8
    gt_depth[::5] = 0.0
    bg = torch.rand(R, 3, generator=g)

    ws = eng._workspace(R, True)
    ws["origins"].copy_(origins)
    ws["directions"].copy_(directions)
    ws["directions_norm"].copy_(dnorm)
    ws["gt_rgb"].copy_(gt_rgb)
    ws["gt_depth"].copy_(gt_depth)
    gt_cov = None
    if cov != "none":
        gt_cov = torch.ones(R) if cov == "ones" else 10.0 ** (torch.rand(R, generator=g) * 3 - 1.5)
        if cov == "varying":
            gt_cov[3::11] = 0.0
            gt_cov[4::13] = -1.0
            gt_cov[6::17] = float("inf")
        ws["gt_depth_cov"].copy_(gt_cov)
        ws["has_depth_cov"] = True
    eng.forward_backward(ws, jitter.to(device), has_depth=True, background=bg.to(device))
    torch.cuda.synchronize()
    if compact:
        # the threshold of THIS test: the median of the transmittance the rays end with, so that about half of them are cut
        # (with random weights next to no ray reaches the production 1e-4)
        om, cm = ws["offsets_m"].cpu().numpy(), ws["counts_m"].cpu().numpy()
        dm, dtm = ws["density_m"].float().cpu().numpy().astype(np.float64), ws["dt_m"].cpu().numpy().astype(np.float64)
        t_end = np.array([np.exp(-np.sum(np.minimum(np.exp(dm[om[r]:om[r] + cm[r]]) * dtm[om[r]:om[r] + cm[r]], 128.0))) for r in range(R)])
        eng.cfg.train_min_transmittance = float(np.median(t_end[cm > 0]))
        assert 1e-6 < eng.cfg.train_min_transmittance < 0.999
        eng.forward_backward(ws, jitter.to(device), has_depth=True, background=bg.to(device))
        torch.cuda.synchronize()
    if cov == "ones":
        # the plain L2 term, bit for bit: the per-sample gradients the loss kernel writes (the parameter gradients behind
        # them and the loss shards are sums of float atomics, whose order varies from launch to launch)
        got_d, got_c, got_l = ws["d_density_pre"].clone(), ws["d_rgb_out"].clone(), eng.loss_dict()
        ws["has_depth_cov"] = False
        eng.forward_backward(ws, jitter.to(device), has_depth=True, background=bg.to(device))
        torch.cuda.synchronize()
        assert torch.equal(got_d, ws["d_density_pre"]) and torch.equal(got_c, ws["d_rgb_out"])
        assert got_d.abs().max() > 0
        for k, v in eng.loss_dict().items():
            assert abs(v - got_l[k]) <= 1e-6 * abs(v)

    orc = _oracle(eng)
    counts, t, dt = orc.march(origins, directions, bf, jitter)
    sfx = "_m" if compact else ""
    got_counts = ws["counts" + sfx].cpu().numpy().astype(np.uint32)
    assert int(counts.sum()) <= eng.cfg.capacity and counts.sum() > 500
    assert (got_counts == counts).all()
    off = ws["offsets" + sfx].cpu().numpy()
    tt = ws["t" + sfx].cpu().numpy()
    for r in range(R):
        n = int(counts[r])
        assert (tt[off[r]:off[r] + n].view(np.uint32) == t[r, :n].view(np.uint32)).all()
    kept = None
    if compact:
        dens = ws["density_m"].float().cpu().numpy()
        thr = eng.cfg.train_min_transmittance
        kept_ref, margin = ON.alive_counts(counts, [dt[r] for r in range(R)], [dens[off[r]:off[r] + int(counts[r])] for r in range(R)], thr)
        kept = ws["kept"].cpu().numpy().astype(np.int64)
        tie = margin < 1e-5  # (a transmittance within the kernel's fp32 rounding of the threshold may fall either way)
        assert (kept == kept_ref)[~tie].all() and (np.abs(kept - kept_ref) <= 1).all() and tie.sum() <= R // 4
        cut = kept < counts
        assert cut.sum() >= 10 and (~cut & (counts > 0)).sum() >= 10, (cut.sum(), (~cut).sum())
        assert (ws["ray_state"].cpu().numpy() == cut.astype(np.int32)).all()
        c2, o2, total = ON.compact_offsets(kept, eng.cfg.capacity)
        assert (ws["counts"].cpu().numpy() == c2).all() and (ws["offsets"].cpu().numpy() == o2).all()
        assert ws["totals"].cpu().tolist() == [total, min(total, eng.cfg.capacity)] and total == int(kept.sum())
        if rounds:
            # the same step with the counts found in two rounds: same counts (ties aside), same packed batch
            single = {k: ws[k].clone() for k in ("kept", "ray_state", "counts", "offsets", "t", "dt", "ray_idx")}
            eng.cfg.train_rounds = (rounds, rounds // 2)
            assert (kept < rounds).sum() >= 5 and (kept > rounds).sum() >= 5 and (kept > rounds + rounds // 2).sum() >= 2
            eng.forward_backward(ws, jitter.to(device), has_depth=True, background=bg.to(device))
            torch.cuda.synchronize()
            k2 = ws["kept"].cpu().numpy().astype(np.int64)
            assert (k2 == kept)[~tie].all() and (np.abs(k2 - kept) <= 1).all()
            if (k2 == kept).all():
                for k, v in single.items():
                    assert torch.equal(v, ws[k]), k
            kept = k2
            c2, o2, total = ON.compact_offsets(kept, eng.cfg.capacity)
        tc, rc = ws["t"].cpu().numpy(), ws["ray_idx"].cpu().numpy()
        for r in range(R):
            n = int(kept[r])
            assert (tc[o2[r]:o2[r] + n].view(np.uint32) == t[r, :n].view(np.uint32)).all() and (rc[o2[r]:o2[r] + n] == r).all()
        assert (rc[total:] == -1).all()

    rgb, depth, acc = orc.forward(origins.double(), directions.double(), counts, t, dt, background=bg.double(), kept=kept)
    ld = orc.loss_dict(rgb, depth, gt_rgb.double(), gt_depth.double(), dnorm.double(),
                       None if gt_cov is None else gt_cov.double())
    sum(ld.values()).backward()
    _assert_close(ws["out_rgb"], rgb.detach(), rtol=4.2e-5, atol_scale=2.1e-5, what="ngp rgb")
    _assert_close(ws["out_depth"], depth.detach(), rtol=3.7e-4, atol_scale=1.85e-4, what="ngp depth")
    _assert_close(ws["out_accumulation"], acc.detach(), rtol=2.6e-4, atol_scale=1.3e-4, what="ngp accumulation")
    got = eng.loss_dict()
    for k in ("rgb_loss", "depth_loss"):
        assert abs(got[k] - float(ld[k].detach())) <= 2e-2 * abs(float(ld[k].detach())) + 1e-7, (k, got[k], float(ld[k].detach()))
    ls = eng.cfg.loss_scale
    nd, nden = eng.n_density_mlp, eng.density_net.n_params
    gr = (eng.grads / ls).double().cpu()
    _assert_close(gr[nden:], orc.params["rgb_mlp"].grad, what="d rgb MLP", rtol=4.8e-3, atol_scale=2.4e-3, max_outlier_frac=1e-4)
    _assert_close(gr[:nd], orc.params["density_mlp"].grad, what="d density MLP", rtol=8.7e-3, atol_scale=4.35e-3, max_outlier_frac=1e-4)
    # through the 16-bit chain: relative L1 error + largest error (_assert_close_chain)
    _assert_close_chain(gr[nd:nden], orc.params["grid"].grad, "d hash grid", 3.6e-3, 0.05)


def test_ngp_extrinsics_gradient_matches_oracle(device):
    """optimize_extrinsics (reference: nerf_vo/mapping/instant_ngp.py:47): dL/d(camera offset) through the
    packed samples' positions vs autograd in the oracle (positions only: the SH directions are detached, as
    the kernel path does).  Tolerance as the nerfacto pose-gradient test: the chain passes through fp16
    d(encoded) buffers: rtol 5e-2, atol 3e-2 * max|ref|."""
    from oracle import occgrid as O
    from oracle import rays as Rr
    from nerf_vo_amd.engine import _call
    from nerf_vo_amd.tinycudann.modules import _ptr, _stream

    eng = _engine(device, extrinsic_l2_reg=0.0)
    rng = np.random.default_rng(3)
    grid = (rng.random((eng.cfg.n_levels, O.CELLS), dtype=np.float32) ** 8) * 0.05
    eng.density_grid.copy_(torch.from_numpy(grid.reshape(-1)).to(device))
    _call("nvo_occ_update", _stream(device), eng.cfg.n_levels, _ptr(eng.density_grid), None, 0.95, 0.01,
          _ptr(eng.bitfield), _ptr(eng._scratch8))
    bf = eng.bitfield.cpu().numpy().reshape(eng.cfg.n_levels, -1)

    F, H, W, R = 4, 24, 32, 96
    g = torch.Generator().manual_seed(6)
    pose = torch.randn(F, 6, generator=g) * 0.02
    pose[2] = 0.0
    eng.pose_adjustment.copy_(pose.reshape(-1).to(device))
    intr = torch.tensor([[30.0, 28.0, 15.7, 11.6]]).repeat(F, 1)
    rot = torch.linalg.qr(torch.randn(F, 3, 3, generator=g))[0]
    c2w = torch.cat([rot, (torch.rand(F, 3, 1, generator=g) - 0.5) * 0.4 + 0.5], dim=2)
    idx = torch.stack([torch.randint(0, F, (R,), generator=g), torch.randint(0, H, (R,), generator=g),
                       torch.randint(0, W, (R,), generator=g)], dim=1)
    images = torch.rand(F, H, W, 3, generator=g)
    depths = torch.rand(F, H, W, 1, generator=g) * 0.8
    jitter = torch.rand(R, generator=g)

    ws = eng._workspace(R, True)
    eng.load_rays(ws, idx.to(device), intr.to(device), c2w.to(device).contiguous(), images.to(device), depths.to(device))
    eng.d_corrections.zero_()  # (a fresh window of the camera optimiser: the per-camera gradient accumulates across steps)
    eng.forward_backward(ws, jitter.to(device), has_depth=True, background=None)
    torch.cuda.synchronize()
    got = (eng.camera_gradient() / eng.cfg.loss_scale).view(F, 6).double().cpu()

    orc = _oracle(eng)
    pose_r = pose.double().requires_grad_(True)
    ro, rd, rn, _ = Rr.generate_rays(idx, intr.double(), c2w.double())
    corr = Rr.exp_map_so3xr3(pose_r)[idx[:, 0]]
    ro2, rd2 = Rr.apply_pose_correction(ro, rd, corr)
    _assert_close(ws["origins"], ro2.detach(), rtol=8.7e-8, atol_scale=8.7e-9, what="corrected origins")
    _assert_close(ws["directions"], rd2.detach(), rtol=4.6e-7, atol_scale=4.6e-8, what="corrected directions")
    # the marcher runs on the kernel's fp32 rays (bit-exact packed samples are covered by the test above)
    o32, d32 = ws["origins"].cpu(), ws["directions"].cpu()
    counts, t, dt = orc.march(o32, d32, bf, jitter)
    assert (ws["counts"].cpu().numpy().astype(np.uint32) == counts).all() and counts.sum() > 300
    rgb, depth, acc = orc.forward(ro2, rd2, counts, t, dt, background=None, sh_directions=rd2.detach())
    gt_rgb = images[idx[:, 0], idx[:, 1], idx[:, 2]].double()
    gt_depth = depths[idx[:, 0], idx[:, 1], idx[:, 2], 0].double()
    ld = orc.loss_dict(rgb, depth, gt_rgb, gt_depth, rn.reshape(-1).detach())
    sum(ld.values()).backward()
    ref = pose_r.grad
    assert ref.abs().max() > 0
    _assert_close(got, ref, rtol=2.3e-2, atol_scale=1.35e-2, what="dL/d(camera offset), occupancy-grid back-end")
    before = eng.pose_adjustment.clone()
    eng.optimizer_step()
    torch.cuda.synchronize()
    assert not torch.equal(before, eng.pose_adjustment)
    assert eng.camera_corrections().shape == (F, 3, 4)


def test_density_grid_update_and_training(device):
    """update_density_grid() must mark cells from the network's own density; 60 steps on a synthetic
    sequence must reduce the loss and keep the packed batch inside the capacity."""
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.ngp_engine import NgpConfig, NgpEngine
    from nerf_vo_amd.synthetic import make_sequence

    n, H, W = 8, 60, 80
    seq = make_sequence(n, H, W, device=device, scene_scale=0.2)
    c2w = opencv_to_opengl(seq["camera_extrinsics"])
    c2w[:, :3, 3] += 0.5  # centre the room on the unit cube of cascade 0
    c2w = c2w[:, :3, :4].contiguous()
    images = seq["frames_color"].permute(0, 2, 3, 1).contiguous()
    depths = seq["frames_depth"].permute(0, 2, 3, 1).contiguous()  # already in scaled scene units
    eng = NgpEngine(NgpConfig(num_images=n, num_rays=512, capacity=1 << 18), device)
    eng.update_density_grid()
    torch.cuda.synchronize()
    occupied = np.unpackbits(eng.bitfield.cpu().numpy()).mean()
    assert 0.01 < occupied <= 1.0
    losses = []
    scale = torch.tensor([n, H, W], device=device)
    for it in range(150):
        idx = torch.floor(torch.rand(512, 3, device=device) * scale).long()
        eng.train_step(idx, seq["camera_intrinsics"], c2w, images, depths)
        losses.append(eng.loss_dict()["rgb_loss"])
        assert eng.samples_last_step() <= eng.cfg.capacity
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)) and np.mean(losses[-10:]) < 0.6 * np.mean(losses[:5]), (losses[:5], losses[-10:])
    assert int(eng.skip_flag.item()) == 0
    # the first step marked the cells no camera sees (NgpConfig.mark_untrained): they stay negative and unoccupied
    unseen = (eng.density_grid < 0).cpu().numpy()
    assert eng._marked_images == n and 0.05 < unseen.mean() < 0.99
    bits = np.unpackbits(eng.bitfield.cpu().numpy(), bitorder="little").astype(bool)
    assert not bits[: unseen.size // eng.cfg.n_levels][unseen[: unseen.size // eng.cfg.n_levels]].any()


def test_weight_ema_matches_tcnn_formula(device):
    """nvo_ema_update vs the debiased moving average of tcnn's EmaOptimizer (restated): ema_t = (ema_{t-1} * d *
    (1 - d^(t-1)) + w_t * (1 - d)) / (1 - d^t); a raised skip flag leaves the average untouched."""
    from nerf_vo_amd.engine import _call
    from nerf_vo_amd.tinycudann.modules import _ptr, _stream

    n, d = 10007, 0.95
    g = torch.Generator().manual_seed(1)
    ema = torch.zeros(n, device=device)
    ema_half = torch.zeros(n, dtype=torch.float16, device=device)
    flag = torch.zeros(1, dtype=torch.int32, device=device)
    ref = torch.zeros(n, dtype=torch.float64)
    for t in range(1, 41):
        w = torch.randn(n, generator=g)
        _call("nvo_ema_update", _stream(device), n, _ptr(w.to(device)), _ptr(ema), _ptr(ema_half), d, t, _ptr(flag))
        ref = (ref * d * (1 - d ** (t - 1)) + w.double() * (1 - d)) / (1 - d ** t)
        if t == 1:
            assert torch.allclose(ema.cpu(), w, rtol=1e-6, atol=1e-7)  # the first average IS the weights
    torch.cuda.synchronize()
    assert torch.allclose(ema.cpu().double(), ref, rtol=2e-5, atol=2e-6)
    assert torch.equal(ema_half.cpu(), ema.cpu().half())
    before = ema.clone()
    flag.fill_(1)
    _call("nvo_ema_update", _stream(device), n, _ptr(torch.randn(n, generator=g).to(device)), _ptr(ema), _ptr(ema_half), d, 41,
          _ptr(flag))
    torch.cuda.synchronize()
    assert torch.equal(before, ema)


def test_adam_tail_in_one_launch_is_bit_identical(device):
    """nvo_adam_step_groups_tail (Adam + weight average + both commits in ONE launch, the commit by the last workgroup to
    finish) against the launches it replaces -- nvo_adam_step_groups, nvo_ema_update_dev_part, nvo_ema_update_dev (with
    its k_ema_commit) and nvo_opt_commit -- from identical states over several steps, one of them skipped: master weights,
    moments, 16-bit copy, average and its 16-bit copy, both counters, the bias corrections and the check-in counter (back
    at zero after every launch).  Three groups at offsets that make one of them take the scalar form, own weight decay."""
    import ctypes as C

    from nerf_vo_amd import _lib
    from nerf_vo_amd.engine import _call
    from nerf_vo_amd.tinycudann.modules import _ptr, _stream

    n = 300_007
    groups = ((0, 4096, 1e-6), (4096, 250_001, 0.0), (4096 + 250_001, n - 4096 - 250_001, 1e-6))
    g = torch.Generator().manual_seed(21)
    b1, b2, eps, decay, lr = 0.9, 0.99, 1e-15, 0.95, 1e-2

    def state():
        return {"p": torch.randn(n, generator=torch.Generator().manual_seed(5)).to(device),
                "p16": torch.zeros(n, dtype=torch.float16, device=device), "m": torch.zeros(n, device=device),
                "v": torch.zeros(n, device=device), "ema": torch.zeros(n, device=device),
                "ema16": torch.zeros(n, dtype=torch.float16, device=device),
                "ema_step": torch.zeros(1, dtype=torch.int32, device=device),
                "applied": torch.zeros(1, dtype=torch.int32, device=device),
                "bias": torch.tensor([1.0 - b1, (1.0 - b2) ** 0.5], device=device),
                "done": torch.zeros(1, dtype=torch.int32, device=device)}

    a, b = state(), state()
    flag = torch.zeros(1, dtype=torch.int32, device=device)
    stream = _stream(device)

    def batch(st):
        arr = [_lib.AdamGroup(offset=o, n=m, lr=lr, step=0, hyper_dev=None, bias_dev=st["bias"].data_ptr(), flag_slot=0,
                              flag_slot_set=1, weight_decay=wd, weight_decay_set=1) for o, m, wd in groups]
        return (_lib.AdamGroup * len(arr))(*arr)

    for it in range(5):
        grads = (torch.randn(n, generator=g) * 128.0).to(device)
        flag.fill_(1 if it == 3 else 0)
        # --- separate launches
        _call("nvo_adam_step_groups", stream, 3, batch(a), _ptr(a["p"]), _ptr(a["p16"]), _ptr(grads), 0, _ptr(a["m"]), _ptr(a["v"]),
              b1, b2, eps, 1.0 / 128.0, 0.0, _ptr(flag))
        head = groups[1][0] + groups[1][1]
        _call("nvo_ema_update_dev_part", stream, head, _ptr(a["p"]), _ptr(a["ema"]), _ptr(a["ema16"]), decay, _ptr(a["ema_step"]),
              _ptr(flag))
        _call("nvo_ema_update_dev", stream, n - head, C.c_void_p(a["p"].data_ptr() + 4 * head),
              C.c_void_p(a["ema"].data_ptr() + 4 * head), C.c_void_p(a["ema16"].data_ptr() + 2 * head), decay,
              _ptr(a["ema_step"]), _ptr(flag))
        _call("nvo_opt_commit", stream, 1, 1, 0, _ptr(a["applied"]), _ptr(flag), None, None, 2.0, 0.5, 2000, 0.0, 0.0,
              _ptr(a["bias"]), b1, b2)
        # --- one launch
        tail = _lib.AdamTail(ema=b["ema"].data_ptr(), ema_half=b["ema16"].data_ptr(), ema_decay=decay,
                             ema_step_dev=b["ema_step"].data_ptr(), ema_flag_slot=0, ema_commit=1,
                             done_counter=b["done"].data_ptr(), n_commit_groups=1, active_mask=1, scale_mask=0,
                             applied=b["applied"].data_ptr(), scale=None, growth_tracker=None, growth_factor=2.0,
                             backoff_factor=0.5, growth_interval=2000, min_scale=0.0, max_scale=0.0,
                             bias=b["bias"].data_ptr())
        _call("nvo_adam_step_groups_tail", stream, 3, batch(b), _ptr(b["p"]), _ptr(b["p16"]), _ptr(grads), 0, _ptr(b["m"]),
              _ptr(b["v"]), b1, b2, eps, 1.0 / 128.0, 0.0, _ptr(flag), 0, None, None, None, C.byref(tail))
        torch.cuda.synchronize()
        for name in a:
            x, y = a[name], b[name]
            x = x.view(torch.int16) if x.dtype == torch.float16 else x.view(torch.int32)
            y = y.view(torch.int16) if y.dtype == torch.float16 else y.view(torch.int32)
            assert torch.equal(x, y), f"step {it}: {name} differs ({int((x != y).sum())} words)"
        expect = it + 1 - (1 if it >= 3 else 0)
        assert int(b["applied"].item()) == expect and int(b["ema_step"].item()) == expect and int(b["done"].item()) == 0
    assert bool((b["ema"] != 0).all()) and bool((b["p16"] != 0).any())


def test_adam_tail_refuses_to_commit_on_a_dirty_counter(device):
    """The fused optimiser tail commits when its check-in counter arrives at EXACTLY the grid size.  A counter somebody
    left dirty never does: the launch must then neither commit (step counter, bias corrections, average counter stay as
    they were) nor reset the counter, and raise the sticky error bit (bit 31) that the host checks -- instead of
    committing before every workgroup has read the scalars the commit rewrites."""
    import ctypes as C

    from nerf_vo_amd import _lib
    from nerf_vo_amd.engine import _call
    from nerf_vo_amd.tinycudann.modules import _ptr, _stream

    n = 200_000
    b1, b2 = 0.9, 0.99
    p = torch.randn(n, device=device)
    p16 = torch.zeros(n, dtype=torch.float16, device=device)
    m, v = torch.zeros(n, device=device), torch.zeros(n, device=device)
    grads = torch.randn(n, device=device)
    applied = torch.zeros(1, dtype=torch.int32, device=device)
    bias = torch.tensor([1.0 - b1, (1.0 - b2) ** 0.5], device=device)
    flag = torch.zeros(1, dtype=torch.int32, device=device)
    done = torch.full((1,), 3, dtype=torch.int32, device=device)  # dirty
    group = (_lib.AdamGroup * 1)(_lib.AdamGroup(offset=0, n=n, lr=1e-2, step=0, hyper_dev=None, bias_dev=bias.data_ptr(),
                                                flag_slot=0, flag_slot_set=1, weight_decay=0.0, weight_decay_set=1))
    tail = _lib.AdamTail(ema=None, ema_half=None, ema_decay=0.0, ema_step_dev=None, ema_flag_slot=0, ema_commit=0,
                         done_counter=done.data_ptr(), n_commit_groups=1, active_mask=1, scale_mask=0,
                         applied=applied.data_ptr(), scale=None, growth_tracker=None, growth_factor=2.0, backoff_factor=0.5,
                         growth_interval=2000, min_scale=0.0, max_scale=0.0, bias=bias.data_ptr())
    bias0 = bias.clone()
    for _ in range(2):  # (the second launch meets the sticky bit and must stay failed)
        _call("nvo_adam_step_groups_tail", _stream(device), 1, group, _ptr(p), _ptr(p16), _ptr(grads), 0, _ptr(m), _ptr(v), b1, b2,
              1e-15, 1.0, 0.0, _ptr(flag), 0, None, None, None, C.byref(tail))
        torch.cuda.synchronize()
        word = int(done.item()) & 0xFFFFFFFF
        assert word & 0x80000000, f"no error bit on a dirty counter (word {word:#x})"
        assert int(applied.item()) == 0 and torch.equal(bias, bias0), "the launch committed on a dirty counter"
    # a clean counter commits and is left at zero
    done.zero_()
    _call("nvo_adam_step_groups_tail", _stream(device), 1, group, _ptr(p), _ptr(p16), _ptr(grads), 0, _ptr(m), _ptr(v), b1, b2,
          1e-15, 1.0, 0.0, _ptr(flag), 0, None, None, None, C.byref(tail))
    torch.cuda.synchronize()
    assert int(done.item()) == 0 and int(applied.item()) == 1 and not torch.equal(bias, bias0)


def test_adaptive_ray_batch_and_ema_inference(device):
    """The ray batch adapts toward the packed-sample target (NerfCounters::update_after_training [UPSTREAM]): after a
    few adaptations the marched samples per step sit within 25 % of the capacity, whatever batch the run started
    with; inference reads the moving average of the weights, training the raw ones."""
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.ngp_engine import NgpConfig, NgpEngine
    from nerf_vo_amd.synthetic import make_sequence

    n, H, W = 8, 60, 80
    seq = make_sequence(n, H, W, device=device, scene_scale=0.2)
    c2w = opencv_to_opengl(seq["camera_extrinsics"])
    c2w[:, :3, 3] += 0.5
    c2w = c2w[:, :3, :4].contiguous()
    images = seq["frames_color"].permute(0, 2, 3, 1).contiguous()
    depths = seq["frames_depth"].permute(0, 2, 3, 1).contiguous()
    scale = torch.tensor([n, H, W], device=device)
    for start in (256, 8192):
        eng = NgpEngine(NgpConfig(num_images=n, num_rays=start, capacity=1 << 16), device)
        seen = []
        for it in range(96):
            R = eng.rays_per_batch
            assert R % 128 == 0 and eng.cfg.min_rays <= R <= eng.cfg.max_rays
            idx = torch.floor(torch.rand(R, 3, device=device) * scale).long()
            eng.train_step(idx, seq["camera_intrinsics"], c2w, images, depths)
            seen.append((R, int(eng._ws["counts"].sum().item())))
        # (in the first hundred steps a ray of this scene still finds 250-600 samples: 2^16 slots take 128-256 rays)
        assert eng.rays_per_batch <= 512, "the batch never adapted"
        tail = np.mean([m for _, m in seen[-16:]])
        assert 0.75 * eng.cfg.capacity <= tail <= 1.25 * eng.cfg.capacity, (start, seen[-16:])
        assert int(eng.skip_flag.item()) == 0
    # EMA: exists, differs from the raw weights, and is what render_rays evaluates
    assert eng.ema_step == 96 and eng.inference_params_half() is eng.params_ema_half
    assert not torch.equal(eng.params_ema_half, eng.params_half)
    o = torch.tensor([[0.5, 0.5, 0.5]], device=device).repeat(64, 1)
    dirs = torch.nn.functional.normalize(torch.randn(64, 3, device=device), dim=-1)
    out_ema = eng.render_rays(o, dirs, torch.ones(64, device=device))["rgb"].clone()
    eng.cfg.ema_decay, saved = 0.0, (eng.params_ema, eng.params_ema_half)
    eng.params_ema = eng.params_ema_half = None
    out_raw = eng.render_rays(o, dirs, torch.ones(64, device=device))["rgb"].clone()
    eng.params_ema, eng.params_ema_half = saved
    assert torch.isfinite(out_ema).all() and not torch.equal(out_ema, out_raw)


def test_adam_and_ema_inside_the_grid_backward_are_bit_identical(device):
    """NgpConfig.fuse_grid_adam: the grid backward of the density network applies Adam AND the weight average to the
    entries of its streamed hashed levels while it still holds their finished gradient in LDS, and leaves that gradient
    unwritten.  Several steps from identical states (the parameters of the unfused engine are copied into the fused one
    before every step, because the MLPs' float-atomic weight gradients make two trajectories drift): the fused range's
    master weights, moments, 16-bit copy, average and its 16-bit copy must equal the separate launches' bit for bit,
    also on a step whose gradients overflow (huge loss scale: nothing may move)."""
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.ngp_engine import NgpConfig, NgpEngine
    from nerf_vo_amd.synthetic import make_sequence

    n, H, W = 8, 60, 80
    seq = make_sequence(n, H, W, device=device, scene_scale=0.2)
    c2w = opencv_to_opengl(seq["camera_extrinsics"])
    c2w[:, :3, 3] += 0.5
    c2w = c2w[:, :3, :4].contiguous()
    images = seq["frames_color"].permute(0, 2, 3, 1).contiguous()
    depths = seq["frames_depth"].permute(0, 2, 3, 1).contiguous()
    engines = {}
    for fuse in (True, False):
        torch.manual_seed(9)
        # (no extrinsics optimisation: its float-atomic pose gradient would let the two engines' rays drift apart)
        engines[fuse] = NgpEngine(NgpConfig(num_images=n, num_rays=512, capacity=1 << 18, fuse_grid_adam=fuse,
                                            optimize_extrinsics=False), device)
        engines[fuse].update_density_grid()
    fe, ue = engines[True], engines[False]
    plan = fe._fused_adam_plan()
    assert plan is not None and ue._fused_adam_plan() is None
    lo, hi = plan
    assert hi == fe.density_net.n_params and hi - lo > 0.5 * fe.density_net.n_params
    scale = torch.tensor([n, H, W], device=device)
    state = ("params", "exp_avg", "exp_avg_sq", "params_half", "params_ema", "params_ema_half")
    for it in range(6):
        overflow = it == 4
        for e in (fe, ue):
            e.cfg.loss_scale = 2.0 ** 60 if overflow else 128.0
        idx = torch.floor(torch.rand(512, 3, device=device) * scale).long()
        # identical state, identical random draws (the step takes its jitter / background from the global generator)
        if it > 0:
            for name in state:
                getattr(fe, name).copy_(getattr(ue, name))
            fe._ema_step_dev.copy_(ue._ema_step_dev)
        before = {name: getattr(ue, name).clone() for name in state} if it > 0 else None
        for e in (fe, ue):
            torch.manual_seed(100 + it)
            e.train_step(idx, seq["camera_intrinsics"], c2w, images, depths)
        torch.cuda.synchronize()
        assert bool(fe.skip_flag.item()) == bool(ue.skip_flag.item()) == overflow, (it, fe.skip_flag.item(), ue.skip_flag.item())
        assert int(fe._ema_step_dev.item()) == int(ue._ema_step_dev.item())
        for name in state:
            x, y = getattr(fe, name)[lo:hi], getattr(ue, name)[lo:hi]
            x = x.view(torch.int16) if x.dtype == torch.float16 else x.view(torch.int32)
            y = y.view(torch.int16) if y.dtype == torch.float16 else y.view(torch.int32)
            assert torch.equal(x, y), f"step {it}: {name} of the fused range differs ({int((x != y).sum())} words)"
            if overflow and before is not None:
                assert torch.equal(getattr(fe, name), before[name]), f"skipped step moved {name}"
        # the rest of the parameters went through the same launches in both engines: close, except where the float atomics
        # of the MLP dW leave a near-zero total with either sign (Adam turns that into +- lr): a handful of entries
        d = (fe.params[:lo] - ue.params[:lo]).abs()
        assert float((d > 1e-5 + 1e-3 * ue.params[:lo].abs()).float().mean()) < 0.02
    assert bool((fe.params[lo:hi] != 0).any()) and bool((fe.params_ema[lo:hi] != 0).any())


def test_graphed_step_matches_eager_step(device):
    """NgpConfig.graph_step: the step replayed from ONE hipGraph against the same launches issued eagerly.  Both engines
    start every step from the eager engine's state and the same generator seed (the MLPs' float-atomic weight gradients
    would let two free-running trajectories drift): the hash grid's range -- deterministic backward, Adam and weight
    average inside it -- must agree bit for bit, the rest closely; covered on the way: the first (eager) step of the graphed
    engine, density-grid refreshes between replays, a ray count the adaptive batch moves (a second capture), an overflowing
    step (nothing moves, the applied-step counter stands still) and the camera offsets' optimiser."""
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.ngp_engine import NgpConfig, NgpEngine
    from nerf_vo_amd.synthetic import make_sequence

    n, H, W = 8, 60, 80
    seq = make_sequence(n, H, W, device=device, scene_scale=0.2)
    c2w = opencv_to_opengl(seq["camera_extrinsics"])
    c2w[:, :3, 3] += 0.5
    c2w = c2w[:, :3, :4].contiguous()
    images = seq["frames_color"].permute(0, 2, 3, 1).contiguous()
    depths = seq["frames_depth"].permute(0, 2, 3, 1).contiguous()
    engines = {}
    for graphed in (True, False):
        torch.manual_seed(9)
        engines[graphed] = NgpEngine(NgpConfig(num_images=n, num_rays=512, capacity=1 << 16, graph_step=graphed,
                                               density_update_every=4, optimize_extrinsics=True,
                                               extrinsic_update_every=3), device)
    ge, ee = engines[True], engines[False]
    lo, hi = ge._fused_adam_plan()
    scale = torch.tensor([n, H, W], device=device)
    state = ("params", "exp_avg", "exp_avg_sq", "params_half", "params_ema", "params_ema_half", "pose_adjustment",
             "pose_exp_avg", "pose_exp_avg_sq", "density_grid", "bitfield", "_ema_step_dev", "_applied_dev", "_opt_dev",
             "d_corrections", "_cam_dev", "_cam_applied_dev")
    ray_counts = set()
    for it in range(11):
        overflow = it == 6
        for e in (ge, ee):
            e.cfg.loss_scale = 2.0 ** 60 if overflow else 128.0
        assert ge.rays_per_batch == ee.rays_per_batch
        R = ge.rays_per_batch
        ray_counts.add(R)
        idx = torch.floor(torch.rand(R, 3, device=device) * scale).long()
        if it > 0:
            for name in state:
                getattr(ge, name).copy_(getattr(ee, name))
        before = ee.params.clone()
        for e in (ge, ee):
            torch.manual_seed(100 + it)
            e.train_step(idx, seq["camera_intrinsics"], c2w, images, depths)
        torch.cuda.synchronize()
        assert bool(ge.skip_flag.item()) == bool(ee.skip_flag.item()) == overflow
        assert ge.applied_steps == ee.applied_steps == it + 1 - (1 if it >= 6 else 0)
        assert torch.equal(ge._ws["counts"][:R], ee._ws["counts"][:R])
        for name in ("params", "exp_avg", "exp_avg_sq", "params_half", "params_ema", "params_ema_half"):
            x, y = getattr(ge, name)[lo:hi], getattr(ee, name)[lo:hi]
            x = x.view(torch.int16) if x.dtype == torch.float16 else x.view(torch.int32)
            y = y.view(torch.int16) if y.dtype == torch.float16 else y.view(torch.int32)
            assert torch.equal(x, y), f"step {it}: {name} of the hash grid differs ({int((x != y).sum())} words)"
        assert torch.equal(ge._opt_dev, ee._opt_dev) and torch.equal(ge.density_grid, ee.density_grid)
        # the camera optimiser steps at the end of every third training step, with its own count and rate
        assert ge.cam_step == ee.cam_step == (it + 1) // 3 and ge._cam_window == ee._cam_window == (it + 1) % 3
        assert torch.equal(ge._cam_dev, ee._cam_dev) and torch.equal(ge._cam_applied_dev, ee._cam_applied_dev)
        if (it + 1) % 3 == 0:
            assert float(ge.d_corrections.abs().max()) == 0.0 == float(ee.d_corrections.abs().max())
        d = (ge.params[:lo] - ee.params[:lo]).abs()  # (MLP weights: float-atomic dW totals, see the test above)
        assert float((d > 1e-5 + 1e-3 * ee.params[:lo].abs()).float().mean()) < 0.02
        # (camera offsets: float-atomic gradient totals; a near-zero total of either sign moves an entry by +- lr)
        dp = (ge.pose_adjustment - ee.pose_adjustment).abs()
        assert int((dp > 1e-6 + 1e-3 * ee.pose_adjustment.abs()).sum()) <= 2
        assert torch.allclose(ge.losses.sum(0), ee.losses.sum(0), rtol=1e-4, atol=1e-7)
        if overflow:
            assert torch.equal(ge.params, before) and torch.equal(ee.params, before)
        else:
            assert not torch.equal(ge.params[lo:hi], before[lo:hi])
    assert len(ray_counts) >= 2 and len(ge._graphs) >= 2 and not ee._graphs
    assert bool((ge.pose_adjustment != 0).any())


def test_rows_past_the_live_count_do_not_influence_the_step(device):
    """The packed position buffers (x01 of the training slots, x01_m of the marched candidates) are written for the LIVE
    slots only; the rows between the live count and the next tile boundary (16 rows for the fused MLPs, 4096 for the
    launches that stop at n_live) keep whatever an earlier step left there, and the kernels DO evaluate them: a tile is
    computed whole, its dead rows carry dL/dout = 0.  The invariant that makes this safe has two halves: (1) those rows
    are always FINITE -- the buffers start zeroed and every writer clamps positions into [0, 1] (a NaN there would turn
    0 x NaN into a NaN weight gradient) -- and (2) nothing of the step depends on them.  This pins (2): with both buffers
    overwritten with fresh random positions before every step the run must be the run without them, bit for bit on the
    hash grid, no overflow flag."""
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.ngp_engine import NgpConfig, NgpEngine
    from nerf_vo_amd.synthetic import make_sequence

    n, H, W = 8, 60, 80
    seq = make_sequence(n, H, W, device=device, scene_scale=0.2)
    c2w = opencv_to_opengl(seq["camera_extrinsics"])
    c2w[:, :3, 3] += 0.5
    c2w = c2w[:, :3, :4].contiguous()
    images = seq["frames_color"].permute(0, 2, 3, 1).contiguous()
    depths = seq["frames_depth"].permute(0, 2, 3, 1).contiguous()
    engines = []
    for _ in range(2):
        torch.manual_seed(9)
        engines.append(NgpEngine(NgpConfig(num_images=n, num_rays=512, capacity=1 << 16, graph_step=False,
                                           density_update_every=4), device))
    clean, poisoned = engines
    lo, hi = clean._fused_adam_plan()
    scale = torch.tensor([n, H, W], device=device)
    state = ("params", "exp_avg", "exp_avg_sq", "params_half", "params_ema", "params_ema_half", "density_grid", "bitfield",
             "_ema_step_dev", "_applied_dev", "_opt_dev")
    hit = 0
    for it in range(8):
        assert clean.rays_per_batch == poisoned.rays_per_batch
        R = clean.rays_per_batch
        idx = torch.floor(torch.rand(R, 3, device=device) * scale).long()
        if it > 0:
            for name in state:  # (the MLPs' float-atomic weight gradients would let two free-running trajectories drift)
                getattr(poisoned, name).copy_(getattr(clean, name))
            for key in ("x01", "x01_m"):
                if poisoned._ws is not None and key in poisoned._ws:
                    poisoned._ws[key].copy_(torch.rand_like(poisoned._ws[key]))
                    hit += 1
        for e in (clean, poisoned):
            torch.manual_seed(100 + it)
            e.train_step(idx, seq["camera_intrinsics"], c2w, images, depths)
        torch.cuda.synchronize()
        assert int(poisoned.skip_flag.item()) == 0 == int(clean.skip_flag.item()), f"step {it}: overflow flag raised"
        for key in ("x01", "x01_m"):  # (1): whatever the step left in the buffers is finite and inside the unit cube
            buf = poisoned._ws[key]
            assert bool(torch.isfinite(buf).all()) and float(buf.min()) >= 0.0 and float(buf.max()) <= 1.0
        for name in ("params", "exp_avg", "exp_avg_sq", "params_half"):
            x, y = getattr(clean, name)[lo:hi], getattr(poisoned, name)[lo:hi]
            x = x.view(torch.int16) if x.dtype == torch.float16 else x.view(torch.int32)
            y = y.view(torch.int16) if y.dtype == torch.float16 else y.view(torch.int32)
            assert torch.equal(x, y), f"step {it}: {name} depends on the stale rows ({int((x != y).sum())} words)"
        assert bool(torch.isfinite(poisoned.params).all())
        assert torch.allclose(clean.losses.sum(0), poisoned.losses.sum(0), rtol=1e-4, atol=1e-7)
    assert hit >= 7, "the position buffers were never poisoned (workspace keys changed?)"


def test_scattered_density_refresh_past_the_warmup(device):
    """Past NgpConfig.density_warmup_steps a refresh evaluates cells / 4 per cascade drawn uniformly + as many among the
    occupied cells instead of every cell: cells no sample fell into decay by exactly `density_decay`
    (the EMA with a zero estimate), sampled ones take max(decayed, estimate) >= decayed, never-visible (negative) cells
    stay, the bitfield follows the oracle's threshold + max-pool of the new grid, and half the density-network
    evaluations of a full sweep are issued."""
    import ctypes as C

    from nerf_vo_amd import _lib
    from nerf_vo_amd.ngp_engine import CELLS, NgpConfig, NgpEngine
    from oracle import occgrid as O

    eng = NgpEngine(NgpConfig(num_images=4, num_rays=512, capacity=1 << 15, density_warmup_steps=0), device)
    L = eng.cfg.n_levels
    rng = np.random.default_rng(11)
    grid = (rng.random((L, CELLS), dtype=np.float32) ** 8) * 0.2
    grid[1, 5000:9000] = -1.0
    eng.density_grid.copy_(torch.from_numpy(grid.reshape(-1)).to(device))
    eng.step = 48
    lib = _lib.lib()
    lib.nvo_profile_enable(1)
    eng.update_density_grid()
    torch.cuda.synchronize()
    need = lib.nvo_profile_summary(None, 0)
    buf = C.create_string_buffer(int(need) + 16)
    lib.nvo_profile_summary(buf, len(buf))
    lib.nvo_profile_enable(0)
    launches = {ln.rsplit(",", 2)[0]: int(ln.rsplit(",", 2)[1]) for ln in buf.value.decode().strip().splitlines()}
    assert launches.get("occ_sample_cells") == 2 * (L * CELLS // 4) // (1 << 19) == launches.get("ngp_thickness_splat")
    assert "occ_cell_positions" not in launches
    new = eng.density_grid.cpu().numpy().reshape(L, CELLS)
    decayed = np.where(grid < 0, grid, grid * np.float32(eng.cfg.density_decay))
    assert (new[grid < 0] == grid[grid < 0]).all()
    assert (new >= decayed).all()
    untouched = new == decayed
    # 2 x (L x cells / 4) samples over L x cells cells, the second pass crowding into the occupied ones (and an estimate
    # below the decayed value changes nothing): well over a third of the cells, not all, stay at the decayed value
    assert 0.3 < untouched.mean() < 0.95, untouched.mean()
    assert (new > decayed).mean() > 0.05
    bf = eng.bitfield.cpu().numpy().reshape(L, -1)
    assert (bf == O.grid_to_bitfield(new, L, eng.cfg.occupancy_threshold)).all()
    # the same step and seed draw the same cells: a second engine from the same state lands on the same grid
    eng2 = NgpEngine(NgpConfig(num_images=4, num_rays=512, capacity=1 << 15, density_warmup_steps=0), device)
    eng2.set_params(eng.params.cpu())
    eng2.density_grid.copy_(torch.from_numpy(grid.reshape(-1)).to(device))
    eng2.step = 48
    eng2.update_density_grid()
    assert torch.equal(eng2.density_grid, eng.density_grid) and torch.equal(eng2.bitfield, eng.bitfield)


def test_rays_dropped_at_the_capacity_leave_the_losses_alone(device):
    """Rays whose samples do not fit the packed capacity are dropped by the march (count 0, slot range kept in the scan).
    They must not enter the losses as empty rays (black against their target): the loss sums of a batch with dropped rays
    equal those of the kept rays alone, rescaled by the ray counts of the two means; the scan's total still reports every
    sample the march found (what the adaptive batch measures)."""
    eng = _engine(device)
    cap = eng.cfg.capacity
    g = torch.Generator().manual_seed(4)
    R = 1024
    origins = ((torch.rand(R, 3, generator=g) - 0.5) * 0.6 + 0.5).to(device)
    directions = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(device)
    jitter = torch.rand(R, generator=g).to(device)
    gt_rgb, gt_depth = torch.rand(R, 3, generator=g).to(device), (torch.rand(R, generator=g) * 0.8).to(device)
    eng.bitfield.fill_(255)  # everything occupied: ~100 samples per ray, 1024 rays overflow 2^15 slots

    def run(sel):
        n = int(sel.numel())
        ws = eng._workspace(n, True)
        ws["origins"].copy_(origins[sel])
        ws["directions"].copy_(directions[sel])
        ws["directions_norm"].fill_(1.0)
        ws["gt_rgb"].copy_(gt_rgb[sel])
        ws["gt_depth"].copy_(gt_depth[sel])
        eng.forward_backward(ws, jitter[sel].contiguous(), has_depth=True)
        torch.cuda.synchronize()
        return ws["counts"][:n].clone(), ws["offsets"][:n + 1].clone(), eng.losses.sum(0)[:2].clone(), eng.grads.clone()

    counts, offsets, loss_all, grads_all = run(torch.arange(R, device=device))
    found = (offsets[1:] - offsets[:-1])
    dropped = (counts == 0) & (found > 0)
    assert int(dropped.sum()) > 100 and int(offsets[-1]) > cap >= int(counts.sum()) > 0.9 * cap
    kept = torch.nonzero(~dropped).flatten()
    counts_k, _, loss_kept, grads_kept = run(kept)
    assert torch.equal(counts_k, counts[kept])
    ratio = kept.numel() / R
    assert torch.allclose(loss_all, loss_kept * ratio, rtol=2e-3, atol=1e-8), (loss_all, loss_kept * ratio)
    ref = grads_kept * ratio
    assert float((grads_all - ref).abs().sum()) <= 2e-2 * float(ref.abs().sum())


@pytest.mark.parametrize("compact", [False, True], ids=["plain", "compact"])
def test_step_on_an_empty_occupancy_grid(device, compact):
    """No occupied cell at all (a grid decayed to nothing, a camera looking out of the scene): the march finds no sample, every
    ray renders its background, the packed batch is empty -- the step must run (no kernel is handed a zero-sized grid it
    cannot launch), report finite losses, leave every gradient exactly zero and let the optimiser and the adaptive ray
    batch pass through; inference of the same rays returns the background with zero accumulation."""
    eng = _engine(device, compact_training=compact)
    g = torch.Generator().manual_seed(8)
    R = 256
    origins = ((torch.rand(R, 3, generator=g) - 0.5) * 0.6 + 0.5).to(device)
    directions = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(device)
    jitter = torch.rand(R, generator=g).to(device)
    eng.bitfield.zero_()
    ws = eng._workspace(R, True)
    ws["origins"].copy_(origins)
    ws["directions"].copy_(directions)
    ws["directions_norm"].fill_(1.0)
    ws["gt_rgb"].copy_(torch.rand(R, 3, generator=g).to(device))
    ws["gt_depth"].copy_((torch.rand(R, generator=g) * 0.8).to(device))
    before = eng.params.clone()
    eng.forward_backward(ws, jitter, has_depth=True)
    torch.cuda.synchronize()
    assert int(ws["totals"][0]) == 0 and int(ws["counts"][:R].sum()) == 0
    assert bool(torch.isfinite(eng.losses).all())
    assert float(eng.grads.abs().max()) == 0.0
    eng.optimizer_step(camera_update=False)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(eng.params).all()) and int(eng.skip_flag.item()) == 0
    # the hash table (zero gradient, zero moments, no decay) stays where it was bit for bit; the MLP weights see their L2
    # decay as the only gradient, which Adam turns into one learning-rate step
    nd, nden = eng.n_density_mlp, eng.density_net.n_params
    assert torch.equal(eng.params[nd:nden], before[nd:nden])
    assert float((eng.params - before).abs().max()) <= eng.cfg.lr * 1.001
    out = eng.render_rays(origins, directions, torch.ones(R, device=device))
    torch.cuda.synchronize()
    assert float(out["accumulation"].abs().max()) == 0.0 and bool(torch.isfinite(out["rgb"]).all())


def test_render_splits_bundles_that_overflow_the_capacity(device):
    """NgpEngine.render_rays shades every sample the march finds; a bundle with more samples than the packed capacity
    must come out exactly as from an engine whose capacity holds it whole (rendered in halves, no ray dropped -- dropped
    rays used to come back black), and switching between training and rendering keeps both workspaces."""
    from nerf_vo_amd.ngp_engine import NgpConfig, NgpEngine

    small = _engine(device, render_capacity=1 << 15)         # inference capacity 2^15
    big = NgpEngine(NgpConfig(num_images=4, capacity=1 << 15, render_capacity=1 << 18), device)
    big.set_params(small.params.cpu())
    for e in (small, big):
        e.bitfield.fill_(255)                                 # every cell occupied: hundreds of samples per ray
    g = torch.Generator().manual_seed(12)
    R = 256
    origins = ((torch.rand(R, 3, generator=g) - 0.5) * 0.6 + 0.5).to(device)
    directions = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(device)
    dnorm = torch.ones(R, device=device)
    ws_train = small._workspace(64, True)
    a = small.render_rays(origins, directions, dnorm)
    b = big.render_rays(origins, directions, dnorm)
    total = int(big._ws["offsets"][-1].item())
    assert small.cfg.render_capacity * 4 < total <= big.cfg.render_capacity
    for k in ("rgb", "depth", "accumulation"):
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    assert float(b["accumulation"].min()) > 0.0
    assert small._workspace(64, True) is ws_train and small._wss[False] is not None


def test_render_in_rounds_matches_the_single_pass(device):
    """NgpEngine.render_rays with a transmittance threshold (two rounds, nvo_occ_march_resume + the compositing kernel's
    carried optical depth): with a threshold nothing can reach (1e-30) every ray goes through both rounds and the result
    must equal the single pass up to the order of the float sums; with the reference's 1e-4 the images differ by less than
    the light the cut-off tail could still have contributed, and far fewer samples are shaded."""
    eng = _engine(device, render_capacity=1 << 18)
    rng = np.random.default_rng(3)
    grid = (rng.random((eng.cfg.n_levels, 128 ** 3), dtype=np.float32) ** 4) * 0.3
    eng.density_grid.copy_(torch.from_numpy(grid.reshape(-1)).to(device))
    from nerf_vo_amd.engine import _call
    from nerf_vo_amd.tinycudann.modules import _ptr, _stream
    _call("nvo_occ_update", _stream(device), eng.cfg.n_levels, _ptr(eng.density_grid), None, 0.95, 0.01,
          _ptr(eng.bitfield), _ptr(eng._scratch8))
    g = torch.Generator().manual_seed(5)
    R = 700
    origins = ((torch.rand(R, 3, generator=g) - 0.5) * 0.6 + 0.5).to(device)
    directions = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(device)
    dnorm = torch.ones(R, device=device)
    # much denser than the random initialisation (both layers of the density MLP x 16: pre-activations of +-70, i.e.
    # densities up to 1e30 -- the compositing caps a sample's optical depth, so weights still sum to at most 1), so that
    # rays do become opaque on their way
    flat = eng.params.clone()
    flat[:eng.n_density_mlp] *= 16.0
    eng.set_params(flat.cpu())
    s0 = eng.render_shaded_total
    single = eng.render_rays(origins, directions, dnorm, 0.0)
    n_single = eng.render_shaded_total - s0
    rounds = eng.render_rays(origins, directions, dnorm, 1e-30)
    # (in two rounds; only rays already blacker than 1e-30 -- a capped sample of optical depth 128 does that -- stop early)
    assert n_single < eng.render_shaded_total - s0 <= 2 * n_single
    for k in ("rgb", "depth", "accumulation"):
        assert torch.allclose(rounds[k], single[k], rtol=1e-4, atol=2e-5), k
    s1 = eng.render_shaded_total
    cut = eng.render_rays(origins, directions, dnorm, 1e-4)
    n_cut = eng.render_shaded_total - s1
    assert float((cut["rgb"] - single["rgb"]).abs().max()) < 2e-3
    assert float((cut["accumulation"] - single["accumulation"]).abs().max()) < 2e-3
    assert float(single["accumulation"].max()) <= 1.0 + 1e-4   # (2.7 before the cap; __expf rounding is what is left)
    opaque = float((single["accumulation"] > 0.999).float().mean())
    assert opaque > 0.2 and n_cut < 0.8 * n_single, (opaque, n_cut, n_single)
