"""Multi-process (gloo, world_size 2, CPU) test of the only collective on the path: the sum
all-reduce of the flat gradient buffer, plus ray sharding helpers."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nerf_vo_amd.parallel import GradientAllReduce

    red = GradientAllReduce(dist, bucket_numel=1000)
    g = torch.Generator().manual_seed(100 + rank)
    n = 10_007  # not a multiple of the bucket size
    grad = torch.randn(n, generator=g)
    mine = grad.clone()
    red(grad)
    # every rank must hold the same sum; rank 0 checks it against a locally recomputed reference
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ref = sum(gathered)
    ok = torch.allclose(grad, ref, atol=1e-6)
    # partial reduction: only one segment is reduced, the rest keeps the local value
    grad2 = mine.clone()
    red(grad2, segments=[(100, 2500)])
    ok = ok and torch.allclose(grad2[100:2600], ref[100:2600], atol=1e-6) and torch.equal(grad2[:100], mine[:100]) \
        and torch.equal(grad2[2600:], mine[2600:])
    # fp16-compressed exchange: same sum up to fp16 rounding of each rank's contribution
    red16 = GradientAllReduce(dist, bucket_numel=3000, compress="fp16")
    grad3 = mine.clone()
    red16(grad3, segments=[(0, 5000), (5000, n - 5000)])
    ref16 = sum(t.half().float() for t in gathered)
    ok = ok and torch.allclose(grad3, ref16, atol=2e-2, rtol=2e-3)
    # bf16-compressed exchange (what bench.py uses): 8 significant bits, fp32's exponent range -- tiny values survive
    redb = GradientAllReduce(dist, compress="bf16")
    tiny = mine.clone() * 1e-12
    back = redb(tiny.clone(), keep_half=True)
    refb = sum((t * 1e-12).to(torch.bfloat16).float() for t in gathered)
    ok = ok and back.dtype == torch.bfloat16 and torch.allclose(back.float(), refb, rtol=2e-2, atol=0.0) \
        and float((back.float() == 0).float().mean()) < 1e-3
    # asynchronous form (the multi-GPU step overlaps the fields reduction with the next sampling prefix): handles out,
    # same sum after wait(); compressed variant through the caller-owned 2-byte buffer
    grad4 = mine.clone()
    handles = red(grad4, segments=[(0, n)], async_op=True)
    ok = ok and len(handles) > 0
    red.wait(handles)
    ok = ok and torch.allclose(grad4, ref, atol=1e-6)
    half = mine.to(torch.bfloat16)
    handles = redb.reduce_half(mine, half, [(0, 4000), (6000, n - 6000)], already_cast=True, async_op=True)
    redb.wait(handles)
    refh = sum(t.to(torch.bfloat16).float() for t in gathered)
    ok = ok and torch.allclose(half[:4000].float(), refh[:4000], rtol=2e-2, atol=1e-2) \
        and torch.equal(half[4000:6000], mine[4000:6000].to(torch.bfloat16))
    # MAX reduction of the occupancy-grid density estimates: identical on every rank, elementwise maximum
    dens = mine.abs().clone()
    red.reduce_max(dens)
    ok = ok and torch.equal(dens, torch.stack([t.abs() for t in gathered]).max(dim=0).values)
    # sharded exchange: reduce-scatter of a wire buffer of `world` chunks (+ flag slots), all-gather of the owners' slices
    # IN PLACE (the shard is a slice of the full buffer, as the optimiser writes it)
    reds = GradientAllReduce(dist, compress="bf16", shard_optimizer=True)
    per, pad = 1000, 8
    wire = torch.zeros(world * (per + pad), dtype=torch.bfloat16)
    for c in range(world):
        wire[c * (per + pad):c * (per + pad) + per] = mine[c * per:(c + 1) * per].to(torch.bfloat16)
        wire[c * (per + pad) + per:(c + 1) * (per + pad)] = 1.0 if rank == 1 else 0.0  # rank 1 "overflowed"
    out = torch.zeros(per + pad, dtype=torch.bfloat16)
    reds.wait(reds.reduce_scatter(out, wire, async_op=True))
    ref_chunk = sum(t[rank * per:(rank + 1) * per].to(torch.bfloat16).float() for t in gathered)
    ok = ok and torch.allclose(out[:per].float(), ref_chunk, rtol=2e-2, atol=1e-2) and float(out[per]) == 1.0
    full = torch.zeros(world * per, dtype=torch.bfloat16)
    full[rank * per:(rank + 1) * per] = out[:per]
    reds.wait(reds.all_gather(full, full[rank * per:(rank + 1) * per], async_op=True))
    chunks = [torch.zeros(per, dtype=torch.bfloat16) for _ in range(world)]
    dist.all_gather(chunks, out[:per].clone())
    ok = ok and torch.equal(full, torch.cat(chunks)) and reds.world == world and reds.rank == rank
    try:
        GradientAllReduce(dist, shard_optimizer=True)
        ok = False  # the sharded optimiser needs a 2-byte wire format
    except ValueError:
        pass
    torch.save({"ok": bool(ok)}, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_all_reduce_gloo_world2(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert torch.load(tmp_path / f"r{r}.pt")["ok"], f"rank {r} saw a wrong reduction"


def test_single_process_world_is_identity():
    from nerf_vo_amd.parallel import GradientAllReduce, shard_ray_count

    class _NoDist:
        @staticmethod
        def is_initialized():
            return False

    red = GradientAllReduce(_NoDist())
    x = torch.arange(10.0)
    y = x.clone()
    red(y)
    assert torch.equal(x, y)
    assert [shard_ray_count(4096, 8, r) for r in range(8)] == [512] * 8
    assert sum(shard_ray_count(4099, 8, r) for r in range(8)) == 4099
