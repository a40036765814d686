"""CPU: the N>1 path (one process per rank, batch sharding + ONE all-gather of the joints) over gloo, world_size 2."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, results):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from sceneego_amd import dist as sdist
    r, w, dev = sdist.init_from_env(backend="gloo", device_type="cpu")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    # every frame's "joints" encode its global frame index, so ordering errors are visible
    lo, hi = sdist.shard_range(total, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 15, 3).contiguous()
    if total % world == 0:
        full = sdist.all_gather_joints(local)
    else:
        full = sdist.all_gather_joints_ragged(local, total)
    sdist.barrier()
    t = sdist.max_over_ranks(float(rank + 1), dev)
    results[rank] = (full[:, 0, 0].tolist(), t, (lo, hi))
    dist.destroy_process_group()


def _run(total):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, total, results), nprocs=world, join=True)
    return dict(results)


def test_all_gather_joints_even():
    res = _run(8)
    for rank in (0, 1):
        got, tmax, rng = res[rank]
        assert got == [float(i) for i in range(8)]
        assert tmax == 2.0
    assert res[0][2] == (0, 4) and res[1][2] == (4, 8)


def test_all_gather_joints_ragged():
    res = _run(7)
    for rank in (0, 1):
        assert res[rank][0] == [float(i) for i in range(7)]
    assert res[0][2] == (0, 4) and res[1][2] == (4, 7)


def test_shard_range_partitions():
    from sceneego_amd import dist as sdist
    for total in (1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [sdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_forward_worker(rank, world, port, per_rank, results):
    """What bench.py does per step, with the network replaced by a frame-wise function: rank-seeded inputs (1234 + rank),
    a per-frame 'forward', ONE all-gather, result identical on every rank and ordered by rank."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from sceneego_amd import dist as sdist
    r, w, dev = sdist.init_from_env(backend="gloo", device_type="cpu")
    g = torch.Generator().manual_seed(1234 + rank)                    # bench.py: device_inputs(batch, rank, ...)
    img = torch.randn((per_rank, 3, 8, 8), generator=g)
    depth = torch.rand((per_rank, 16, 20), generator=g) * 2.7 + 0.3
    joints = torch.stack([img.mean(dim=(2, 3)), depth.mean(dim=(1, 2)).unsqueeze(1).expand(-1, 3)], dim=1)   # [b,2,3] per frame
    joints = joints.repeat(1, 8, 1)[:, :15]                           # [b,15,3]
    full = sdist.all_gather_joints(joints.contiguous())
    sdist.barrier()
    results[rank] = full
    dist.destroy_process_group()


def test_sharded_fake_forward_rank_seeds_and_gather_order():
    world, per_rank = 2, 4
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_fake_forward_worker, args=(world, _free_port(), per_rank, results), nprocs=world, join=True)
    res = dict(results)
    # single-process restatement: rank r's frames come from seed 1234 + r and land at rows [r*per_rank, (r+1)*per_rank)
    want = []
    for r in range(world):
        g = torch.Generator().manual_seed(1234 + r)
        img = torch.randn((per_rank, 3, 8, 8), generator=g)
        depth = torch.rand((per_rank, 16, 20), generator=g) * 2.7 + 0.3
        j = torch.stack([img.mean(dim=(2, 3)), depth.mean(dim=(1, 2)).unsqueeze(1).expand(-1, 3)], dim=1)
        want.append(j.repeat(1, 8, 1)[:, :15])
    want = torch.cat(want)
    assert tuple(res[0].shape) == (world * per_rank, 15, 3)
    assert torch.equal(res[0], want) and torch.equal(res[1], want)
    assert not torch.equal(want[:per_rank], want[per_rank:])          # the two ranks really had different inputs


def test_bench_refuses_world_size_mismatch():
    """bench.py must never print a 1-GPU number as an N-GPU line: under a launcher whose WORLD_SIZE differs from --gpus it exits
    non-zero before touching the GPU."""
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode != 0
    assert "refusing" in r.stderr


def test_eight_ranks_gather_order_and_thread_pinning():
    """configs[3] runs 8 ranks per node: the same init + sharding + ONE all-gather with world_size 8 (gloo), every rank pinned to
    its own share of the host cores."""
    world, per_rank = 8, 1
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_fake_forward_worker, args=(world, _free_port(), per_rank, results), nprocs=world, join=True)
    res = dict(results)
    assert sorted(res) == list(range(world))
    for r in range(1, world):
        assert torch.equal(res[r], res[0])
    assert tuple(res[0].shape) == (world * per_rank, 15, 3)
    assert len({tuple(res[0][i].flatten().tolist()) for i in range(world)}) == world      # eight different shards, in rank order
    g = torch.Generator().manual_seed(1234 + 5)
    img = torch.randn((per_rank, 3, 8, 8), generator=g)
    assert torch.equal(res[0][5, 0], img.mean(dim=(2, 3))[0])


def test_init_without_master_port_exits_with_a_message():
    """No fixed default rendezvous port: a rank started with WORLD_SIZE > 1 and no MASTER_PORT stops with exit code 4."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_PORT", "MASTER_ADDR")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    code = ("import sys; sys.path.insert(0, %r); from sceneego_amd import dist as d; d.init_from_env(backend='gloo', device_type='cpu')"
            % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 4, (r.returncode, r.stderr[-300:])
    assert "MASTER_PORT" in r.stderr


def test_init_timeout_reports_and_exits():
    """A ring that cannot form (the peer never arrives) ends the rank with exit code 4 and the backend's message within the
    configured timeout instead of hanging."""
    import subprocess
    env = dict(os.environ, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               SCENEEGO_DIST_TIMEOUT_S="3")
    code = ("import sys; sys.path.insert(0, %r); from sceneego_amd import dist as d; d.init_from_env(backend='gloo', device_type='cpu')"
            % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 4, (r.returncode, r.stderr[-300:])
    assert "process group failed" in r.stderr
