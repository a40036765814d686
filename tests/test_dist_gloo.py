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
