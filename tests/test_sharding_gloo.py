"""N > 1 path on CPU: world_size-2 and world_size-4 gloo runs of the batch partition and of the
pipelined, chunked gains all-gather (the RCCL path of bench.py uses the same object), including
chunks that become ready in a different order on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sip_optimal_control_amd.sharding import GainsAllGather, chunk_bounds, shard_range


def test_shard_range_partitions_everything():
    for total in (1, 7, 8, 4096, 32768, 32771):
        for world in (1, 2, 4, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(32768, 8, 3) == (3 * 4096, 4 * 4096)  # BASELINE config 5
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_chunk_bounds_and_addressing():
    assert chunk_bounds(4096, 8) == [(512 * c, 512 * (c + 1)) for c in range(8)]  # 8 x 512
    assert chunk_bounds(5, 2) == [(0, 3), (3, 5)]
    with pytest.raises(ValueError):
        chunk_bounds(4, 5)
    ag = GainsAllGather(7, 3, torch.float64, "cpu", chunks=3)   # one rank: world == 1
    assert ag.world == 1 and [ag.rows_of(0, c) for c in range(3)] == [(0, 3), (3, 5), (5, 7)]
    buf = ag.acquire(0)
    buf.copy_(torch.arange(21, dtype=torch.float64).reshape(7, 3))
    out = ag.launch(0)
    assert torch.equal(out, buf) and torch.equal(ag.rank_major(out), buf)
    assert torch.equal(ag.problem(out, 4), buf[4])


def _fake_gains(rank, step, local_batch, gains_len):
    base = np.arange(local_batch * gains_len, dtype=np.float64).reshape(local_batch, gains_len)
    return torch.from_numpy(base + 1000.0 * rank + 1e6 * step)


def _worker(rank, world, port, local_batch, gains_len, steps, chunks, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ag = GainsAllGather(local_batch, gains_len, torch.float64, "cpu", chunks=chunks, mode=mode)
        results = []
        for i in range(steps):
            buf = ag.acquire(i)
            src = _fake_gains(rank, i, local_batch, gains_len)
            # the chunks of a sweep become ready in an order of this rank's own (rotated by rank and
            # step): the gathers still pair up, because they are ISSUED in chunk order on every rank
            order = [(c + rank + i) % chunks for c in range(chunks)]
            if rank % 2:
                order.reverse()
            for c in order:
                lo, hi = ag.bounds[c]
                buf[lo:hi].copy_(src[lo:hi])                     # stands for the kernel of chunk c
                ag.mark_ready(i, c)
            results.append(ag.launch(i))
            if i >= 1:  # pipelined: check the previous sweep's gather after launching this one
                ag.wait(i - 1)
                want = torch.cat([_fake_gains(r, i - 1, local_batch, gains_len) for r in range(world)])
                assert torch.equal(ag.rank_major(results[i - 1]), want)
                g = (world - 1) * local_batch + local_batch // 2
                assert torch.equal(ag.problem(results[i - 1], g), want[g])
        ag.finish()
        want = torch.cat([_fake_gains(r, steps - 1, local_batch, gains_len) for r in range(world)])
        assert torch.equal(ag.rank_major(results[-1]), want)
        out.put((rank, True, ""))
    except Exception as exc:  # pragma: no cover
        out.put((rank, False, repr(exc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allgather", "mesh"])
@pytest.mark.parametrize("world,local_batch,chunks", [(2, 5, 1), (2, 8, 4), (4, 7, 3), (4, 16, 8)])
def test_gains_all_gather_gloo(world, local_batch, chunks, mode):
    gains_len, steps = 52, 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, local_batch, gains_len, steps, chunks, mode, out))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, msg in results:
        assert ok, f"rank {rank}: {msg}"
