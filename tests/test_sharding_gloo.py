"""N > 1 path on CPU: world_size-2 gloo run of the batch partition and of the
pipelined gains all-gather (the RCCL path of bench.py uses the same object)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sip_optimal_control_amd.sharding import GainsAllGather, shard_range


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


def _fake_gains(rank, step, local_batch, gains_len):
    base = np.arange(local_batch * gains_len, dtype=np.float64).reshape(local_batch, gains_len)
    return torch.from_numpy(base + 1000.0 * rank + 1e6 * step)


def _worker(rank, world, port, local_batch, gains_len, steps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ag = GainsAllGather(local_batch, gains_len, torch.float64, "cpu")
        results = []
        for i in range(steps):
            buf = ag.acquire(i)
            buf.copy_(_fake_gains(rank, i, local_batch, gains_len))  # stands for the kernel
            results.append(ag.launch(i))
            if i >= 1:  # pipelined: check the previous sweep's gather after launching this one
                ag._work[ag.slot(i - 1)] and ag._work[ag.slot(i - 1)].wait()
                want = torch.cat([_fake_gains(r, i - 1, local_batch, gains_len) for r in range(world)])
                assert torch.equal(results[i - 1], want)
        ag.finish()
        want = torch.cat([_fake_gains(r, steps - 1, local_batch, gains_len) for r in range(world)])
        assert torch.equal(results[-1], want)
        out.put((rank, True, ""))
    except Exception as exc:  # pragma: no cover
        out.put((rank, False, repr(exc)))
    finally:
        dist.destroy_process_group()


def test_gains_all_gather_world2_gloo():
    world, local_batch, gains_len, steps = 2, 5, 52, 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, local_batch, gains_len, steps, out))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, msg in results:
        assert ok, f"rank {rank}: {msg}"
