"""Batch sharding over the GPUs of one node and the all-gather of the gains.

Problem instances are independent (no shared state between the reference's
`LQR`/`Workspace` pairs), so the batch is block-partitioned over ranks with no
data-path collective; the one exchange step is the all-gather of the feedback
gains (K, k) -- one RCCL all-gather per sweep over xGMI (torch.distributed
backend "nccl" is RCCL on ROCm), pipelined against the next sweep's compute:
sweep i's gather runs on a side stream while sweep i+1 computes, with the gains
double-buffered.  On CPU tensors (gloo, tests) the same object degrades to
synchronous calls.
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous block partition: rank g owns [lo, hi)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GainsAllGather:
    """Double-buffered, stream-pipelined all-gather of per-rank gains."""

    def __init__(self, local_batch, gains_len, dtype, device, group=None, depth=2):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device(device)
        self.depth = depth
        self.on_gpu = self.device.type == "cuda"
        self.local = [torch.empty(local_batch, gains_len, dtype=dtype, device=self.device)
                      for _ in range(depth)]
        self.gathered = [torch.empty(self.world * local_batch, gains_len, dtype=dtype, device=self.device)
                         for _ in range(depth)]
        self._done = [None] * depth
        self._work = [None] * depth
        self.comm_stream = torch.cuda.Stream(self.device) if self.on_gpu else None

    def slot(self, i):
        return i % self.depth

    def acquire(self, i):
        """Local gains buffer for sweep i, once the gather that last read it is done."""
        s = self.slot(i)
        if self.on_gpu:
            if self._done[s] is not None:
                torch.cuda.current_stream(self.device).wait_event(self._done[s])
        elif self._work[s] is not None:
            self._work[s].wait()
            self._work[s] = None
        return self.local[s]

    def launch(self, i):
        """Start the all-gather of sweep i's gains (after the compute enqueued so far)."""
        s = self.slot(i)
        if self.world == 1:
            self.gathered[s].copy_(self.local[s])
            return self.gathered[s]
        if self.on_gpu and dist.get_backend(self.group) != "nccl":
            # rehearsal backends (gloo) take host tensors: stage through the CPU, synchronously
            torch.cuda.current_stream(self.device).synchronize()
            host = torch.empty(self.gathered[s].shape, dtype=self.gathered[s].dtype)
            dist.all_gather_into_tensor(host, self.local[s].cpu(), group=self.group)
            self.gathered[s].copy_(host)
        elif self.on_gpu:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                dist.all_gather_into_tensor(self.gathered[s], self.local[s], group=self.group)
                done = torch.cuda.Event()
                done.record(self.comm_stream)
            self._done[s] = done
        else:
            self._work[s] = dist.all_gather_into_tensor(self.gathered[s], self.local[s],
                                                        group=self.group, async_op=True)
        return self.gathered[s]

    def finish(self):
        """Block (stream-wise on GPU) until every outstanding gather has landed."""
        for s in range(self.depth):
            if self.on_gpu:
                if self._done[s] is not None:
                    torch.cuda.current_stream(self.device).wait_event(self._done[s])
            elif self._work[s] is not None:
                self._work[s].wait()
                self._work[s] = None
