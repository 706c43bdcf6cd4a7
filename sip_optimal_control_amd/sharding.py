"""Batch sharding over the GPUs of one node and the all-gather of the gains.

Problem instances are independent (no shared state between the reference's
`LQR`/`Workspace` pairs), so the batch is block-partitioned over ranks with no
data-path collective; the one exchange step is the all-gather of the feedback
gains (K, k) over RCCL / xGMI (torch.distributed backend "nccl" is RCCL on
ROCm).

The exchange of sweep i is cut into `chunks` contiguous problem ranges.  Chunk c
is gathered as soon as ITS gains are final (`mark_ready(i, c)`: an event behind
the launch that produced them -- the whole sweep, or one of several sub-batch
launches), on a side stream, while the rest of sweep i and all of sweep i + 1
compute; the gains are double-buffered, and a buffer is rewritten only once
every chunk gather that reads it has drained.  Collectives have to be issued in
one order on every rank, so the chunk gathers of a sweep are issued in chunk
order 0 .. chunks-1 whatever the order in which the chunks became ready (each
waits for its own event).  On CPU tensors (gloo; the tests) the same object
runs with asynchronous work handles instead of streams.

Two forms of the exchange (`mode`):
  "allgather"  one `all_gather_into_tensor` per chunk (RCCL picks the algorithm: a ring pushes every
               shard through one xGMI link seven times);
  "mesh"       per chunk one `batch_isend_irecv` of world-1 sends of the local slice and world-1
               receives straight into the gathered buffer: every rank talks to every peer at once, so
               the seven xGMI links of a GPU each carry one copy of its slice (full
               mesh, shard_bytes / link bandwidth), in one RCCL group per chunk, no extra memory.
Both fill the same layout.

Layout of the gathered gains: chunk-major, `gathered[c][r]` = rows of rank r's
chunk c (what one ncclAllGather per chunk produces without any re-packing);
`rows_of(rank, chunk)` / `problem(global_index)` address it.
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


def chunk_bounds(local_batch, chunks):
    """[lo, hi) of every chunk of a shard: equal sizes (the last ones one smaller when it does not
    divide); every rank must use the same (local_batch, chunks)."""
    if chunks < 1 or chunks > max(1, local_batch):
        raise ValueError("chunks must be in 1 .. local_batch")
    return [shard_range(local_batch, chunks, c) for c in range(chunks)]


class GainsAllGather:
    """Double-buffered, stream-pipelined, chunked all-gather of per-rank gains."""

    MODES = ("allgather", "mesh")

    def __init__(self, local_batch, gains_len, dtype, device, group=None, depth=2, chunks=1, mode="allgather"):
        if mode not in self.MODES:
            raise ValueError(f"mode must be one of {self.MODES}")
        self.mode = mode
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device(device)
        self.depth = depth
        self.on_gpu = self.device.type == "cuda"
        self.local_batch, self.gains_len = local_batch, gains_len
        self.bounds = chunk_bounds(local_batch, chunks)
        self.chunks = chunks
        self.local = [torch.empty(local_batch, gains_len, dtype=dtype, device=self.device)
                      for _ in range(depth)]
        # chunk-major: chunk c occupies rows [world * lo_c, world * hi_c), rank-major inside
        self.gathered = [torch.empty(self.world * local_batch, gains_len, dtype=dtype, device=self.device)
                         for _ in range(depth)]
        self._ready = [[None] * chunks for _ in range(depth)]  # GPU: event behind the chunk's producer
        self._done = [None] * depth                            # GPU: event behind the slot's last chunk gather
        self._work = [[] for _ in range(depth)]                # CPU: async work handles of the slot
        self.comm_stream = torch.cuda.Stream(self.device) if self.on_gpu else None

    def slot(self, i):
        return i % self.depth

    # ---- addressing of the gathered buffer ------------------------------------------------------
    def rows_of(self, rank, chunk):
        """Row range of rank `rank`'s chunk `chunk` inside a gathered buffer."""
        lo, hi = self.bounds[chunk]
        base = self.world * lo + rank * (hi - lo)
        return base, base + (hi - lo)

    def problem(self, gathered, global_index):
        """Gains row of global problem `global_index` (rank-major block partition of the batch)."""
        rank, local = divmod(global_index, self.local_batch)
        for c, (lo, hi) in enumerate(self.bounds):
            if lo <= local < hi:
                return gathered[self.rows_of(rank, c)[0] + (local - lo)]
        raise IndexError(global_index)

    def rank_major(self, gathered):
        """A [world * local_batch, gains_len] copy in plain rank-major order (tests, consumers that
        want the layout of one whole-shard all-gather)."""
        if self.chunks == 1:
            return gathered.clone()
        parts = [gathered[slice(*self.rows_of(r, c))] for r in range(self.world) for c in range(self.chunks)]
        return torch.cat(parts)

    # ---- the pipeline --------------------------------------------------------------------------
    def acquire(self, i):
        """Local gains buffer for sweep i, once every gather that last read it is done."""
        s = self.slot(i)
        if self.on_gpu:
            if self._done[s] is not None:
                torch.cuda.current_stream(self.device).wait_event(self._done[s])
        else:
            for w in self._work[s]:
                w.wait()
            self._work[s] = []
        self._ready[s] = [None] * self.chunks
        return self.local[s]

    def mark_ready(self, i, chunk=None):
        """The gains of chunk `chunk` of sweep i (None: of every chunk) are final behind the work
        enqueued so far on the current stream.  May be called in any chunk order."""
        s = self.slot(i)
        which = range(self.chunks) if chunk is None else (chunk,)
        ev = None
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        for c in which:
            self._ready[s][c] = ev if self.on_gpu else True

    def launch(self, i):
        """Issue the chunk gathers of sweep i, in chunk order, each behind its own readiness.
        Chunks not marked ready explicitly are taken to be ready behind the work enqueued so far."""
        s = self.slot(i)
        if any(r is None for r in self._ready[s]):
            missing = [c for c, r in enumerate(self._ready[s]) if r is None]
            for c in missing:
                self.mark_ready(i, c)
        out = self.gathered[s]
        if self.world == 1:
            out.copy_(self.local[s])
            return out
        if self.on_gpu and dist.get_backend(self.group) != "nccl":
            # rehearsal backends (gloo) take host tensors: stage through the CPU, synchronously
            torch.cuda.current_stream(self.device).synchronize()
            for c, (lo, hi) in enumerate(self.bounds):
                host = torch.empty(self.world * (hi - lo), self.gains_len, dtype=out.dtype)
                mine = self.local[s][lo:hi].cpu()
                if self.mode == "mesh":
                    for w in self._mesh_ops(mine, host.view(self.world, hi - lo, self.gains_len), own_copy=True):
                        w.wait()
                else:
                    dist.all_gather_into_tensor(host, mine, group=self.group)
                out[self.world * lo:self.world * hi].copy_(host)
        elif self.on_gpu:
            with torch.cuda.stream(self.comm_stream):
                for c, (lo, hi) in enumerate(self.bounds):
                    self.comm_stream.wait_event(self._ready[s][c])
                    dst = out[self.world * lo:self.world * hi]
                    if self.mode == "mesh":
                        for w in self._mesh_ops(self.local[s][lo:hi], dst.view(self.world, hi - lo, self.gains_len),
                                                own_copy=True):
                            w.wait()   # stream-wise: the comm stream continues behind RCCL's group
                    else:
                        dist.all_gather_into_tensor(dst, self.local[s][lo:hi], group=self.group)
                done = torch.cuda.Event()
                done.record(self.comm_stream)
            self._done[s] = done
        else:
            for c, (lo, hi) in enumerate(self.bounds):
                dst = out[self.world * lo:self.world * hi]
                if self.mode == "mesh":
                    self._work[s].extend(self._mesh_ops(self.local[s][lo:hi],
                                                        dst.view(self.world, hi - lo, self.gains_len), own_copy=True))
                else:
                    self._work[s].append(dist.all_gather_into_tensor(dst, self.local[s][lo:hi], group=self.group,
                                                                     async_op=True))
        return out

    def _mesh_ops(self, mine, parts, own_copy):
        """The full-mesh exchange of one chunk: `mine` ([rows, gains_len]) goes to every peer and peer
        p's slice arrives in parts[p]; one batch (one RCCL group: all sends and receives progress
        together, which is what lets the seven links work at the same time).  Returns the work handles."""
        if own_copy:
            parts[self.rank].copy_(mine)
        glob = (lambda r: dist.get_global_rank(self.group, r)) if self.group is not None else (lambda r: r)
        ops = []
        for d in range(1, self.world):  # rank r sends to r + d while it receives from r - d
            to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
            ops.append(dist.P2POp(dist.isend, mine, glob(to), self.group))
            ops.append(dist.P2POp(dist.irecv, parts[frm], glob(frm), self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def wait(self, i):
        """Block (stream-wise on GPU) until the gathers of sweep i have landed."""
        s = self.slot(i)
        if self.on_gpu:
            if self._done[s] is not None:
                torch.cuda.current_stream(self.device).wait_event(self._done[s])
        else:
            for w in self._work[s]:
                w.wait()
            self._work[s] = []

    def finish(self):
        """Block (stream-wise on GPU) until every outstanding gather has landed."""
        for s in range(self.depth):
            self.wait(s)
