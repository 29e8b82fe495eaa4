"""Multi-GPU sharding of self-play: one process per GPU, games sharded by index, no exchange while playing;
the only collective is an all-gather (RCCL over xGMI; gloo on CPU in the tests) that pools the fixed-size
training tuples -- the counterpart of the reference's mpsc collection, agent/async_agent/mod.rs:129-134.

TuplePool is the pipelined form of that exchange.  The engine writes finished games' tuples into one of two torch-owned
device buffers; `exchange()` retires the filled one, hands the engine the other and pools the retired records on a SIDE
stream while the engine's own stream keeps playing:

    engine stream :  rounds k ........ | swap | rounds k+1 ........................ | swap | ...
    side stream   :                    wait(ev) all-gather(counts) -> host ; all-gather(max(count) records) ; zero count

Only `max(counts)` records travel (rounded up to 64), nothing at all when no game finished on any rank, and the host only
ever waits for work that is BEHIND rounds already enqueued.  (The engine's stream waits for the pooling of the buffer it is
about to write again -- two intervals back -- so it idles only if an all-gather takes longer than an interval of rounds.)
A rank that finishes more records in one interval than a buffer holds loses the excess (the engine stops writing at `cap`):
that is counted in `dropped` and raised by `check()`, never silent -- size `cap` for games x stones per interval.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_games(total_games, world_size, rank):
    """games per rank (weak scaling uses the same count on every rank) and this rank's first C6RNG game uid"""
    per = total_games // world_size
    rem = total_games % world_size
    mine = per + (1 if rank < rem else 0)
    first = rank * per + min(rank, rem)
    return mine, first


def uid_plan(games_per_rank, world_size, rank):
    """C6RNG game uids: slot i of rank r starts at r*G + i and a recycled slot advances by world*G, so that the
    union over ranks is the same set of games whatever the number of GPUs"""
    return rank * games_per_rank, world_size * games_per_rank


def gather_tuples(buf, count, world_size, group=None):
    """buf: uint8 tensor [cap, stride] (device or CPU), count: int32 tensor [1].
    Returns (pooled [world, cap, stride], counts [world]) on every rank.  Whole-buffer form (tests, small buffers);
    TuplePool below is the production path."""
    if world_size == 1:
        return buf.unsqueeze(0), count.reshape(1)
    pooled = torch.empty((world_size,) + tuple(buf.shape), dtype=buf.dtype, device=buf.device)
    counts = torch.empty(world_size, dtype=count.dtype, device=count.device)
    dist.all_gather_into_tensor(pooled.view(-1), buf.reshape(-1).contiguous(), group=group)
    dist.all_gather_into_tensor(counts, count.reshape(1), group=group)
    return pooled, counts


def compact(pooled, counts):
    """concatenate the valid records of every rank (host side)"""
    pooled = pooled.cpu().numpy()
    counts = counts.cpu().numpy()
    cap = pooled.shape[1]
    return np.concatenate([pooled[r, : min(int(counts[r]), cap)] for r in range(pooled.shape[0])], axis=0)


class TuplePool:
    """Double-buffered, side-stream pooling of an engine's training tuples over all ranks.

    backend "nccl": device tensors, RCCL all-gather on a side stream.  Any other backend (gloo: CPU tests, or several ranks
    sharing one GPU): the same protocol through host copies of the retired buffer.
    `consumer(pooled, counts)` (optional) is called on the side stream with the pooled device tensor [world, m, stride] and
    the host list of valid records per rank -- e.g. ReplayBuffer.push_device per rank segment."""

    QUANTUM = 64

    def __init__(self, eng, cap, world, backend="nccl", group=None, consumer=None, keep=True, collective=None):
        from . import lib
        self.eng, self.cap, self.world, self.backend, self.group, self.consumer = eng, int(cap), world, backend, group, consumer
        # collective: issue the all-gathers.  Default: whenever there is more than one rank.  True with world == 1 (and an initialised
        # one-rank nccl group) sends a single GPU through the very same RCCL calls, streams and events as a rank of eight -- what a
        # one-GPU box can execute of the N > 1 branch (tests/test_gpu_configs.py::test_config4_per_gpu_workload)
        self.collective = (world > 1) if collective is None else bool(collective)
        if self.collective and not dist.is_initialized():
            raise RuntimeError("TuplePool(collective=True) needs an initialised torch.distributed process group")
        self.stride = int(lib().c6_tuple_stride(eng.n))
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.bufs = [torch.zeros((self.cap, self.stride), dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.counts = [torch.zeros(1, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self.cur = 0
        self.keep = keep                      # keep pooled records for collect() (tests); the bench only counts them
        self.pending = []                     # host copies of pooled records not yet handed out
        self.total = torch.zeros(1, dtype=torch.int64, device=self.dev)   # records pooled so far (all ranks), device side
        self.exchanges = 0
        self.payload_gathers = 0
        self.dropped = 0                      # records beyond `cap` in some interval (all ranks): lost by the engine, counted here
        self.side = torch.cuda.Stream(device=self.dev)
        self.eng_stream = torch.cuda.ExternalStream(eng.stream_ptr(), device=self.dev)
        self.free_ev = [None, None]           # side-stream event after which buffer i may be written again
        self._retired = None
        self._prewarm()
        eng.swap_tuple_buffer(self.bufs[0].data_ptr(), self.cap, self.counts[0].data_ptr())

    def _prewarm(self):
        """Every torch op / collective of an exchange once, on dummies, BEFORE the engine plays: on a fresh box the first use of a
        torch kernel pages its code object in from disk (seconds) and the first RCCL collective builds its channels -- neither
        belongs into somebody's timed region."""
        with torch.cuda.stream(self.side):
            cnt = torch.zeros(1, dtype=torch.int32, device=self.dev)
            if self.collective and self.backend == "nccl":
                d = torch.empty(self.world, dtype=torch.int32, device=self.dev)
                dist.all_gather_into_tensor(d, cnt, group=self.group)
                p = torch.empty((self.world, self.QUANTUM, self.stride), dtype=torch.uint8, device=self.dev)
                dist.all_gather_into_tensor(p.view(-1), self.bufs[1][:self.QUANTUM].reshape(-1), group=self.group)
            else:
                d = cnt.clone()
            t = torch.zeros(1, dtype=torch.int64, device=self.dev)
            t += d.clamp(max=self.cap).sum()
            d.clamp(max=self.cap).cpu().tolist()
            self.bufs[1][:self.QUANTUM].unsqueeze(0).clone().cpu()
            cnt.zero_()
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.eng_stream.wait_event(ev)
            ev2 = torch.cuda.Event()
            ev2.record(self.eng_stream)
            self.side.wait_event(ev2)
        if self.collective and self.backend != "nccl":
            c = torch.empty(self.world, dtype=torch.int32)
            dist.all_gather_into_tensor(c, torch.zeros(1, dtype=torch.int32), group=self.group)
        torch.cuda.synchronize()

    # -- the exchange ------------------------------------------------------------------------------------------
    def exchange(self):
        """One call per interval of rounds.  Pools the buffer retired by the PREVIOUS call (its rounds are an interval
        behind what has been enqueued since, so the host's wait for its counts overlaps enqueued work), then retires
        the buffer the engine has been writing and switches the engine to the other one."""
        if self._retired is not None:
            self._pool_retired()
        old, new = self.cur, self.cur ^ 1
        if self.free_ev[new] is not None:
            self.eng_stream.wait_event(self.free_ev[new])     # the engine may not write `new` before its last pooling is done
        self.eng.swap_tuple_buffer(self.bufs[new].data_ptr(), self.cap, self.counts[new].data_ptr())
        self.cur = new
        ev = torch.cuda.Event()
        ev.record(self.eng_stream)
        self.exchanges += 1
        self._retired = (old, ev)

    def _pool_retired(self):
        old, ev = self._retired
        self._retired = None
        buf, cnt = self.bufs[old], self.counts[old]
        if self.backend != "nccl":
            ev.synchronize()
            n_raw = int(cnt.item())
            if self.collective:
                counts = torch.empty(self.world, dtype=torch.int32)
                dist.all_gather_into_tensor(counts, torch.tensor([n_raw], dtype=torch.int32), group=self.group)
                counts = counts.tolist()
            else:
                counts = [n_raw]
            self.dropped += sum(max(0, c - self.cap) for c in counts)
            counts = [min(c, self.cap) for c in counts]
            m = self._padded(max(counts))
            if m:
                mine = buf[:m].cpu()
                if self.collective:
                    pooled = torch.empty((self.world, m, self.stride), dtype=torch.uint8)
                    dist.all_gather_into_tensor(pooled.view(-1), mine.view(-1), group=self.group)
                else:
                    pooled = mine.unsqueeze(0)
                self.payload_gathers += 1
                self._deliver(pooled, counts, host=True)
            cnt.zero_()
            torch.cuda.current_stream().synchronize()
            return
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            if self.collective:
                dcounts = torch.empty(self.world, dtype=torch.int32, device=self.dev)
                dist.all_gather_into_tensor(dcounts, cnt, group=self.group)
            else:
                dcounts = cnt.clone()
            self.total += dcounts.clamp(max=self.cap).sum()
            raw = dcounts.cpu().tolist()                             # the only host wait: work behind already enqueued rounds
            self.dropped += sum(max(0, c - self.cap) for c in raw)
            counts = [min(c, self.cap) for c in raw]
            m = self._padded(max(counts))
            if m:                                                    # every rank takes the same branch: counts are pooled
                if self.collective:
                    pooled = torch.empty((self.world, m, self.stride), dtype=torch.uint8, device=self.dev)
                    dist.all_gather_into_tensor(pooled.view(-1), buf[:m].reshape(-1), group=self.group)
                else:
                    pooled = buf[:m].unsqueeze(0).clone()
                self.payload_gathers += 1
                self._deliver(pooled, counts, host=False)
            cnt.zero_()
            fe = torch.cuda.Event()
            fe.record(self.side)
            self.free_ev[old] = fe

    def _padded(self, m):
        return 0 if m <= 0 else min(self.cap, (m + self.QUANTUM - 1) // self.QUANTUM * self.QUANTUM)

    def _deliver(self, pooled, counts, host):
        if self.consumer is not None:
            self.consumer(pooled, counts)
        if host:
            self.total += sum(counts)
        if self.keep:
            self.pending.append((pooled if host else pooled.cpu(), counts))

    # -- host-side access (tests / small runs) -------------------------------------------------------------------
    def collect(self):
        """records pooled so far (all ranks, valid ones only) as one uint8 array, or None"""
        if not self.pending:
            return None
        parts = []
        for pooled, counts in self.pending:
            p = pooled.numpy()
            parts.extend(p[r, :counts[r]] for r in range(p.shape[0]))
        self.pending = []
        return np.concatenate(parts, axis=0) if parts else None

    def flush(self):
        """pool whatever the engine wrote since the last exchange() and return everything not yet collected"""
        self.exchange()
        self.exchange()
        torch.cuda.synchronize()
        return self.collect()

    def check(self):
        """raise if any rank has lost records to a full buffer"""
        if self.dropped:
            raise RuntimeError(f"TuplePool: {self.dropped} training tuples were dropped (buffer of {self.cap} records too small for an interval)")

    def pooled_total(self):
        torch.cuda.synchronize()
        return int(self.total.item())
