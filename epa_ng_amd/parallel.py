"""Query sharding across ranks (one process per GPU) and the result gather.

The path shards by query with no data-path collective, exactly like the reference's MPI mode:
rank r of R takes the contiguous slice of `local_seq_package` (src/net/epa_mpi_util.cpp:10-30).
The only exchange is the gather of the per-pair results to rank 0 (the reference gathers jplace
byte ranges instead, src/io/jplace_writer.hpp:117-129).  torch.distributed is plumbing: backend
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
import numpy as np


def local_seq_package(num_sequences, rank, world):
    """-> (offset, count) of rank's slice; part size ceil(n / world), trailing ranks may be empty."""
    part = -(-num_sequences // world)
    offset = min(part * rank, num_sequences)
    return offset, max(0, min(part, num_sequences - offset))


def gather_results(pairs, results, seq_offset, dist=None, dst=0):
    """pairs: structured (branch_id, seq_id) local to the rank, results: (lnl, pendant, distal).
    Returns on dst the concatenation over ranks with GLOBAL sequence ids, elsewhere None."""
    import torch
    n = len(pairs)
    rec = np.empty((n, 5), np.float64)
    rec[:, 0] = pairs["branch_id"]
    rec[:, 1] = pairs["seq_id"].astype(np.float64) + seq_offset
    rec[:, 2] = results["lnl"]
    rec[:, 3] = results["pendant_length"]
    rec[:, 4] = results["distal_length"]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=dev))
    cap = int(max(int(c.item()) for c in counts))
    buf = torch.zeros((cap, 5), dtype=torch.float64, device=dev)
    if n:
        buf[:n] = torch.from_numpy(rec).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return np.concatenate([o[:int(c.item())].cpu().numpy() for o, c in zip(out, counts)], axis=0)


class AsyncResultGather:
    """The path's only exchange, overlapped with compute: after every chunk each rank posts its
    (pair, result) rows to rank 0 with an asynchronous gather (RCCL runs it on its own stream over
    xGMI while the next chunk's kernels execute).  Rows are packed as 4 x f64: the 8-byte
    (branch_id, seq_id) pair reinterpreted as one f64, then lnl, pendant, distal.

    No per-chunk host synchronisation and no size negotiation between the ranks: every gather has
    the SAME fixed size on every rank (`rows_cap` rows + one sentinel row whose first field is the
    number of valid rows), so post() only enqueues work.  A rank whose chunk produced more than
    rows_cap rows sends the first rows_cap and carries the rest over into its next gather (a rare
    path: rows_cap is sized from the expected candidates per read); finish() drains what is still
    carried with extra rounds, agreed on by the only all_reduce of the object's life.
    `depth` send / receive slots are used round-robin, a slot is reused only after its previous
    gather has completed.  On rank dst, with host_copy, a retired slot's VALID rows go to pinned
    host memory on a side stream (the compute stream never waits for PCIe).

    Stream contract (GPU): the producer of `pairs` / `results` runs on torch's CURRENT stream (the
    evaluator is pointed at it with Evaluator.set_stream), so the staging copies enqueued here are
    ordered after the chunk's kernels and before the next chunk's -- no synchronize().

    post() is collective: every rank calls it once per chunk, in the same order."""

    def __init__(self, dist, rows_cap, device, dst=0, depth=2, host_copy=False):
        import torch
        self.torch = torch
        self.dist, self.dst, self.depth = dist, dst, depth
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.rows_cap = int(rows_cap)
        self.cdev = device if dist.get_backend() == "nccl" else torch.device("cpu")
        self.gpu = self.cdev.type == "cuda"
        mk = lambda: torch.zeros((self.rows_cap + 1, 4), dtype=torch.float64, device=self.cdev)
        self.send = [mk() for _ in range(depth)]
        self.recv = [[mk() for _ in range(self.world)] for _ in range(depth)] if self.rank == dst else None
        self.work = [None] * depth
        self.step = 0
        self.carry = None          # (k, 4) rows this rank could not send yet
        self.carried_rows = 0      # diagnostics: rows that ever took the carry path
        self.collected = []        # dst, keep=True: per gather a list of per-rank (n, 4) arrays
        self.counts = []           # dst, host_copy or keep: per gather the per-rank valid row counts
        self.host = self.host_cnt = self.side = None
        if self.rank == dst and self.gpu:
            self.side = torch.cuda.Stream(device=self.cdev)
            self.host_cnt = torch.zeros((depth, self.world, 4), dtype=torch.float64).pin_memory()
            if host_copy:
                self.host = [torch.empty((self.world, self.rows_cap, 4), dtype=torch.float64).pin_memory()
                             for _ in range(depth)]

    def _pack(self, pairs_i32, results_f64, n, out):
        t = self.torch
        out[:n, 0] = pairs_i32[:n].contiguous().view(t.int64).view(t.float64).reshape(-1).to(self.cdev)
        out[:n, 1:4] = results_f64[:n].to(self.cdev)

    def post(self, pairs_i32, results_f64, n, keep=False):
        """pairs_i32: (cap, 2) int32 tensor, results_f64: (cap, 3) float64 tensor, n valid rows."""
        t = self.torch
        slot = self.step % self.depth
        self._retire(slot, keep)
        buf, cap = self.send[slot], self.rows_cap
        if self.carry is None and n <= cap:            # the usual path: straight into the send slot
            if n:
                self._pack(pairs_i32, results_f64, n, buf)
            m = n
        else:
            rows = t.empty((n, 4), dtype=t.float64, device=self.cdev)
            if n:
                self._pack(pairs_i32, results_f64, n, rows)
            if self.carry is not None:
                rows = t.cat([self.carry, rows], 0)
            m = min(cap, rows.shape[0])
            buf[:m] = rows[:m]
            self.carry = rows[m:].clone() if rows.shape[0] > m else None
            self.carried_rows += rows.shape[0] - m
        buf[cap, 0] = float(m)
        out = self.recv[slot] if self.rank == self.dst else None
        self.work[slot] = self.dist.gather(buf, out, dst=self.dst, async_op=True)
        self.step += 1

    def _retire(self, slot, keep):
        w = self.work[slot]
        if w is None:
            return
        self.work[slot] = None
        t = self.torch
        if not self.gpu:
            w.wait()
            if self.rank == self.dst and keep:
                cnt = [int(r[self.rows_cap, 0].item()) for r in self.recv[slot]]
                self.counts.append(cnt)
                self.collected.append([r[:k].numpy().copy() for r, k in zip(self.recv[slot], cnt)])
            return
        if self.rank != self.dst:
            w.wait()                                   # stream-level: the slot's next write is ordered behind the gather
            return
        with t.cuda.stream(self.side):
            w.wait()                                   # the SIDE stream waits for the gather, not the compute stream
            for rk, r in enumerate(self.recv[slot]):
                self.host_cnt[slot, rk].copy_(r[self.rows_cap], non_blocking=True)
            # wait for exactly these copies (an event, not the stream: nothing queued on the side stream later --
            # the previous slot's row copies -- is waited for); the gather was posted `depth` chunks ago
            ev = t.cuda.Event()
            ev.record(self.side)
            ev.synchronize()
            cnt = [int(v) for v in self.host_cnt[slot, :, 0].tolist()]
            if self.host is not None:
                for rk, (r, k) in enumerate(zip(self.recv[slot], cnt)):
                    if k:
                        self.host[slot][rk, :k].copy_(r[:k], non_blocking=True)
            if keep:
                self.collected.append([r[:k].cpu().numpy().copy() for r, k in zip(self.recv[slot], cnt)])
        self.counts.append(cnt)
        t.cuda.current_stream().wait_stream(self.side)  # recv slot is free for the next gather into it

    def finish(self, keep=False):
        """drains the carried rows (extra rounds agreed on by one all_reduce) and waits for every
        outstanding gather (oldest first)"""
        t = self.torch
        pend = t.tensor([0 if self.carry is None else int(self.carry.shape[0])], dtype=t.int64, device=self.cdev)
        self.dist.all_reduce(pend, op=self.dist.ReduceOp.MAX)
        extra = -(-int(pend.item()) // self.rows_cap)
        empty_p = t.zeros((0, 2), dtype=t.int32, device=self.cdev)
        empty_r = t.zeros((0, 3), dtype=t.float64, device=self.cdev)
        for _ in range(extra):
            self.post(empty_p, empty_r, 0, keep)
        for i in range(self.depth):
            self._retire((self.step + i) % self.depth, keep)
        if self.gpu:
            if self.side is not None:
                self.side.synchronize()
            t.cuda.current_stream().synchronize()


def unpack_rows(rows):
    """(n, 4) f64 rows of AsyncResultGather -> (branch_id, seq_id, lnl, pendant, distal) arrays"""
    ids = np.ascontiguousarray(rows[:, 0]).view(np.int64)
    branch = (ids & 0xffffffff).astype(np.uint32)
    seq = (ids >> 32).astype(np.uint32)
    return branch, seq, rows[:, 1], rows[:, 2], rows[:, 3]
