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
    (branch_id, seq_id) pair reinterpreted as one f64, then lnl, pendant, distal; one extra row
    carries the rank's row count.  `depth` slots are used round-robin, a slot is reused only after
    its previous gather has completed.

    post() is collective: every rank calls it once per chunk, in the same order."""

    def __init__(self, dist, max_rows, device, dst=0, depth=2, host_copy=False):
        """host_copy: rank dst also copies every gathered slot into pinned host memory when it
        retires it (the results' way to the host pipeline; puts the D2H inside a timed loop)"""
        import torch
        self.dist, self.dst, self.depth = dist, dst, depth
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.max_rows = int(max_rows)
        self.cdev = device if dist.get_backend() == "nccl" else torch.device("cpu")
        mk = lambda: torch.empty((self.max_rows + 1, 4), dtype=torch.float64, device=self.cdev)
        self.send = [mk() for _ in range(depth)]
        self.recv = [[mk() for _ in range(self.world)] for _ in range(depth)] if self.rank == dst else None
        self.work = [None] * depth
        self.rows = [0] * depth
        self.nmax = torch.zeros(1, dtype=torch.int64, device=self.cdev)
        self.step = 0
        self.collected = []   # dst only: list of (step, [per-rank (n, 4) arrays]) when keep=True
        self.host = None
        if host_copy and self.rank == dst and self.cdev.type == "cuda":
            self.host = [torch.empty((self.world, self.max_rows + 1, 4), dtype=torch.float64).pin_memory()
                         for _ in range(depth)]

    def post(self, pairs_i32, results_f64, n, keep=False):
        """pairs_i32: (cap, 2) int32 tensor, results_f64: (cap, 3) float64 tensor, n valid rows."""
        import torch
        slot = self.step % self.depth
        self._retire(slot, keep)
        self.nmax[0] = n
        self.dist.all_reduce(self.nmax, op=self.dist.ReduceOp.MAX)   # common row count of this gather
        m = int(self.nmax.item())
        if m > self.max_rows:
            raise RuntimeError("AsyncResultGather: %d rows exceed max_rows %d" % (m, self.max_rows))
        buf = self.send[slot]
        if n:
            buf[:n, 0] = pairs_i32[:n].contiguous().view(torch.int64).view(torch.float64).reshape(-1).to(self.cdev)
            buf[:n, 1:4] = results_f64[:n].to(self.cdev)
        buf[m, 0] = float(n)
        if self.cdev.type == "cuda":
            # the caller's buffers are rewritten by the next chunk's kernels on the library's own HIP
            # stream, which knows nothing of torch's: the staging copy above has to be complete first
            torch.cuda.current_stream().synchronize()
        out = [r[:m + 1] for r in self.recv[slot]] if self.rank == self.dst else None
        self.work[slot] = self.dist.gather(buf[:m + 1], out, dst=self.dst, async_op=True)
        self.rows[slot] = m
        self.step += 1

    def _retire(self, slot, keep):
        w = self.work[slot]
        if w is None:
            return
        w.wait()
        self.work[slot] = None
        if self.host is not None:
            m = self.rows[slot]
            for rk, r in enumerate(self.recv[slot]):
                self.host[slot][rk, :m + 1].copy_(r[:m + 1], non_blocking=True)
        if keep and self.rank == self.dst:
            m = self.rows[slot]
            parts = []
            for r in self.recv[slot]:
                k = int(r[m, 0].item())
                parts.append(r[:k].cpu().numpy().copy())
            self.collected.append(parts)

    def finish(self, keep=False):
        """waits for every outstanding gather (oldest first)"""
        for i in range(self.depth):
            self._retire((self.step + i) % self.depth, keep)
        if self.host is not None:
            import torch
            torch.cuda.current_stream().synchronize()


def unpack_rows(rows):
    """(n, 4) f64 rows of AsyncResultGather -> (branch_id, seq_id, lnl, pendant, distal) arrays"""
    ids = np.ascontiguousarray(rows[:, 0]).view(np.int64)
    branch = (ids & 0xffffffff).astype(np.uint32)
    seq = (ids >> 32).astype(np.uint32)
    return branch, seq, rows[:, 1], rows[:, 2], rows[:, 3]
