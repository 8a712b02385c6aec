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
