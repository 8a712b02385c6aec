#!/usr/bin/env python3
"""Benchmark of the placement hot path on MI355X (contract: see the task's bench.py section).

Metric (BASELINE.json): query placements/sec, 512-tip GTR+G4 DNA reference, preplace + thorough.
Workload = cfg2 of SURVEY.md section 8d: 512-tip random-join tree (seed 1), 1500-column MSA
simulated on that tree (seed 2), 150 bp reads with 3 % substitutions (seed 3 + 1000*rank),
dynamic heuristic 0.99999.  A "step" is one chunk of --chunk reads through
    preplace (Q x B lookup sums) -> candidate selection -> thorough NR placement
with the encoded reads already resident in HBM (default 50 000 reads per step: the GPU wants
larger chunks than the reference's CPU default of 5000, it is the same --chunk-size knob); every rank works on its own reads (weak
scaling, no data-path collective) and rank 0 gathers the per-pair results over RCCL.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --steps 5 --warmup 1
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == matrix peak (spec); the kernel is fp64 VALU
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--chunk", type=int, default=100000,
                   help="reads per step (EPA-ng --chunk-size; default = the whole cfg2 query set)")
    p.add_argument("--tips", type=int, default=512)
    p.add_argument("--width", type=int, default=1500)
    p.add_argument("--read-len", type=int, default=150)
    p.add_argument("--cpu-sample", type=int, default=20000, help="reads timed on the CPU baseline")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--workload", choices=["dna", "aa"], default="dna",
                   help="dna = cfg2 (the metric's config); aa = cfg3 shape (use --tips 2000 --width 500 "
                        "--read-len 100), a parity/measurement case, not the headline")
    return p.parse_args()


def main():
    a = parse()
    import torch
    import epa_ng_amd as epa
    from epa_ng_amd import hostlib, parallel, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d" % (a.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only compute path")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("EPA_BENCH_BACKEND", "nccl")   # "gloo" only for 1-GPU plumbing tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)
    cdev = dev if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")  # collective buffers

    # ---------------- workload (identical reference on every rank, rank-private reads)
    n_chunks = a.steps + a.warmup
    states = 4 if a.workload == "dna" else 20
    if a.workload == "dna":
        subst, freqs, alpha, seeds = synth.CFG2_SUBST, synth.CFG2_FREQS, synth.CFG2_ALPHA, (1, 2, 3)
    else:
        subst, freqs = synth.aa_model()
        alpha, seeds = 0.563473, (11, 12, 13)
    root = synth.random_tree(a.tips, seeds[0])
    rates = synth.gamma_rates(alpha)
    labels, seqs = synth.simulate_msa(root, a.width, subst, freqs, rates, seeds[1])
    newick = synth.newick(root)
    reads, _ = synth.make_reads(seqs, n_chunks * a.chunk, a.read_len, 0.03, seeds[2] + 1000 * rank,
                                states=states)
    ref = hostlib.Reference(newick, labels, seqs, states=states, subst=subst, freqs=freqs, rates=rates)
    ev = ref.evaluator(device=local)
    t0 = time.time()
    ev.build_lookup()
    torch.cuda.synchronize()
    lookup_ms = ev.kernel_ms("lookup")
    B, W, Q = ref.B, ref.W, a.chunk

    chunks = []
    for c in range(n_chunks):
        # compact wire format: one row per read holding only its window (152 B instead of 1500 B)
        codes, wb, ws = epa.encode_queries(states, reads[c * Q:(c + 1) * Q], compact=True)
        chunks.append((torch.from_numpy(codes).to(dev), torch.from_numpy(wb.view(np.int32)).to(dev),
                       torch.from_numpy(ws.view(np.int32)).to(dev), codes, wb, ws))
    cap = Q * 64   # pair capacity per step (dynamic heuristic selects a handful per read)
    d_pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    d_res = torch.zeros((cap, 3), dtype=torch.float64, device=dev)
    exch = parallel.AsyncResultGather(dist, min(cap, 8 * Q), dev) if world > 1 else None

    th_ms, th_pairs, th_rounds, th_evals, pre_ms, sel_ms = [], [], [], [], [], []

    def step(i, record):
        dc, dwb, dws = chunks[i][0], chunks[i][1], chunks[i][2]
        # one fused call = the reference's chunk body: place() -> apply_heuristic() -> place_thorough()
        n = ev.place_chunk(dc, dwb, dws, Q=Q, threshold=0.99999, max_span=a.read_len, max_pairs=cap,
                           pairs_out=d_pairs, results_out=d_res)
        if world > 1:
            # the path's only exchange: every rank's candidate placements -> rank 0 (RCCL over
            # xGMI; the reference gathers jplace byte ranges, src/io/jplace_writer.hpp:117-129).
            # Posted asynchronously: it overlaps the next chunk's kernels (parallel.py).
            exch.post(d_pairs, d_res, n)
        if record:
            th_ms.append(ev.kernel_ms("thorough")); pre_ms.append(ev.kernel_ms("preplace"))
            sel_ms.append(ev.kernel_ms("select"))
            th_pairs.append(n); th_rounds.append(ev.last_stats["rounds"])
            th_evals.append(ev.last_stats["newton_evals"])
        return n

    for i in range(a.warmup):
        step(i, False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.warmup, n_chunks):
        step(i, True)
    if world > 1:
        exch.finish()   # the last chunks' gathers are part of the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_reads = world * a.steps * Q
    value = total_reads / elapsed
    # ---------------- roofline of the dominant kernel (thorough NR), per launch
    pairs = float(np.mean(th_pairs))
    R = float(np.sum(th_rounds)) / float(np.sum(th_pairs))
    kbar = float(np.sum(th_evals)) / max(1.0, 2.0 * float(np.sum(th_rounds)))
    nq = a.read_len
    if states == 4:
        flops_pair = nq * (884.0 + R * (1258.0 + 240.0 * kbar))
        # the initial inner CLV (2 x U products, the fold, the U^-1 product: 432 of the 884 flop
        # per site) is served from the per-branch precompute of k_build_lookup, not recomputed
        flops_exec = nq * (452.0 + R * (1258.0 + 240.0 * kbar))
    else:  # same accounting with s = 20: P = c(4s^2+s), E = c(2s^2+2s), D = 6cs
        P_, E_, D_ = 4 * (4 * 400 + 20), 4 * (2 * 400 + 40), 6 * 4 * 20
        flops_pair = nq * ((2 * P_ + E_) + R * (4 * P_ + E_ + 2 * kbar * D_))
        # initial inner CLV from the per-branch precompute: only the fold with the query and the
        # lnL contraction (3cs flop per site) are executed of the 2P + E
        flops_exec = nq * (3 * 4 * 20 + R * (4 * P_ + E_ + 2 * kbar * D_))
    cs = 4 * states
    bytes_pair = 2 * nq * cs * 8 + 2 * nq * 4 + nq + 24
    t_th = float(np.mean(th_ms)) * 1e-3
    ach_tflops = pairs * flops_pair / t_th / 1e12
    traffic = None   # HBM bytes per launch from the committed PMC passes, same workload only
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))["k_thorough_dna"]
        if states == 4 and tj["reads_per_step"] == Q and abs(tj["pairs_per_launch"] - pairs) < 0.02 * pairs:
            traffic = (tj["fetch_kb"] + tj["write_kb"]) * 1024.0
    except (OSError, KeyError, ValueError):
        pass
    roof = {"bound": "mfma", "kernel": "k_thorough_dna" if states == 4 else "k_thorough_aa", "achieved": round(ach_tflops, 3),
            "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach_tflops / FP64_PEAK_TFLOPS, 4),
            "traffic": traffic,
            "note": "fp64 VALU kernel priced against the fp64 vector=matrix peak (78.6 TF spec)",
            "pairs_per_launch": pairs, "rounds_per_pair": round(R, 3), "newton_iters_per_solve": round(kbar, 3),
            "flops_per_pair": round(flops_pair), "flops_executed_per_pair": round(flops_exec),
            "frac_executed": round(pairs * flops_exec / t_th / 1e12 / FP64_PEAK_TFLOPS, 4), "ms_per_launch": round(t_th * 1e3, 4),
            "hbm_algorithmic_GBs": round(pairs * bytes_pair / t_th / 1e9, 1),
            "hbm_frac": round(pairs * bytes_pair / t_th / 1e9 / HBM_PEAK_GBS, 4)}

    # ---------------- CPU baseline: the oracle's OpenMP restatement on a bounded sample
    cpu = None
    parity = None
    if not a.no_cpu_baseline and world == 1:   # rank 0 at N=1 only (bench contract)
        # the oracle's OpenMP threads = the CPUs this process may really use (a container can see
        # 256 CPUs and be limited to 16 by its cgroup quota: oversubscribing would slow the baseline)
        eff = hostlib.configure_threads()
        from oracle_lib import Oracle, lib as orc_lib
        ns = min(a.cpu_sample, Q)
        sample = reads[a.warmup * Q: a.warmup * Q + ns]
        o = Oracle(newick, labels, seqs, states, subst, freqs, rates)
        o.preplace(sample[:8])                      # builds the per-branch lookups (one-off)
        codes, wb, ws = epa.encode_queries(states, sample)
        lnl_gpu = ev.preplace(codes, wb, ws)
        prs = ev.select(lnl_gpu, ns, 0.99999)
        res_gpu = ev.thorough(prs, codes, wb, ws)
        c0 = time.perf_counter()
        lnl_cpu = o.preplace(sample)
        tl, tp, td = o.thorough(prs["branch_id"], prs["seq_id"], sample)
        cpu_t = time.perf_counter() - c0
        cores = min(eff, orc_lib().orc_max_threads())
        cpu = {"value": round(ns / cpu_t, 2), "unit": "placements/s", "cores": cores, "kind": "port",
               "sample": "%d reads of step 0 through the oracle's OpenMP preplace (B=%d) + thorough "
                         "(%d pairs), lookups prebuilt" % (ns, B, len(prs))}
        parity = {"preplace_max_abs_dlnl": float(np.max(np.abs(lnl_gpu - lnl_cpu))),
                  "thorough_max_abs_dlnl": float(np.max(np.abs(res_gpu["lnl"] - tl))),
                  "pairs_checked": int(len(prs))}

    metric = ("query placements/sec (whole node), 512-tip GTR+G4 DNA ref, preplace+thorough" if states == 4
              else "query placements/sec (whole node), AA PROTGTR+G4 ref (cfg3 shape), preplace+thorough")
    out = {"metric": metric,
           "value": round(value, 2), "unit": "placements/s", "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": ("cfg2: %d-tip DNA GTR+G4 ref, W=%d, %d bp reads, dyn-heur 0.99999, "
                                   "preplace+thorough" if states == 4 else
                                   "cfg3-shape: %d-tip AA PROTGTR+G4 ref, W=%d, %d aa queries, dyn-heur "
                                   "0.99999, preplace+thorough") % (a.tips, a.width, a.read_len),
                      "reads_per_step_per_gpu": Q, "branches": B, "parallelism": "query-shard x%d" % world,
                      "lookup_build_ms_once": round(lookup_ms, 3),
                      "kernel_ms_per_step": {"preplace": round(float(np.mean(pre_ms)), 3),
                                             "select": round(float(np.mean(sel_ms)), 3),
                                             "thorough": round(float(np.mean(th_ms)), 3)}},
           "roofline": roof, "cpu_baseline": cpu, "parity": parity}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
