#!/usr/bin/env python3
"""Benchmark of the placement hot path on MI355X (contract: see the task's bench.py section).

Metric (BASELINE.json): query placements/sec, 512-tip GTR+G4 DNA reference, preplace + thorough.
Workload = cfg2 of SURVEY.md section 8d: 512-tip random-join tree (seed 1), 1500-column MSA
simulated on that tree (seed 2), 150 bp reads with 3 % substitutions (seed 3; chunk g of the read
stream uses seed 3 + 1000003 g), dynamic heuristic 0.99999.  A "step" is one chunk of --chunk
reads through the reference's chunk body
    place() (Q x B lookup sums) -> apply_heuristic() -> place_thorough() (NR placement).

Two timed loops over the SAME chunks, both in the JSON line:
  value            inputs resident in HBM when the clock starts, results left in HBM (the bench
                   contract's definition of `value`): epa_dev_place_chunk on device buffers;
  pcie_inclusive   SURVEY 8d's definition (H2D of the queries + D2H of the results inside the
                   step): the double-buffered epa_dev_chunk_stage / _launch / _finish pipeline,
                   4-bit wire format up, pairs + results down, copies on the copy stream.
N > 1: one process per GPU, every rank --chunk reads per step in BOTH scaling modes (a step of the
job is --chunk x N reads, sharded with the reference's local_seq_package formula: contiguous
slices).  --scaling weak (default, the bench contract's K steps): per-GPU work fixed, total work
grows with N.  --scaling strong --reads R: the job is R reads whatever N is, steps =
ceil(R / (chunk x N)) -- BASELINE configs[3] (cfg4) is `--gpus 8 --scaling strong --reads 10000000`.
Every default run ALSO times that fixed 10^7-read job on its N GPUs (`strong_cfg4` in the line), so
that the driver's N = 1, 2, 4, 8 lines carry a strong-scaling curve next to the weak one.
The only exchange is the RCCL gather of the per-pair results to rank 0, overlapped, no per-chunk
host synchronisation: the PRODUCT's gather (libepa_dev.so epa_comm_*, csrc/comm.hip); torch.distributed
launches, barriers and broadcasts the 128-byte id (EPA_BENCH_GATHER=torch: the torch harness of the same
protocol, epa_ng_amd/parallel.py).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --steps 5 --warmup 1
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_PEAK_TFLOPS = 78.6       # AMD MI355X spec sheet: fp64 vector = fp64 matrix = 78.6 TFLOP/s
FP64_MEASURED_CEILING = 63.3  # dependency-free v_fma_f64 stream on this chip (profiles/r1_mfma_overlap.txt)
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md (spec; 6290 GB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=5)   # the first launches run ~20 % slower (clocks, TLB): profiles/r3 timeline
    p.add_argument("--chunk", type=int, default=100000,
                   help="reads per step (EPA-ng --chunk-size; default = the whole cfg2 query set)")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    p.add_argument("--reads", type=int, default=0,
                   help="strong scaling: total reads of the job (steps = ceil(reads / (chunk x N)))")
    p.add_argument("--strong-reads", type=int, default=10000000,
                   help="size of the fixed job timed as `strong_cfg4` (BASELINE configs[3]: 10^7 reads); 0 = skip")
    p.add_argument("--pool", type=int, default=12,
                   help="distinct synthetic chunks resident per GPU; longer runs cycle through them")
    p.add_argument("--tips", type=int, default=512)
    p.add_argument("--width", type=int, default=1500)
    p.add_argument("--read-len", type=int, default=150)
    p.add_argument("--cpu-sample", type=int, default=20000, help="reads timed on the CPU baseline")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--parity-sample", type=int, default=0,
                   help="with --no-cpu-baseline: still check this many reads of step 0 against the oracle (untimed)")
    p.add_argument("--no-extras", action="store_true",
                   help="skip the secondary measurements (chunk-5000 rate, reference-binary hook)")
    p.add_argument("--serial-schedule", action="store_true",
                   help="run every loop in the serialised two-slot order (Newton launches never overlap): the command whose "
                        "kernel trace must agree with roofline.ms_per_launch (profiles/rN_kernel_trace_stats_serialised.txt)")
    p.add_argument("--workload", choices=["dna", "aa", "cfg5"], default="dna",
                   help="dna = cfg2 (the metric's config); aa = cfg3 shape (use --tips 2000 --width 500 "
                        "--read-len 100), a parity/measurement case, not the headline")
    return p.parse_args()


def reference_binary_check(newick, labels, seqs, sample_codes, wb, ws, W, states, model, ours):
    """SURVEY 8d / BASELINE.md:33: if an `epa-ng` executable is on PATH, run it on the same files
    and diff the placements; otherwise say so.  `ours`: {read index: (best edge, its lnL)}."""
    exe = shutil.which("epa-ng")
    if not exe:
        return {"status": "reference binary unavailable (no epa-ng on PATH)"}
    from epa_ng_amd import synth
    try:
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "ref.tre"), "w") as f:
                f.write(newick + "\n")
            with open(os.path.join(d, "ref.fasta"), "w") as f:
                for l, s in zip(labels, seqs):
                    f.write(">%s\n%s\n" % (l, s))
            reads = synth.compact_to_ascii(sample_codes, wb, ws, W, states)
            with open(os.path.join(d, "q.fasta"), "w") as f:
                for i, s in enumerate(reads):
                    f.write(">q%d\n%s\n" % (i, s))
            t0 = time.perf_counter()
            r = subprocess.run([exe, "--tree", os.path.join(d, "ref.tre"), "--ref-msa", os.path.join(d, "ref.fasta"),
                                "--query", os.path.join(d, "q.fasta"), "--model", model, "--outdir", d,
                                "--redo"], capture_output=True, text=True, timeout=1800)
            secs = time.perf_counter() - t0
            if r.returncode != 0:
                return {"status": "epa-ng failed", "stderr": r.stderr[-400:]}
            jp = json.load(open(os.path.join(d, "epa_result.jplace")))
        same_edge, dl = 0, []
        for pq in jp["placements"]:
            q = int(pq["n"][0][1:])
            best = max(pq["p"], key=lambda p: p[2])
            if q in ours:
                same_edge += int(best[0] == ours[q][0])
                if best[0] == ours[q][0]:
                    dl.append(abs(best[1] - ours[q][1]))
        return {"status": "ran", "reads": len(reads), "wall_s": round(secs, 2),
                "placements_per_s_incl_setup": round(len(reads) / secs, 1),
                "best_edge_agree": same_edge, "max_abs_dlnl_on_agreeing": float(max(dl)) if dl else None}
    except Exception as e:  # noqa: BLE001  (a diagnostic hook must never take the bench line down)
        return {"status": "hook error: %r" % (e,)}


def cli_e2e_leg(newick, labels, seqs, chunks, W, model):
    """The drop-in EXECUTABLE end to end: the cfg2 reference and the given reads written
    once as aligned FASTA and as the reference's binary fasta (.bfast), `epa-ng-amd` run on each with its defaults --
    everything a user's run does per read besides the device calls is inside: reading / indexing the file, ASCII or
    4-bit -> wire codes, the chunk pipeline, LWR + filter, jplace text, the write.  reads/s = reads / loop_s (first
    read -> closed jplace; the one-off reference setup is reported beside it), stage times are busy seconds of stages
    that overlap on the host's cores."""
    from epa_ng_amd import hostlib, synth
    import hashlib
    exe = hostlib.cli_exe()
    out = {"host_cores": hostlib.configure_threads()}
    d = tempfile.mkdtemp(prefix="epa_cli_e2e_")
    try:
        with open(os.path.join(d, "ref.tre"), "w") as f:
            f.write(newick + "\n")
        with open(os.path.join(d, "ref.fasta"), "w") as f:
            for l, s_ in zip(labels, seqs):
                f.write(">%s\n%s\n" % (l, s_))
        t0 = time.perf_counter()
        n = synth.write_query_files(os.path.join(d, "q.fasta"), os.path.join(d, "q.bfast"), chunks, W)
        out["reads"] = n
        out["write_inputs_s"] = round(time.perf_counter() - t0, 2)
        digests = {}
        for kind in ("fasta", "bfast"):
            od = os.path.join(d, "out_" + kind)
            os.mkdir(od)
            best = None
            for rep in range(2):                       # the second run reads the query file from the page cache
                r = subprocess.run([exe, "-t", os.path.join(d, "ref.tre"), "-s", os.path.join(d, "ref.fasta"),
                                    "-q", os.path.join(d, "q." + kind), "-m", model, "-w", od,
                                    "--stats-json", os.path.join(od, "stats.json")], capture_output=True, text=True, timeout=600)
                if r.returncode != 0:
                    raise RuntimeError("epa-ng-amd failed on the %s file: %s" % (kind, (r.stdout + r.stderr)[-400:]))
                st = json.load(open(os.path.join(od, "stats.json")))
                if best is None or st["loop_s"] < best["loop_s"]:
                    best = st
            h = hashlib.sha256()
            with open(os.path.join(od, "epa_result.jplace"), "rb") as f:
                for line in f:
                    if b'"invocation"' not in line:   # the command line differs by the query file's name
                        h.update(line)
            digests[kind] = h.hexdigest()[:16]
            out[kind] = {"reads_per_s": round(n / best["loop_s"], 1), "reads_per_s_incl_setup": round(n / best["elapsed_s"], 1),
                         "jplace_bytes": os.path.getsize(os.path.join(od, "epa_result.jplace")),
                         "query_file_bytes": os.path.getsize(os.path.join(d, "q." + kind)),
                         "stages_s": {k: round(v, 4) for k, v in best.items() if k.endswith("_s")}}
        out["jplace_identical"] = digests["fasta"] == digests["bfast"]
        out["jplace_sha16"] = digests
        out["note"] = ("epa-ng-amd end to end on %d reads, defaults (chunk 50000), best of two runs; loop_s = first read -> "
                       "closed jplace; stages are busy seconds of overlapping host stages on host_cores threads" % n)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def sclk_fields(clks, exec_tflops):
    """the shader clock the timed Newton launches really ran at (in-kernel s_memtime over s_memrealtime of a
    wave that lives as long as the launch: epa_dev_last_sclk_mhz) and the kernel's fp64 rate against the
    peak AT THAT CLOCK: 256 CUs x 4 SIMDs x 16 fp64 FMA lanes x 2 flop x sclk"""
    clks = [c for c in clks if c > 0]
    if not clks:
        return {"sclk_mhz": None, "frac_at_measured_clock": None}
    mhz = float(np.mean(clks))
    peak = 256 * 4 * 16 * 2 * mhz * 1e6 / 1e12
    return {"sclk_mhz": round(mhz, 1), "sclk_mhz_per_launch": [round(c) for c in clks[:32]], "peak_at_measured_clock": round(peak, 2),
            "frac_at_measured_clock": round(exec_tflops / peak, 4),
            "sclk_note": "spec peak assumes 2400 MHz; sclk_mhz = shader cycles / wall time of the launch's first wave "
                         "(profiles/r5_fma_clock.txt: the clock and cycles per FMA of a bare v_fma_f64 stream)"}


def cfg5_leg(a):
    """BASELINE configs[4] shape on ONE GPU, the `prescoring == false` branch of the chunk body
    (src/core/place.cpp:219-231: Work = all B x Q pairs, every one gets the Newton-Raphson BLO): 4000-tip
    DNA reference (B = 7997, per-rate scalers as the reference turns on above 2000 tips,
    src/io/file_io.cpp:211-214), `--chunk` reads x 7997 branches through epa_dev_place_all (thorough on every
    pair, LWR over all branches + filter on the device).  Prints one JSON object: pairs/s, the Newton
    kernel's roofline, and a 6-read parity sample against the oracle (47 982 pairs)."""
    import epa_ng_amd as epa
    from epa_ng_amd import hostlib, synth
    root = synth.random_tree(4000, 21)
    rates = synth.gamma_rates(synth.CFG2_ALPHA)
    labels, seqs = synth.simulate_msa(root, a.width, synth.CFG2_SUBST, synth.CFG2_FREQS, rates, 22)
    newick = synth.newick(root)
    ref = hostlib.Reference(newick, labels, seqs, states=4, subst=synth.CFG2_SUBST, freqs=synth.CFG2_FREQS, rates=rates)
    ev = ref.evaluator(device=0, rate_scalers=True)
    B, Q, nq = ref.B, a.chunk, a.read_len
    codes, wb, ws = synth.make_reads_compact(seqs, Q, nq, 0.03, 23, 4)
    wire = epa.pack_codes_4bit(codes)
    times, th, stats = [], [], None
    for i in range(a.warmup + a.steps):
        t0 = time.perf_counter()
        got = ev.place_all(wire, wb, ws, Q=Q, min_lwr=0.01, filter_min=1, filter_max=7, max_span=nq)
        if i >= a.warmup:
            times.append(time.perf_counter() - t0)
            th.append(ev.kernel_ms("thorough"))
            stats = dict(ev.last_stats)
    pairs = float(stats["pairs"])
    R = stats["rounds"] / pairs
    kbar = stats["newton_evals"] / max(1.0, 2.0 * stats["rounds"])
    flops_pair = nq * (884.0 + R * (1258.0 + 240.0 * kbar))
    flops_exec = nq * (452.0 + R * (1258.0 + 240.0 * kbar))          # the initial inner CLV comes from the lookup precompute
    t_th = float(np.mean(th)) * 1e-3
    t_all = float(np.mean(times))
    out = {"workload": "cfg5 shape: 4000-tip DNA GTR+G4 ref (B=%d, per-rate scalers), W=%d, %d x %d bp reads, --no-heur: "
                       "all B x Q pairs through epa_dev_place_all (NR BLO on every pair, LWR + filter on device)" % (B, a.width, Q, nq),
           "value": round(pairs / t_all, 1), "unit": "pairs/s", "reads_per_s": round(Q / t_all, 2),
           "pairs_per_call": int(pairs), "ms_per_call": round(t_all * 1e3, 3), "calls_timed": a.steps,
           "kept_per_query": round(float(np.mean([len(g[0]) for g in got])), 3),
           "roofline": {"bound": "fp64-valu", "kernel": "k_thorough_dna", "achieved": round(pairs * flops_exec / t_th / 1e12, 3),
                        "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(pairs * flops_exec / t_th / 1e12 / FP64_PEAK_TFLOPS, 4),
                        "achieved_algorithmic": round(pairs * flops_pair / t_th / 1e12, 3),
                        "frac_algorithmic": round(pairs * flops_pair / t_th / 1e12 / FP64_PEAK_TFLOPS, 4),
                        "ms_per_launch": round(t_th * 1e3, 3), "pairs_per_launch": int(pairs),
                        **sclk_fields([ev.sclk_mhz()], pairs * flops_exec / t_th / 1e12),
                        "rounds_per_pair": round(R, 3), "newton_iters_per_solve": round(kbar, 3),
                        "traffic": None, "traffic_source": "no PMC pass of this leg"},
           "eight_gpu_projection": "cfg5 = 10^6 reads over 8 GPUs = 125 000 reads = %.3g pairs per GPU: %.1f s at this rate"
                                   % (125000.0 * B, 125000.0 * B / (pairs / t_all))}
    if a.parity_sample > 0:
        import oracle_lib
        from oracle_lib import Oracle
        hostlib.configure_threads()
        ns = min(a.parity_sample, Q)
        sample = synth.compact_to_ascii(codes[:ns], wb[:ns], ws[:ns], a.width, 4)
        o = Oracle(newick, labels, seqs, 4, synth.CFG2_SUBST, synth.CFG2_FREQS, rates, rate_scalers=True)
        pb = np.repeat(np.arange(B), ns)
        ps = np.tile(np.arange(ns), B)
        al, ap, ad = o.thorough(pb, ps, sample)
        al = al.reshape(B, ns)
        dl, edges_same = 0.0, 0
        c6 = np.ascontiguousarray(codes[:ns])
        got6 = ev.place_all(epa.pack_codes_4bit(c6), wb[:ns].copy(), ws[:ns].copy(), Q=ns, min_lwr=0.01, filter_min=1,
                            filter_max=7, max_span=nq)
        rounds_equal = bool(ev.last_stats["rounds"] == o.last_stats["rounds"])
        for q in range(ns):
            bid, l = got6[q][0], got6[q][1]
            assert np.array_equal(bid, got[q][0]) and np.array_equal(l, got[q][1])   # a read's result does not depend on the call's size
            dl = max(dl, float(np.max(np.abs(l - al[bid, q]))))
            order = np.lexsort((np.arange(B), -al[:, q]))[:len(bid)]
            edges_same += int(np.array_equal(np.sort(order), np.sort(bid)))
        out["parity"] = {"reads_checked": ns, "pairs_checked": int(ns * B),
                         "kept_placements_max_abs_dlnl": dl, "kept_edge_sets_equal": "%d / %d" % (edges_same, ns),
                         "optimiser_rounds_equal_oracle": rounds_equal}
    print(json.dumps(out))


def main():
    a = parse()
    if a.workload == "cfg5":
        return cfg5_leg(a)
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
    if os.environ.get("EPA_BENCH_ONE_GPU") == "1":
        local = 0     # plumbing runs on a 1-GPU box: every rank on device 0, the gather over tests/fake_rccl.cpp
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

    # ---------------- workload (identical reference on every rank)
    if a.scaling == "strong" and a.reads:
        a.steps = -(-a.reads // (a.chunk * world))
    n_steps = a.steps + a.warmup
    n_chunks = min(n_steps, max(a.pool, a.warmup + 2))   # distinct chunks per rank; step i uses chunk i % n_chunks
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
    ref = hostlib.Reference(newick, labels, seqs, states=states, subst=subst, freqs=freqs, rates=rates)
    ev = ref.evaluator(device=local)
    # the library's kernels go to torch's current stream: everything torch enqueues around a chunk
    # call (the packing for the result gather) is stream-ordered with it, no host synchronisation
    ev.set_stream(torch.cuda.current_stream().cuda_stream)
    if os.environ.get("EPA_BENCH_AA_VALU"):        # A/B: the lane = site VALU kernel for the 20-state windows
        ev.set_option("aa_valu", 1)
    for kv in filter(None, os.environ.get("EPA_BENCH_OPTS", "").split(",")):   # A/B switches: key=value,... (epa_dev_set_option)
        k_, v_ = kv.split("=")
        ev.set_option(k_, int(v_))
    ev.build_lookup()
    torch.cuda.synchronize()
    lookup_ms = ev.kernel_ms("lookup")
    B, W = ref.B, ref.W

    # reads of this rank, generated straight in the compact wire layout.  A step of the job is
    # chunk x world consecutive reads of the stream; rank r holds the contiguous slice
    # local_seq_package gives it (src/net/epa_mpi_util.cpp:10-30), i.e. its own `chunk` reads: slice
    # (step c, rank r) of the stream is generated from seed index c * world + r, nothing else.
    assert parallel.local_seq_package(a.chunk * world, rank, world) == (a.chunk * rank, a.chunk)
    host_chunks = []
    for c in range(n_chunks):
        g = c * world + rank
        codes, wb, ws = synth.make_reads_compact(seqs, a.chunk, a.read_len, 0.03, seeds[2] + 1000003 * g, states)
        wire = epa.pack_codes_4bit(codes) if states == 4 else codes   # what crosses PCIe
        host_chunks.append((codes, wb, ws, wire))
    Q = len(host_chunks[0][1])                    # reads per step on this rank
    step_reads = a.chunk * world                  # reads per step, whole job
    dev_chunks = [(torch.from_numpy(c).to(dev), torch.from_numpy(b.view(np.int32)).to(dev),
                   torch.from_numpy(s.view(np.int32)).to(dev)) for c, b, s, _ in host_chunks]
    cap = max(Q, 1) * 64   # pair capacity per step (dynamic heuristic selects a handful per read)
    d_pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    d_res = torch.zeros((cap, 3), dtype=torch.float64, device=dev)

    th_ms, th_pairs, th_rounds, th_evals, pre_ms, sel_ms, th_clk = [], [], [], [], [], [], []

    rank_elapsed = {}

    def timed(loop_body, finish, tag, n_timed=None, n_warm=None, record=True):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides, MAX over ranks;
        the per-rank times (before the MAX) are kept in rank_elapsed[tag]"""
        n_warm = a.warmup if n_warm is None else n_warm
        n_timed = a.steps if n_timed is None else n_timed
        region = getattr(loop_body, "region", None)      # deep pipelines: chunks are begun ahead inside [lo, hi) only
        if region is not None:
            region[:] = [0, n_warm]
        for i in range(n_warm):
            loop_body(i, False)
        finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if region is not None:
            region[:] = [n_warm, n_warm + n_timed]
        t0 = time.perf_counter()
        for i in range(n_warm, n_warm + n_timed):
            loop_body(i, record)
        finish()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            mt = torch.tensor([mine], dtype=torch.float64, device=cdev)
            allm = [torch.zeros_like(mt) for _ in range(world)]
            dist.all_gather(allm, mt)
            rank_elapsed[tag] = [float(x.item()) for x in allm]
        else:
            rank_elapsed[tag] = [mine]
        return el

    # ---------------- loop 1: inputs resident in HBM (the contract's `value`)
    # rows per gather: 4 candidates per read on average is ample for the dynamic heuristic at
    # 0.99999 (2.6 measured on cfg2); a chunk beyond it is carried into the next gather
    rows_cap = min(cap, 4 * max(Q, 1))
    # N > 1: the gather is the PRODUCT's (libepa_dev.so: epa_comm_create / epa_dev_gather_results /
    # epa_comm_collect / epa_comm_flush, csrc/comm.hip -- RCCL bound by the library itself);
    # torch.distributed only launches, barriers and broadcasts the 128-byte id.  EPA_BENCH_GATHER=torch
    # selects the torch.distributed harness of the same protocol (epa_ng_amd/parallel.py) instead.
    gather_path = os.environ.get("EPA_BENCH_GATHER", "epa_comm") if world > 1 else None

    # The first real N > 1 run must not be able to lose the line: the product's communicator
    # is created and PROBED -- one one-row gather + one all-reduce through everything a real gather uses, every wait
    # bounded by probe_s -- inside the collective agreement below; a transport that cannot move data between these
    # processes costs probe_s seconds, then all ranks abort it and run the torch.distributed harness together.
    probe_s = float(os.environ.get("EPA_BENCH_PROBE_S", "45"))
    run_timeout_s = float(os.environ.get("EPA_BENCH_COMM_TIMEOUT_S", "180"))
    gather_info = {"rccl_path": None, "devices": None, "probe_seconds": None}

    def bind_transport():
        """the product binds the librccl this process has ALREADY mapped (torch's own copy once the nccl backend is
        up) -- never a second RCCL beside it; EPA_RCCL_LIB (the tests' stand-in) wins"""
        if os.environ.get("EPA_RCCL_LIB"):
            return
        path = epa.mapped_rccl_path()
        if path:
            try:
                epa.comm_set_library(path)
            except epa.EpaError:
                pass          # already loaded by an earlier communicator of this process: the same file

    class ProductGather:
        """post / finish / carried_rows of parallel.AsyncResultGather, on api.Comm"""

        def __init__(self, host_copy):
            bind_transport()
            epa.comm_set_default_timeout(probe_s)            # bounds ncclCommInitRank as well
            idt = torch.zeros(128, dtype=torch.uint8, device=cdev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(epa.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            self.depth = 2
            self.comm = None
            t0 = time.perf_counter()
            self.comm = epa.Comm(ev, bytes(idt.cpu().numpy().tobytes()), rank, world, rows_cap, depth=self.depth)
            devices = self.comm.probe(probe_s)               # raises within ~probe_s if the transport does not deliver
            self.comm.set_timeout(run_timeout_s)
            if rank == 0:
                gather_info.update(rccl_path=epa.comm_library_path(), devices=devices,
                                   probe_seconds=round(time.perf_counter() - t0, 3))
            self.host_copy, self.posted, self.collected, self.rows_seen = host_copy, 0, 0, 0
            self.counts = []

        def _collect(self, t):
            if self.host_copy:
                got = self.comm.collect(t)                 # VALID rows of every rank -> pinned host memory
                cnt = [len(g) for g in got]
            else:
                cnt = self.comm.collect_counts(t)          # rows stay in HBM (the contract's `value`)
            self.counts.append(cnt)
            self.rows_seen += sum(cnt)

        def post(self, pairs, res, n):
            # global sequence id of this rank's first read of the step (contiguous slices, local_seq_package)
            off = ((self.posted * world + rank) * a.chunk) & 0x7fffffff
            if rank == 0:
                # the laziest schedule the slots allow: gather T reuses the slot of gather T - depth, so that one is
                # collected right before T is posted -- it has had `depth` whole steps to land, rank 0's host never
                # waits for a transfer that is still queued behind the next chunk's kernels
                for u in range(self.collected, self.posted - self.depth + 1):
                    self._collect(u)
                self.collected = max(self.collected, self.posted - self.depth + 1)
            self.comm.post(pairs, res, n, seq_offset=off)
            self.posted += 1

        def finish(self):
            if rank == 0:
                for t in range(self.collected, self.posted):
                    self._collect(t)
                self.collected = self.posted
            extra = self.comm.flush(on_ticket=self._collect if rank == 0 else None)
            self.posted += len(extra)                      # tickets count the flush's extra gathers as well
            self.collected = self.posted
            torch.cuda.synchronize()

        @property
        def carried_rows(self):
            return self.comm.carried_rows

    gather_note = []

    def make_gather(host_copy):
        nonlocal gather_path
        if world == 1:
            return None
        if gather_path != "torch":
            # collective creation: if ANY rank cannot create the product's communicator (no RCCL library, an
            # ncclCommInitRank failure) every rank falls back to the torch.distributed harness together --
            # the line then says so instead of the job hanging half way
            g, err = None, ""
            holder = {}
            try:
                g = ProductGather.__new__(ProductGather)
                holder["g"] = g
                g.__init__(host_copy)
            except Exception as e:  # noqa: BLE001
                err, g = repr(e), None
            ok = torch.tensor([1 if g is not None else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                gather_info["devices_known"] = True       # (on every rank: rank 0 holds the list)
                return g
            half = holder.get("g")
            if half is not None and getattr(half, "comm", None) is not None:
                half.comm.abort()            # ncclCommAbort: nobody waits for a peer that gave up
                half.comm = None
            errs = [None] * world
            dist.all_gather_object(errs, err)
            gather_note.append("epa_comm gather unavailable (%s): torch.distributed harness used"
                               % "; ".join("rank %d: %s" % (r_, e_) for r_, e_ in enumerate(errs) if e_))
            gather_path = "torch"
        if not gather_info.get("devices_known"):
            # the torch harness: still say which devices the ranks sat on
            gather_info["devices_known"] = True
            devs = [None] * world
            pr = torch.cuda.get_device_properties(local)
            dist.all_gather_object(devs, "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                                             getattr(pr, "pci_device_id", 0)))
            gather_info["devices"] = devs
        return parallel.AsyncResultGather(dist, rows_cap, dev, host_copy=host_copy)

    exch = make_gather(False)

    # Both loops run the library's two-slot chunk pipeline in the same host order,
    #     launch_begin(i); finish(i-1); stage(i+1); launch_end(i)
    # (begin queues preplacement + selection and returns; the previous chunk is retired and the next
    # one staged while they run; end waits for the candidate count and queues the Newton kernels).
    # Loop 1 stages HBM-resident chunks (read in place, results stay in HBM: the contract's `value`),
    # loop 2 host chunks (H2D of the 4-bit codes + D2H of pairs / results inside the clock).  A
    # plain place_chunk() per step measures the same kernels plus one exposed host round trip between
    # steps (exp/pcie_probe.py: 6.90 vs 6.84 ms per step).
    bufs = [(d_pairs, d_res), (torch.empty_like(d_pairs), torch.empty_like(d_res))]

    def make_loop(resident, gather):
        st = {"staged": None, "inflight": None, "rec": False, "bytes_up": 0, "bytes_down": 0, "n_of_slot": {},
              "t_stage": 0.0, "t_launch": 0.0, "t_end": 0.0, "t_finish": 0.0, "timed": False}

        def stage(i):
            t = time.perf_counter()
            if resident:
                dc, dwb, dws = dev_chunks[i % n_chunks]
                ev.chunk_stage(i & 1, dc, dwb, dws)              # HBM-resident: read in place, no copy
            else:
                _, hb, hs, wire = host_chunks[i % n_chunks]
                ev.chunk_stage(i & 1, wire, hb, hs)              # host -> pinned -> async H2D (copy stream)
                st["bytes_up"] += (wire.data if isinstance(wire, epa.Packed4) else wire).nbytes + 8 * Q
            st["t_stage"] += (time.perf_counter() - t) if i > a.warmup else 0.0   # timed steps only
            st["staged"] = i

        def retire(slot):
            t = time.perf_counter()
            if resident or world > 1:
                n = ev.chunk_finish_device(slot)
                if world > 1:
                    # the path's only exchange: every rank's candidate placements -> rank 0 (RCCL over
                    # xGMI; the reference gathers jplace byte ranges, src/io/jplace_writer.hpp:117-129).
                    # Posted asynchronously: it overlaps the next chunk's kernels (parallel.py).
                    gather.post(bufs[slot][0], bufs[slot][1], n)
            else:
                p, r = ev.chunk_finish(slot, copy=False)        # views of the slot's pinned host buffer
                n = len(p)
            st["t_finish"] += (time.perf_counter() - t) if st["timed"] else 0.0
            st["n_of_slot"][slot] = n
            st["bytes_down"] += n * 32
            if st["rec"]:                                        # the retired chunk's Newton kernel
                th_ms.append(ev.kernel_ms("thorough")); th_pairs.append(n); th_clk.append(ev.sclk_mhz())
                th_rounds.append(ev.last_stats["rounds"]); th_evals.append(ev.last_stats["newton_evals"])
            st["inflight"] = None

        def step(i, record):
            if not Q:
                if world > 1:
                    gather.post(d_pairs, d_res, 0)
                return
            slot = i & 1
            if st["staged"] != i:                                # first step of a loop: nothing prefetched
                stage(i)
            kw = dict(pairs_out=bufs[slot][0], results_out=bufs[slot][1], keep_on_device=True) if (resident or world > 1) else {}
            # EPA_CHUNK_HOST_ORDERED: this loop touches a slot's rows only after chunk_finish returned (N > 1: the gather's
            # pack kernel reads them stream-ordered, so the library's stream ordering stays on)
            kw["host_ordered"] = world == 1
            t = time.perf_counter()
            ev.chunk_launch_begin(slot, threshold=0.99999, max_span=a.read_len, max_pairs=cap, **kw)
            st["t_launch"] += (time.perf_counter() - t) if i >= a.warmup else 0.0
            if st["inflight"] is not None:                       # previous chunk: D2H / gather behind its Newton kernel
                retire(st["inflight"])
            st["timed"] = i > a.warmup
            if i + 1 < n_steps and i + 1 != a.warmup:            # next chunk under this one's kernels
                stage(i + 1)                                     # (never across the warmup / timed boundary)
            t = time.perf_counter()
            ev.chunk_launch_end(slot)
            st["t_end"] += (time.perf_counter() - t) if i >= a.warmup else 0.0
            if record and resident:                              # this chunk's preplacement + selection are complete
                pre_ms.append(ev.kernel_ms("preplace")); sel_ms.append(ev.kernel_ms("select"))
            st["rec"] = bool(record and resident)
            st["inflight"] = slot

        def finish():
            if st["inflight"] is not None:
                retire(st["inflight"])
            st["rec"] = False
            if world > 1:
                gather.finish()

        return st, step, finish

    # The schedule `value` is measured on: a DEEP pipeline -- S = A + 2
    # slots, A chunks begun ahead.  Per step k:  launch_end(k); finish(k - 2); stage(k + A); launch_begin(k + A)  -- chunk
    # k's Newton kernel is queued while chunk k - 1's still runs and fills its tail wave by wave, the preplacement +
    # selection chains of chunks k + 1 .. k + A are already queued on their own streams.  Bit-identical rows
    # (tests/test_gpu_pipeline.py: the five-slot order equals place_chunk).  Per-launch kernel durations overlap in
    # this order, so the roofline's ms_per_launch / kernel_ms_per_step come from a SERIALISED pass of the two-slot
    # order above over the same chunks (outside `value`'s clock; the line says so).
    S_DEEP = int(os.environ.get("EPA_BENCH_SLOTS", "5"))
    A_DEEP = int(os.environ.get("EPA_BENCH_AHEAD", str(max(1, S_DEEP - 2))))
    LAG = S_DEEP - A_DEEP                       # finish(k - LAG) right after launch_end(k)
    assert 1 <= A_DEEP and LAG >= 1
    bufs_deep = [bufs[0], bufs[1]] + [(torch.empty_like(d_pairs), torch.empty_like(d_res)) for _ in range(S_DEEP - 2)]

    def make_loop_deep(resident, gather):
        st = {"bytes_up": 0, "bytes_down": 0, "n_of_slot": {}, "begun": set(), "ended": [],
              "t_stage": 0.0, "t_launch": 0.0, "t_end": 0.0, "t_finish": 0.0}
        region = [0, 0]

        def begin(j, timed_):
            slot = j % S_DEEP
            t = time.perf_counter()
            if resident:
                dc, dwb, dws = dev_chunks[j % n_chunks]
                ev.chunk_stage(slot, dc, dwb, dws)
            else:
                _, hb, hs, wire = host_chunks[j % n_chunks]
                ev.chunk_stage(slot, wire, hb, hs)
                st["bytes_up"] += (wire.data if isinstance(wire, epa.Packed4) else wire).nbytes + 8 * Q
            t1 = time.perf_counter()
            kw = dict(pairs_out=bufs_deep[slot][0], results_out=bufs_deep[slot][1], keep_on_device=True) if (resident or world > 1) else {}
            kw["host_ordered"] = world == 1
            ev.chunk_launch_begin(slot, threshold=0.99999, max_span=a.read_len, max_pairs=cap, **kw)
            if timed_:
                st["t_stage"] += t1 - t
                st["t_launch"] += time.perf_counter() - t1
            st["begun"].add(j)

        def retire(j, timed_):
            slot = j % S_DEEP
            t = time.perf_counter()
            if resident or world > 1:
                n = ev.chunk_finish_device(slot)
                if world > 1:
                    gather.post(bufs_deep[slot][0], bufs_deep[slot][1], n)
            else:
                p, r = ev.chunk_finish(slot, copy=False)
                n = len(p)
            if timed_:
                st["t_finish"] += time.perf_counter() - t
            st["n_of_slot"][slot] = n
            st["bytes_down"] += n * 32

        def step(i, record):
            if not Q:
                if world > 1:
                    gather.post(d_pairs, d_res, 0)
                return
            timed_ = i >= a.warmup and region[0] > 0
            for j in range(i, min(i + A_DEEP, region[1])):       # steady state: only i + A - 1 is new here ...
                if j not in st["begun"]:
                    begin(j, timed_)
            t = time.perf_counter()
            ev.chunk_launch_end(i % S_DEEP)
            if timed_:
                st["t_end"] += time.perf_counter() - t
            st["begun"].discard(i)
            st["ended"].append(i)
            while len(st["ended"]) > LAG:                         # chunk i - LAG: its Newton kernel is long done
                retire(st["ended"].pop(0), timed_)
            j = i + A_DEEP                                        # ... and the chunk A ahead goes in behind launch_end(i)
            if j < region[1] and j not in st["begun"]:
                begin(j, timed_)

        step.region = region

        def finish():
            while st["ended"]:
                retire(st["ended"].pop(0), False)
            if world > 1:
                gather.finish()

        return st, step, finish

    if a.serial_schedule:
        S_DEEP = 2

        def make_loop_deep(resident, gather):          # noqa: F811  (--serial-schedule: the two-slot order everywhere)
            return make_loop(resident, gather)
    # Resource set-up, outside every clock: each pipeline slot allocates its scratch bank, result buffers and pinned
    # bounce buffers on first use (~1 GB of hipMalloc / hipHostMalloc per slot at 100k reads).  With W >= the slot count
    # the warm-up steps would do it; a short warm-up (the cfg3 leg runs W = 2) would leave it inside the timed steps.
    if Q:
        for slot in range(S_DEEP):
            ev.chunk_stage(slot, *dev_chunks[0])
            ev.chunk_launch(slot, threshold=0.99999, max_span=a.read_len, max_pairs=cap, pairs_out=bufs_deep[slot][0],
                            results_out=bufs_deep[slot][1], keep_on_device=True)
            ev.chunk_finish_device(slot)
            _, hb0, hs0, wire0 = host_chunks[0]
            ev.chunk_stage(slot, wire0, hb0, hs0)
            ev.chunk_launch(slot, threshold=0.99999, max_span=a.read_len, max_pairs=cap)
            ev.chunk_finish(slot, copy=False)
        torch.cuda.synchronize()
    st_res, step_resident, fin_resident = make_loop_deep(True, exch)
    elapsed = timed(step_resident, fin_resident, "resident", record=False)
    # what the LAST TIMED step left in HBM (outside the clock): the rows the `parity` block checks
    last_i = a.warmup + a.steps - 1
    last_out = None
    if rank == 0 and world == 1 and Q:
        n_last = st_res["n_of_slot"].get(last_i % S_DEEP, 0)
        last_out = (last_i % n_chunks, bufs_deep[last_i % S_DEEP][0][:n_last].cpu().numpy().copy(),
                    bufs_deep[last_i % S_DEEP][1][:n_last].cpu().numpy().copy())
    # the serialised pass: the two-slot order, Newton launches never overlap -> per-launch kernel times for the roofline
    class NoGather:                # the serialised pass measures kernels, not the exchange: rows stay where they are
        carried_rows = 0

        def post(self, *_):
            pass

        def finish(self):
            pass

    _, step_serial, fin_serial = make_loop(True, NoGather())
    elapsed_serial = timed(step_serial, fin_serial, "serial", n_warm=1)

    # ---------------- loop 2: PCIe inside the step, overlapped on the copy streams
    exch2 = make_gather(True)
    state, step_pcie, finish_pcie = make_loop_deep(False, exch2)
    elapsed_pcie = timed(step_pcie, finish_pcie, "pcie")

    # ---------------- the fixed-size job of BASELINE configs[3] (cfg4: 10^7 reads) on these N GPUs:
    # ceil(R / (chunk x N)) steps of `chunk` reads per rank, inputs resident, gather to rank 0 included
    strong = None
    if a.strong_reads and states == 4 and not a.no_extras and not (a.scaling == "strong" and a.reads):
        s_steps = -(-a.strong_reads // (a.chunk * world))
        el_s = timed(step_resident, fin_resident, "strong", n_timed=s_steps, n_warm=1, record=False)
        strong = {"reads": s_steps * a.chunk * world, "value": round(s_steps * a.chunk * world / el_s, 2),
                  "unit": "placements/s", "seconds": round(el_s, 4), "steps_per_rank": s_steps,
                  "reads_per_step_per_gpu": Q, "sclk_mhz_last_launch": round(ev.sclk_mhz(), 1),
                  "per_rank_seconds": [round(x, 4) for x in rank_elapsed["strong"]],
                  "note": "BASELINE configs[3] (cfg4): a fixed job of 10^7 reads dealt over the N GPUs in "
                          "contiguous slices, %d-read chunks per GPU cycling through %d distinct resident "
                          "synthetic chunks, result gather to rank 0 inside the clock; compare `value` across "
                          "the N = 1, 2, 4, 8 lines for the strong-scaling curve" % (Q, n_chunks)}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_reads = a.steps * step_reads
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
        flops_exec = nq * (3 * 4 * 20 + R * (4 * P_ + E_ + 2 * kbar * D_))
    cs = 4 * states
    bytes_pair = 2 * nq * cs * 8 + 2 * nq * 4 + nq + 24
    t_th = float(np.mean(th_ms)) * 1e-3
    exec_tflops = pairs * flops_exec / t_th / 1e12
    alg_tflops = pairs * flops_pair / t_th / 1e12
    traffic = traffic_pre = None   # HBM bytes per launch from the committed PMC passes, same workload only
    aa_mfma = states == 20 and a.read_len <= 192 and not os.environ.get("EPA_BENCH_AA_VALU")
    kname = "k_thorough_dna" if states == 4 else ("k_thorough_aa_mfma" if aa_mfma else "k_thorough_aa")
    # HBM bytes per launch come from the committed PMC passes of the SAME kernel sources (hash in the
    # file, profiles/make_traffic.py) and the same workload; anything else reports null
    traffic_note = "no PMC profile of these kernel sources and this workload under profiles/"
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from make_traffic import src_hash
        here = src_hash()
    except Exception:  # noqa: BLE001
        here = None
    for tf in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
        try:
            tall = json.load(open(os.path.join(ROOT, "profiles", tf)))
            tj = tall[kname]
            if tall.get("kernel_sources_sha16") != here:
                continue
            if tj["reads_per_step"] == Q and abs(tj["pairs_per_launch"] - pairs) < 0.02 * pairs:
                # FETCH_SIZE x its calibrated correction (2.0 on gfx950, profiles/r2_traffic_calibration.txt)
                traffic = (tall.get("fetch_correction", 1.0) * tj["fetch_kb"] + tj["write_kb"]) * 1024.0
                traffic_pre = tall.get("k_preplace_pairs" if states == 4 else "k_preplace_sites")
                if traffic_pre:
                    traffic_pre = (tall.get("fetch_correction", 1.0) * traffic_pre["fetch_kb"] + traffic_pre["write_kb"]) * 1024.0
                traffic_note = "profiles/" + tf
                break
        except (OSError, KeyError, ValueError):
            pass
    roof = {"bound": "mfma" if aa_mfma else "fp64-valu", "kernel": kname,
            "achieved": round(exec_tflops, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(exec_tflops / FP64_PEAK_TFLOPS, 4),
            "traffic": traffic, "traffic_source": traffic_note,
            "note": ("achieved / frac price the fp64 flop the kernel EXECUTES (its 20 x 20 products and the Newton "
                     "contraction are v_mfma_f64_4x4x4_4b_f64, no padding); " if aa_mfma else
                     "achieved / frac price the fp64 flop the kernel EXECUTES (zero MFMA instructions: "
                     "fp64 vector FMA kernel); ") + "*_algorithmic price SURVEY 8d's per-pair figure",
            "peak_source": "AMD MI355X spec sheet, fp64 vector = fp64 matrix = 78.6 TFLOP/s (the microarch "
                           "guide lists no fp64 row); measured ceiling of a dependency-free v_fma_f64 "
                           "stream on this chip: %.1f TFLOP/s (profiles/r1_mfma_overlap.txt)" % FP64_MEASURED_CEILING,
            "frac_of_measured_fma_ceiling": round(exec_tflops / FP64_MEASURED_CEILING, 4),
            **sclk_fields(th_clk, exec_tflops),
            "achieved_algorithmic": round(alg_tflops, 3), "frac_algorithmic": round(alg_tflops / FP64_PEAK_TFLOPS, 4),
            "pairs_per_launch": pairs, "rounds_per_pair": round(R, 3), "newton_iters_per_solve": round(kbar, 3),
            "flops_per_pair": round(flops_pair), "flops_executed_per_pair": round(flops_exec),
            "ms_per_launch": round(t_th * 1e3, 4),
            "hbm_algorithmic_GBs": round(pairs * bytes_pair / t_th / 1e9, 1),
            "hbm_frac": round(pairs * bytes_pair / t_th / 1e9 / HBM_PEAK_GBS, 4)}
    try:   # the XCDs' current shares of a Newton launch's pair list (epa_dev_xcd_shares; an eighth each = not adapted)
        roof["xcd_shares"] = [round(float(x), 4) for x in ev.xcd_shares()]
    except Exception:  # noqa: BLE001
        pass
    # preplacement kernel against the HBM roofline, SURVEY 8d's algorithmic bytes for ONE launch:
    # query windows + the Q x B result table + one read of the lookup table
    ncol = 16 if states == 4 else 24
    pre_bytes = Q * nq + 8.0 * Q * B + 8.0 * ncol * W * B
    t_pre = float(np.mean(pre_ms)) * 1e-3
    roof_pre = {"bound": "hbm", "kernel": "k_preplace_pairs" if states == 4 else "k_preplace_sites",
                "achieved": round(pre_bytes / t_pre / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(pre_bytes / t_pre / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic_pre,
                "algorithmic_bytes_per_launch": pre_bytes,
                "ms_per_launch": round(t_pre * 1e3, 4),
                # the kernel's own unit of work: one 8-byte LDS gather per (query, branch, site pair) [DNA]
                # or site [20 states], against the LDS rate of ds_read_b64 (32 lanes / clk / CU, 256 CUs, 2.4 GHz)
                "lds_gathers_per_launch": float(Q) * B * (nq / 2.0 if states == 4 else nq),
                "lds_gather_frac": round(float(Q) * B * (nq / 2.0 if states == 4 else nq) / t_pre / (256 * 32 * 2.4e9), 4)}

    extras = {}
    # ---------------- CPU baseline: the oracle's OpenMP restatement on a bounded sample
    cpu = None
    parity = None
    if (not a.no_cpu_baseline or a.parity_sample > 0) and world == 1:   # rank 0 at N=1 only (bench contract)
        # the oracle's OpenMP threads = the CPUs this process may really use (a container can see
        # 256 CPUs and be limited to 16 by its cgroup quota: oversubscribing would slow the baseline)
        eff = hostlib.configure_threads()
        import oracle_lib
        from oracle_lib import Oracle
        ns = min(a.parity_sample if a.no_cpu_baseline else a.cpu_sample, Q)
        # the sample = the first ns reads of the chunk the LAST TIMED step placed; the (pair, result) rows
        # checked against the oracle are the ones that step left in its output buffers
        hc, hb, hs, _ = host_chunks[last_out[0]]
        sample = synth.compact_to_ascii(hc[:ns], hb[:ns], hs[:ns], W, states)
        codes, wb, ws = epa.encode_queries(states, sample)
        lnl_gpu = ev.preplace(codes, wb, ws)              # the Q x B table is internal to the chunk body: recomputed
        prs_sep = ev.select(lnl_gpu, ns, 0.99999)
        keep = last_out[1][:, 1] < ns                     # rows of the sample's reads, branch-major order kept
        prs = np.zeros(int(keep.sum()), epa.PAIR_DTYPE)
        prs["branch_id"], prs["seq_id"] = last_out[1][keep, 0], last_out[1][keep, 1]
        res_gpu = np.zeros(len(prs), epa.RESULT_DTYPE)
        for j, k_ in enumerate(("lnl", "pendant_length", "distal_length")):
            res_gpu[k_] = last_out[2][keep, j]
        same_candidates = bool(len(prs) == len(prs_sep) and np.array_equal(prs["branch_id"], prs_sep["branch_id"])
                               and np.array_equal(prs["seq_id"], prs_sep["seq_id"]))
        # timed leg: the oracle's source built with full optimisation for THIS host CPU
        # (oracle_lib.FAST_CFLAGS); parity leg: the strict build (-O2, no FMA contraction), untimed
        if not a.no_cpu_baseline:
            of = Oracle(newick, labels, seqs, states, subst, freqs, rates, fast=True)
            of.preplace(sample[:8])                     # builds the per-branch lookups (one-off)
            c0 = time.perf_counter()
            of.preplace(sample)
            tl_fast = of.thorough(prs["branch_id"], prs["seq_id"], sample)[0]
            cpu_t = time.perf_counter() - c0
            cores = min(eff, of.L.orc_max_threads())
            del of
        o = Oracle(newick, labels, seqs, states, subst, freqs, rates)
        s0 = time.perf_counter()
        lnl_cpu = o.preplace(sample)
        tl, tp, td = o.thorough(prs["branch_id"], prs["seq_id"], sample)
        strict_t = time.perf_counter() - s0         # includes the strict build's lookup construction
        sc_at = o.score_at(prs["branch_id"], prs["seq_id"], sample, res_gpu["pendant_length"], res_gpu["distal_length"])
        if not a.no_cpu_baseline:
            cpu = {"value": round(ns / cpu_t, 2), "unit": "placements/s", "cores": cores, "kind": "port",
                   "cflags": oracle_lib.FAST_CFLAGS,
                   "sample": "%d reads of step 0 through the oracle's OpenMP preplace (B=%d) + thorough "
                             "(%d pairs), lookups prebuilt" % (ns, B, len(prs)),
                   "strict_build_seconds_same_sample": round(strict_t, 2),
                   "kernels": ("4-state x 4-category loops in libpll's AVX2 shape (states of a site = one vector, transposed "
                               "matrices, omp simd; oracle/epa_oracle.c ORC_FAST_KERNELS)" if states == 4 else
                               "20-state x 4-category loops in libpll's AVX2 shape (matrix-vector products accumulate whole columns "
                               "of the transposed matrix, five 4-wide FMAs per entry, omp simd; oracle/epa_oracle.c ORC_FAST_KERNELS)")
                              + "; the preplacement gathers are scalar as in src/core/Lookup_Store.hpp:110-141",
                   "fast_vs_strict_max_abs_dlnl": float(np.max(np.abs(tl_fast - tl)))}
        # the optimiser-path rule of the parity sweep (tests/sweep_util.py) on this sample: pairs whose
        # lengths differ from the oracle's own must be reproduced by a rounded sibling of the oracle
        import sweep_util
        flat = sweep_util.lengths_differ(res_gpu["pendant_length"], res_gpu["distal_length"], tp, td)
        rep = sweep_util.reproduce_flat_pairs(o, sample, prs["branch_id"], prs["seq_id"], res_gpu, flat,
                                              dl=np.abs(res_gpu["lnl"] - tl))
        parity = {"preplace_max_abs_dlnl": float(np.max(np.abs(lnl_gpu - lnl_cpu))),
                  "thorough_max_abs_dlnl": float(np.max(np.abs(res_gpu["lnl"] - tl))),
                  "evaluator_max_abs_dlnl_at_device_lengths": float(np.max(np.abs(res_gpu["lnl"] - sc_at))),
                  "pairs_checked": int(len(prs)), "reads_checked": int(ns),
                  "rows_checked": "pairs / results the last timed step (chunk %d) left in its HBM output buffers" % last_out[0],
                  "timed_step_candidates_equal_separate_select_call": same_candidates,
                  "flat_pairs": rep["flat_pairs"], "flat_reproduced": rep["flat_reproduced"],
                  "max_variant": rep["max_amplitude_log2_ulp"], "flat_decisions": rep["decisions"],
                  "max_variant_unit": "log2 ulp of the rounding sibling needed (0 with no flat pair)"}
        if not a.no_extras and not a.no_cpu_baseline:
            # the reference's own executable, if the box happens to have one (it never did so far)
            nref = min(ns, 2000)
            best = {}
            row_best = np.argmax(lnl_gpu[:nref], axis=1)
            for q in range(nref):
                m = prs["seq_id"] == q
                k = int(np.argmax(res_gpu["lnl"][m]))
                best[q] = (int(prs["branch_id"][m][k]), float(res_gpu["lnl"][m][k]))
            del row_best
            model = ("GTR{%s}+FU{%s}+G4{%r}" % ("/".join(map(repr, subst)), "/".join(map(repr, freqs)), alpha)
                     if states == 4 else "PROTGTR{%s}+FU{%s}+G4{%r}" % ("/".join(map(repr, subst)),
                                                                         "/".join(map(repr, freqs)), alpha))
            extras["reference_binary"] = reference_binary_check(newick, labels, seqs, hc[:nref], hb[:nref], hs[:nref],
                                                                W, states, model, best)
    if world == 1 and not a.no_extras and Q >= 5000:
        # the reference's default chunk size (--chunk-size 5000, src/util/Options.hpp) through the
        # same double-buffered pipeline: 40 chunks cut from step 0's reads, PCIe inside the clock
        hc, hb, hs, wire = host_chunks[last_i % n_chunks]
        nsm = min(40, Q // 5000)
        small = []
        for k in range(nsm):
            sl = slice(k * 5000, (k + 1) * 5000)
            c5 = np.ascontiguousarray(hc[sl])
            small.append((epa.pack_codes_4bit(c5) if states == 4 else c5, hb[sl].copy(), hs[sl].copy()))
        for rep in range(4):                                     # the first passes warm the slots' buffers; the last is reported
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            # five slots, three chunks begun ahead: chunk k's Newton kernel is queued (launch_end) while
            # chunk k - 1's still runs and fills its tail wave by wave; preplacement + selection of
            # chunks k + 1 .. k + 3 are already queued on their own streams (launch_begin).  Two slots:
            # 6.6 M/s, five: 8.3 (exp/chunk5000.py)
            hostt = {"stage": 0.0, "begin": 0.0, "end": 0.0, "finish": 0.0}

            def call(name, fn, *aa, **kw):
                t = time.perf_counter()
                fn(*aa, **kw)
                hostt[name] += time.perf_counter() - t
            kw5 = dict(threshold=0.99999, max_span=a.read_len, max_pairs=5000 * 64, host_ordered=True)
            S5, A5 = 5, 3
            for k in range(min(A5, nsm)):
                call("stage", ev.chunk_stage, k % S5, *small[k])
                call("begin", ev.chunk_launch_begin, k % S5, **kw5)
            for k in range(nsm):
                call("end", ev.chunk_launch_end, k % S5)
                if k >= 2:
                    call("finish", ev.chunk_finish, (k - 2) % S5, copy=False)
                if k + A5 < nsm:
                    call("stage", ev.chunk_stage, (k + A5) % S5, *small[k + A5])
                    call("begin", ev.chunk_launch_begin, (k + A5) % S5, **kw5)
            for k in range(max(0, nsm - 2), nsm):
                call("finish", ev.chunk_finish, k % S5, copy=False)
            t5 = time.perf_counter() - t0
        per_chunk = {"value": round(nsm * 5000 / t5, 1), "unit": "placements/s", "chunks": nsm,
                     "ms_per_chunk": round(t5 / nsm * 1e3, 3),
                     "host_ms_per_chunk": {k_: round(v_ / nsm * 1e3, 3) for k_, v_ in hostt.items()},
                     "note": "one chunk body per 5000-read chunk, H2D/D2H inside the clock, five pipeline slots, three chunks begun ahead"}
        # the same 5000-read chunks through GROUP launches (epa_dev_chunk_launch_many): the caller still stages and
        # finishes chunk by chunk, the library runs one chunk body -- one preplacement, one selection, one Newton
        # launch -- per four staged chunks and regroups the rows by chunk.  Twelve slots, two groups begun ahead.
        G = int(os.environ.get("EPA_BENCH_GROUP", "4"))
        small = small * (40 // nsm) if nsm < 40 else small      # 40 chunks: the pipeline's fill / drain amortised
        ngr = len(small) // G
        kwg = dict(threshold=0.99999, max_span=a.read_len, max_pairs=G * 5000 * 64, host_ordered=True)
        slots_of = lambda g: [(g % 3) * G + j for j in range(G)]
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hostg = {"stage": 0.0, "begin": 0.0, "end": 0.0, "finish": 0.0}

            def callg(name, fn, *aa, **kw):
                t = time.perf_counter()
                r_ = fn(*aa, **kw)
                hostg[name] += time.perf_counter() - t
                return r_

            def gbegin(g):
                for j, sl in enumerate(slots_of(g)):
                    callg("stage", ev.chunk_stage, sl, *small[g * G + j])
                callg("begin", ev.chunk_launch_many_begin, slots_of(g), **kwg)
            rows5 = 0
            for g in range(min(2, ngr)):
                gbegin(g)
            for g in range(ngr):
                callg("end", ev.chunk_launch_end, slots_of(g)[0])
                if g >= 1:
                    for sl in slots_of(g - 1):
                        rows5 += len(callg("finish", ev.chunk_finish, sl, copy=False)[0])
                if g + 2 < ngr:
                    gbegin(g + 2)
            for sl in slots_of(ngr - 1):
                rows5 += len(callg("finish", ev.chunk_finish, sl, copy=False)[0])
            t5g = time.perf_counter() - t0
        extras["chunk5000"] = {"value": round(ngr * G * 5000 / t5g, 1), "unit": "placements/s", "chunks": ngr * G,
                               "ms_per_chunk": round(t5g / (ngr * G) * 1e3, 3), "chunks_per_group_launch": G,
                               "rows": rows5,
                               "host_ms_per_chunk": {k_: round(v_ / (ngr * G) * 1e3, 3) for k_, v_ in hostg.items()},
                               "note": "reference default --chunk-size 5000 (src/util/Options.hpp:26) through the C-ABI: stage / finish per "
                                       "5000-read chunk, H2D/D2H inside the clock; epa_dev_chunk_launch_many runs ONE chunk body per four "
                                       "staged chunks (twelve slots, two groups begun ahead); per chunk the same bits as a launch of its own",
                               "per_chunk_launch": per_chunk}

    if world == 1 and states == 4 and not a.no_extras and not os.environ.get("EPA_BENCH_NO_CFG3"):
        # BASELINE configs[2] (cfg3: 2000-tip 20-state reference, 100-residue queries) as a short second
        # measurement by the same script in its own process, so that the driver's line carries it too
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "aa", "--tips", "2000", "--width", "500",
                                "--read-len", "100", "--chunk", "50000", "--steps", "6", "--warmup", "2", "--pool", "3",
                                "--cpu-sample", "1000", "--no-extras"],
                               capture_output=True, text=True, timeout=900)
            aj = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            extras["cfg3_aa"] = {"value": aj["value"], "unit": aj["unit"], "ms_per_step": aj["ms_per_step"],
                                 "workload": aj["config"]["workload"], "reads_per_step": aj["config"]["reads_per_step_per_gpu"],
                                 "kernel_ms_per_step": aj["config"]["kernel_ms_per_step"],
                                 "pcie_inclusive": aj["pcie_inclusive"]["value"],
                                 "roofline": {k: aj["roofline"].get(k) for k in ("bound", "kernel", "achieved", "frac", "achieved_algorithmic",
                                                                                "frac_algorithmic", "pairs_per_launch", "ms_per_launch",
                                                                                "sclk_mhz", "frac_at_measured_clock",
                                                                                "traffic", "traffic_source")},
                                 "roofline_preplace": aj.get("roofline_preplace"),
                                 "cpu_baseline": aj.get("cpu_baseline"),
                                 "parity": aj.get("parity")}
        except Exception as e:  # noqa: BLE001  (a secondary measurement must never take the bench line down)
            extras["cfg3_aa"] = {"status": "failed: %r" % (e,)}
        # BASELINE configs[4] (cfg5) at a 2000-read sample of one GPU's shard: --no-heur, every pair optimised
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "cfg5", "--chunk", "2000", "--steps", "3",
                                "--warmup", "1", "--parity-sample", "6"], capture_output=True, text=True, timeout=900)
            extras["cfg5_noheur"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:  # noqa: BLE001
            extras["cfg5_noheur"] = {"status": "failed: %r" % (e,)}

    if world == 1 and states == 4 and not a.no_extras and not os.environ.get("EPA_BENCH_NO_CLI"):
        try:
            nrd = int(os.environ.get("EPA_BENCH_CLI_READS", "1000000"))
            use = [(c, b, s_) for c, b, s_, _ in host_chunks[:max(1, -(-nrd // max(Q, 1)))]]
            model = "GTR{%s}+FU{%s}+G4{%r}" % ("/".join(map(repr, subst)), "/".join(map(repr, freqs)), alpha)
            extras["cli_e2e"] = cli_e2e_leg(newick, labels, seqs, use, W, model)
        except Exception as e:  # noqa: BLE001  (a secondary measurement must never take the bench line down)
            extras["cli_e2e"] = {"status": "failed: %r" % (e,)}

    metric = ("query placements/sec (whole node), 512-tip GTR+G4 DNA ref, preplace+thorough" if states == 4
              else "query placements/sec (whole node), AA PROTGTR+G4 ref (cfg3 shape), preplace+thorough")
    pcie = {"value": round(total_reads / elapsed_pcie, 2), "unit": "placements/s",
            "ms_per_step": round(elapsed_pcie / a.steps * 1e3, 3),
            "h2d_bytes_per_step": state["bytes_up"] // max(1, n_steps),
            "d2h_bytes_per_step": state["bytes_down"] // max(1, n_steps),
            "host_ms_per_step": {"stage (memcpy to pinned + H2D enqueue)": round(state["t_stage"] / max(1, a.steps - 1) * 1e3, 3),
                                 "launch_begin (queues preplacement + selection)": round(state["t_launch"] / a.steps * 1e3, 3),
                                 "finish (wait for the previous chunk's kernels + D2H)": round(state["t_finish"] / max(1, a.steps - 1) * 1e3, 3),
                                 "launch_end (waits for the candidate count, queues the Newton kernels)": round(state["t_end"] / a.steps * 1e3, 3)},
            "how": "per step: launch_begin(i); finish(i-1); stage(i+1); launch_end(i) -- %s codes + windows up, "
                   "pairs + results down, copies on the copy streams under the previous / next chunk's kernels%s"
                   % ("4-bit" if states == 4 else "1-byte",
                      "; N > 1: results gathered over RCCL to rank 0, which copies them to the host" if world > 1 else "")}
    out = {"metric": metric,
           "value": round(value, 2), "unit": "placements/s", "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
           "schedule": {"order": ("launch_begin(k); finish(k-1); stage(k+1); launch_end(k)  [--serial-schedule]" if a.serial_schedule else
                                  "launch_end(k); finish(k-%d); stage(k+%d); launch_begin(k+%d)" % (LAG, A_DEEP, A_DEEP)),
                        "slots": S_DEEP, "chunks_begun_ahead": 0 if a.serial_schedule else A_DEEP,
                        "ms_per_step_serialised_two_slot_order": (round(elapsed_serial / a.steps * 1e3, 3) if elapsed_serial else None),
                        "note": "value / ms_per_step: the deep pipeline (chunk k's Newton kernel queued into chunk k-1's tail; the "
                                "order the CLI's chunk loop runs); roofline.ms_per_launch, kernel_ms_per_step, sclk: a serialised "
                                "pass of the two-slot order over the same chunks, where launches do not overlap"},
           "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": ("cfg2: %d-tip DNA GTR+G4 ref, W=%d, %d bp reads, dyn-heur 0.99999, "
                                   "preplace+thorough" if states == 4 else
                                   "cfg3-shape: %d-tip AA PROTGTR+G4 ref, W=%d, %d aa queries, dyn-heur "
                                   "0.99999, preplace+thorough") % (a.tips, a.width, a.read_len),
                      "reads_per_step_whole_job": step_reads, "reads_per_step_per_gpu": Q, "branches": B,
                      "parallelism": "query-shard x%d (%s)" % (world, a.scaling),
                      "value_definition": "inputs resident in HBM when the clock starts (bench contract); "
                                          "see pcie_inclusive for SURVEY 8d's H2D/D2H-inclusive rate",
                      "lookup_build_ms_once": round(lookup_ms, 3),
                      "kernel_ms_per_step": {"preplace": round(float(np.mean(pre_ms)), 3),
                                             "select": round(float(np.mean(sel_ms)), 3),
                                             "thorough": round(float(np.mean(th_ms)), 3)}},
           "pcie_inclusive": pcie, "strong_cfg4": strong,
           "rccl_ranks": world if (world > 1 and (dist.get_backend() == "nccl" if gather_path == "torch"
                                                  else not os.environ.get("EPA_RCCL_LIB"))) else 0,
           # rank 0's view of the result gather: which implementation ran, rows per rank and gather, rows that
           # took the carry path (0 expected)
           "gather": ({"path": ("epa_comm: libepa_dev.so epa_dev_gather_results / epa_comm_collect / epa_comm_flush (csrc/comm.hip)"
                                if gather_path != "torch" else "torch.distributed harness (epa_ng_amd/parallel.py AsyncResultGather)"),
                       "transport": (("stand-in " + os.path.basename(os.environ["EPA_RCCL_LIB"]) + " (test infrastructure, not a scaling measurement)")
                                     if (gather_path != "torch" and os.environ.get("EPA_RCCL_LIB")) else
                                     ("RCCL" if (gather_path != "torch" or dist.get_backend() == "nccl") else dist.get_backend())),
                       "fallback": gather_note or None,
                       "rccl_path": gather_info["rccl_path"], "devices": gather_info["devices"],
                       "distinct_devices": len(set(gather_info["devices"] or [])),
                       "probe_seconds": gather_info["probe_seconds"],
                       "rows_cap": rows_cap, "carried_rows_rank0": int(exch.carried_rows + exch2.carried_rows),
                       "rows_collected_rank0": (int(getattr(exch, "rows_seen", 0) + getattr(exch2, "rows_seen", 0)) if gather_path != "torch" else None)}
                      if world > 1 else None),
           "per_rank_ms_per_step": [round(x / a.steps * 1e3, 3) for x in rank_elapsed["resident"]],
           "roofline": roof, "roofline_preplace": roof_pre, "cpu_baseline": cpu, "parity": parity}
    out.update(extras)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
