"""The product library's RCCL gather (include/epa_dev.h epa_comm_*, epa_ng_amd/csrc/comm.hip): the
exchange that replaces src/net/epa_mpi_util.cpp:10-30 + src/io/jplace_writer.hpp:117-129 for the
one-process-per-GPU mode.  On the 1-GPU box: a 1-rank communicator, once with rank 0's rows taken the
local way and once sent to itself through ncclSend / ncclRecv (epa_comm_set_self_send / --comm-self-send), carry path and flush
included; wherever >= 2 GPUs are visible: one process per GPU over the C-ABI and the CLI's --rank mode."""
import os
import subprocess
import sys

import numpy as np
import pytest

import epa_ng_amd as epa
from epa_ng_amd import hostlib, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _workload():
    w = synth.dna_workload(48, 600, 1200, 150, (71, 72, 73))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    return w, ref


def _rows_of(pairs, res, seq_offset):
    rows = np.zeros(len(pairs), epa.ROW_DTYPE)
    rows["branch_id"] = pairs["branch_id"]
    rows["seq_id"] = pairs["seq_id"] + seq_offset
    for k in ("lnl", "pendant_length", "distal_length"):
        rows[k] = res[k]
    return rows


@pytest.mark.parametrize("self_send", [False, True])
def test_one_rank_gather_with_carry_and_flush(self_send, monkeypatch):
    """Three chunks through the staged pipeline with the results kept in HBM, each posted to the gather
    (epa_dev_gather_slot / epa_dev_gather_results); rows_cap is smaller than a chunk's candidate count,
    so every post carries rows into the next one and flush() drains the rest.  What rank 0 collects,
    in ticket order, is exactly the chunks' (pair, result) rows with global sequence ids."""
    import torch
    w, ref = _workload()
    ev = ref.evaluator()
    Q = 400
    chunks = [epa.encode_queries(4, w["reads"][c * Q:(c + 1) * Q], compact=True) for c in range(3)]
    expect = [ev.place_chunk(*ch, max_span=150) for ch in chunks]
    n_exp = [len(p) for p, _ in expect]
    cap = Q * 64
    rows_cap = max(n_exp) * 2 // 3            # forces the carry path
    comm = epa.Comm(ev, epa.comm_unique_id(), 0, 1, rows_cap, depth=2, self_send=self_send)
    ids = comm.probe(30.0)                     # the handshake leaves the communicator as created (tickets from 0)
    assert len(ids) == 1 and len(ids[0].split(":")) == 3
    dev = torch.device("cuda", 0)
    bufs = [(torch.zeros((cap, 2), dtype=torch.int32, device=dev), torch.zeros((cap, 3), dtype=torch.float64, device=dev))
            for _ in range(2)]
    got, tickets = [], []
    for k, ch in enumerate(chunks):
        ev.chunk_stage(k & 1, *ch)
        ev.chunk_launch(k & 1, max_span=150, max_pairs=cap, pairs_out=bufs[k & 1][0], results_out=bufs[k & 1][1],
                        keep_on_device=True)
        if k == 1:   # the explicit-buffer entry point
            n = ev.chunk_finish_device(k & 1)
            tickets.append(comm.post(bufs[k & 1][0], bufs[k & 1][1], n, seq_offset=1000 * k))
        else:        # straight from the slot, no host wait for the kernels
            tickets.append(comm.post_slot(k & 1, seq_offset=1000 * k))
            ev.chunk_finish_device(k & 1)
        if k >= 1:   # collect within `depth` posts
            got.append(comm.collect(tickets[k - 1])[0])
    got.append(comm.collect(tickets[-1])[0])
    extra = comm.flush(on_ticket=lambda t: got.append(comm.collect(t)[0]))
    assert len(extra) >= 1 and comm.carried_rows > 0
    assert all(len(g) <= rows_cap for g in got)
    allrows = np.concatenate(got)
    want = np.concatenate([_rows_of(p, r, 1000 * k) for k, (p, r) in enumerate(expect)])
    assert len(allrows) == len(want)
    assert np.array_equal(allrows, want)
    with pytest.raises(epa.EpaError):
        comm.collect(tickets[0])              # that gather's slot has been reused
    comm.close()


def test_gather_argument_errors():
    w, ref = _workload()
    ev = ref.evaluator()
    uid = epa.comm_unique_id()
    with pytest.raises(epa.EpaError):
        epa.Comm(ev, uid, 1, 1, 100)          # rank outside the world
    with pytest.raises(epa.EpaError):
        epa.Comm(ev, uid, 0, 1, 0)            # no rows
    comm = epa.Comm(ev, uid, 0, 1, 100)
    host_pairs = np.zeros(4, epa.PAIR_DTYPE)
    host_res = np.zeros(4, epa.RESULT_DTYPE)
    with pytest.raises(epa.EpaError):
        comm.post(host_pairs, host_res, 4)    # host memory: the gather reads device buffers
    with pytest.raises(epa.EpaError):
        comm.post_slot(0)                     # nothing launched on the slot
    comm.close()


WORKER = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ["EPA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["EPA_ROOT"], "tests"))
import torch
import epa_ng_amd as epa
from epa_ng_amd import hostlib, synth, parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
idf = os.environ["EPA_COMM_FILE"]
one_gpu = os.environ.get("EPA_TEST_ONE_GPU") == "1"      # every rank on device 0 (transport stand-in, EPA_RCCL_LIB)
w = synth.dna_workload(48, 600, 1200, 150, (71, 72, 73))
reads_all = w["reads"][:int(os.environ.get("EPA_TEST_NREADS", "1200"))]
ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"], freqs=w["freqs"], rates=w["rates"])
ev = ref.evaluator(device=0 if one_gpu else rank)
if rank == 0:
    uid = epa.comm_unique_id()
    open(idf + ".tmp", "wb").write(uid); os.replace(idf + ".tmp", idf)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 120
        time.sleep(0.05)
    uid = open(idf, "rb").read()
off, cnt = parallel.local_seq_package(len(reads_all), rank, world)
Q = int(os.environ.get("EPA_TEST_CHUNK", "200"))
rows_cap = int(os.environ.get("EPA_TEST_ROWS_CAP", str(Q * 8)))
depth = int(os.environ.get("EPA_TEST_DEPTH", "2"))
comm = epa.Comm(ev, uid, rank, world, rows_cap=rows_cap, depth=depth)
dev = torch.device("cuda", 0 if one_gpu else rank)
cap = Q * 64
bufs = [(torch.zeros((cap, 2), dtype=torch.int32, device=dev), torch.zeros((cap, 3), dtype=torch.float64, device=dev)) for _ in range(2)]
nchunks = -(-(-(-len(reads_all) // world)) // Q)       # the same number of posts on every rank (collective)
tickets, got, pend_seen = [], [], []
def collect(t):
    got.extend(comm.collect(t))
    pend_seen.append(list(comm.last_pending))
for k in range(nchunks):
    lo = off + k * Q
    reads = reads_all[lo:max(lo, min(off + cnt, lo + Q))]
    if reads:
        ev.chunk_stage(k & 1, *epa.encode_queries(4, reads, compact=True))
        ev.chunk_launch(k & 1, max_span=150, max_pairs=cap, pairs_out=bufs[k & 1][0], results_out=bufs[k & 1][1], keep_on_device=True)
        tickets.append(comm.post_slot(k & 1, seq_offset=lo))
        ev.chunk_finish_device(k & 1)
    else:
        tickets.append(comm.post(None, None, 0))
    if rank == 0 and k >= depth - 1:                    # gather k's slot is posted again at k + depth
        collect(tickets[k - (depth - 1)])
if rank == 0:
    for k in range(max(0, nchunks - (depth - 1)), nchunks):
        collect(tickets[k])
extra = comm.flush(on_ticket=collect if rank == 0 else None)
print("COMM_RANK rank=%d carried=%d extra=%d" % (rank, comm.carried_rows, len(extra)), flush=True)
if rank == 0:
    rows = np.concatenate(got) if got else np.zeros(0, epa.ROW_DTYPE)
    rows = rows[np.lexsort((rows["branch_id"], rows["seq_id"]))]
    np.save(os.environ["EPA_COMM_OUT"], rows)
    assert pend_seen[-1] == [0] * world, pend_seen[-1]
    print("COMM_GATHER_OK world=%d rows=%d" % (world, len(rows)))
comm.close()
"""


def test_gather_one_process_per_gpu(tmp_path):
    """world = every visible GPU (>= 2): the contiguous query slices of local_seq_package, one process and
    epa_ctx per GPU, every chunk's rows gathered to rank 0 over RCCL through the C-ABI -- and rank 0 holds
    the rows of a single-process run over all reads."""
    n = min(8, epa.device_count())
    if n < 2:
        pytest.skip("needs >= 2 GPUs (RCCL wants one device per rank)")
    script = tmp_path / "comm_worker.py"
    script.write_text(WORKER)
    out = tmp_path / "rows.npy"
    env = dict(os.environ, EPA_ROOT=ROOT, WORLD_SIZE=str(n), EPA_COMM_FILE=str(tmp_path / "uid"), EPA_COMM_OUT=str(out),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(n)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "COMM_GATHER_OK world=%d" % n in outs[0][0]
    rows = np.load(out)
    w, ref = _workload()
    ev = ref.evaluator()
    pairs, res = ev.place_chunk(*epa.encode_queries(4, w["reads"], compact=True), max_span=150)
    want = _rows_of(pairs, res, 0)
    want = want[np.lexsort((want["branch_id"], want["seq_id"]))]
    assert np.array_equal(rows, want)


def _expected_rows(nreads):
    w, ref = _workload()
    ev = ref.evaluator()
    pairs, res = ev.place_chunk(*epa.encode_queries(4, w["reads"][:nreads], compact=True), max_span=150)
    want = _rows_of(pairs, res, 0)
    return want[np.lexsort((want["branch_id"], want["seq_id"]))]


def _expected_carried(want, nreads, world, Q, rows_cap):
    """the carry bookkeeping of epa_dev_gather_results replayed on a single-process run's rows: per rank
    the rows that ever took the carry path"""
    from epa_ng_amd import parallel
    nchunks = -(-(-(-nreads // world)) // Q)
    out = []
    for r in range(world):
        off, cnt = parallel.local_seq_package(nreads, r, world)
        carry = carried = 0
        for k in range(nchunks):
            lo, hi = off + k * Q, max(off + k * Q, min(off + cnt, off + (k + 1) * Q))
            n = int(np.count_nonzero((want["seq_id"] >= lo) & (want["seq_id"] < hi)))
            m = min(carry + n, rows_cap)
            from_new = m - min(carry, m)
            carried += n - from_new
            carry = carry + n - m
        out.append(carried)
    return out


def _run_ranks(cmds_env, timeout=600):
    """starts one process per (argv, env), waits, kills exactly those on a timeout"""
    procs = [subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for argv, env in cmds_env]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    except subprocess.TimeoutExpired:
        for p in procs:
            if p.poll() is None:
                p.kill()
        outs = [p.communicate() for p in procs]
        raise AssertionError("rank processes timed out: %r" % (outs,))
    return procs, outs


# (world, reads, chunk, rows_cap, depth): uneven slices, trailing EMPTY ranks, rows_cap below a chunk's
# candidates (every post carries, the flush needs several calls), depth 1 and 2
ONE_GPU_CASES = [
    (2, 1200, 200, 1600, 2),     # the plain case: nothing carried
    (3, 1000, 100, 40, 2),       # part 334 / 334 / 332; ~250 rows per chunk against rows_cap 40: carry grows, many flush calls
    (3, 1000, 100, 40, 1),       # the same with ONE slot: collect right after every post
    (8, 49, 3, 6, 2),            # part 7: rank 7 is empty (all its posts are empty), last chunk has 1 read; carry
    (8, 1001, 50, 400, 2),       # part 126, rank 7 has 119: trailing partial chunk
    (2, 5, 10, 100, 2),          # fewer reads than a chunk: one post per rank
]


@pytest.mark.parametrize("world,nreads,Q,rows_cap,depth", ONE_GPU_CASES)
def test_gather_n_processes_on_one_gpu_over_the_transport_standin(tmp_path, world, nreads, Q, rows_cap, depth):
    """The 1-GPU twin of test_gather_one_process_per_gpu: `world` processes share device 0 and the product's
    gather (comm.hip) runs over tests/fake_rccl.cpp (EPA_RCCL_LIB) instead of RCCL, which refuses two ranks
    on one device.  Everything rank-count dependent runs for real: the same number of posts on every rank,
    empty trailing slices, slot reuse at `depth`, rank 0's per-rank pending bookkeeping, the all-reduce-agreed
    flush rounds.  Rank 0 holds the rows of a single-process run; every rank's carried_rows equals the
    bookkeeping replayed on those rows."""
    import fake_rccl_util
    script = tmp_path / "comm_worker.py"
    script.write_text(WORKER)
    out = tmp_path / "rows.npy"
    env = fake_rccl_util.env(dict(os.environ, EPA_ROOT=ROOT, WORLD_SIZE=str(world), EPA_COMM_FILE=str(tmp_path / "uid"),
                                  EPA_COMM_OUT=str(out), EPA_TEST_ONE_GPU="1", EPA_TEST_NREADS=str(nreads),
                                  EPA_TEST_CHUNK=str(Q), EPA_TEST_ROWS_CAP=str(rows_cap), EPA_TEST_DEPTH=str(depth)))
    procs, outs = _run_ranks([([sys.executable, str(script)], dict(env, RANK=str(r))) for r in range(world)])
    assert all(p.returncode == 0 for p in procs), outs
    assert "COMM_GATHER_OK world=%d" % world in outs[0][0]
    rows = np.load(out)
    want = _expected_rows(nreads)
    assert np.array_equal(rows, want)
    carried = []
    for r in range(world):
        line = [l for l in outs[r][0].splitlines() if l.startswith("COMM_RANK rank=%d " % r)][0]
        carried.append(int(line.split("carried=")[1].split()[0]))
    assert carried == _expected_carried(want, nreads, world, Q, rows_cap)
    if rows_cap < 100:
        assert sum(carried) > 0       # the case did exercise the carry path


def test_a_failing_rank_does_not_hang_its_peers(tmp_path):
    """rank 1 of 2 dies before its first post: rank 0's collect gives up after EPA_COMM_TIMEOUT_S with an
    error instead of waiting for ever (the reference's MPI build would abort the job)."""
    import fake_rccl_util
    script = tmp_path / "comm_worker.py"
    script.write_text(WORKER.replace("comm = epa.Comm(", "comm = epa.Comm(", 1).replace(
        "tickets, got, pend_seen = [], [], []", "tickets, got, pend_seen = [], [], []\nif rank == 1: os._exit(3)", 1))
    env = fake_rccl_util.env(dict(os.environ, EPA_ROOT=ROOT, WORLD_SIZE="2", EPA_COMM_FILE=str(tmp_path / "uid"),
                                  EPA_COMM_OUT=str(tmp_path / "rows.npy"), EPA_TEST_ONE_GPU="1", EPA_TEST_NREADS="400"),
                             timeout_s=8)
    env["EPA_COMM_TIMEOUT_S"] = "12"
    procs, outs = _run_ranks([([sys.executable, str(script)], dict(env, RANK=str(r))) for r in range(2)], timeout=300)
    assert procs[1].returncode == 3
    assert procs[0].returncode not in (0, None)
    assert "did not complete" in outs[0][1] or "peer" in outs[0][1] or "sentinel" in outs[0][1], outs[0][1][-800:]


def _cli_case(tmp_path, nreads=900):
    import json
    w = synth.dna_workload(30, 400, nreads, 100, (121, 122, 123))
    tre, aln, qf = tmp_path / "r.tre", tmp_path / "r.fasta", tmp_path / "q.fasta"
    tre.write_text(w["newick"] + "\n")
    with open(aln, "w") as f:
        for l, s in zip(w["labels"], w["seqs"]):
            f.write(">%s\n%s\n" % (l, s))
    with open(qf, "w") as f:
        for i, s in enumerate(w["reads"]):
            f.write(">q%d\n%s\n" % (i, s))
    model = "GTR{%s}+FU{%s}+G4{0.478218}" % ("/".join(map(repr, w["subst"])), "/".join(map(repr, w["freqs"])))
    base = [hostlib.cli_exe(), "-t", str(tre), "-s", str(aln), "-q", str(qf), "-m", model, "--chunk-size", "100"]

    def load(od):
        jp = json.load(open(od / "epa_result.jplace"))
        jp.pop("metadata", None)
        return jp
    return base, load


@pytest.mark.parametrize("rows_per_read,self_send", [(8, False), (1, True)])
def test_cli_rank_mode_one_rank_same_jplace(tmp_path, rows_per_read, self_send):
    """epa-ng-amd --rank 0 --world 1 --comm-file F: the one-process-per-GPU chunk loop (place_ranks.cpp) with a
    1-rank communicator -- results stay in HBM, every chunk's rows go through epa_dev_gather_slot /
    epa_comm_collect (rows_per_read = 1: smaller than a chunk's candidate count, so the carry path and the
    flush rounds run; self_send: rank 0's rows travel through ncclSend / ncclRecv) -- writes the jplace of
    the threaded chunk loop."""
    base, load = _cli_case(tmp_path)
    ref_dir = tmp_path / "out_threads"
    ref_dir.mkdir()
    r = subprocess.run(base + ["-w", str(ref_dir)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    od = tmp_path / "out_rank"
    od.mkdir()
    extra = ["--comm-rows-per-read", str(rows_per_read)] + (["--comm-self-send"] if self_send else [])
    r = subprocess.run(base + ["-w", str(od), "--rank", "0", "--world", "1", "--comm-file", str(tmp_path / "uid")] + extra,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    a, b = load(ref_dir), load(od)
    assert len(b["placements"]) == 900
    assert a == b


def test_cli_one_process_per_gpu_same_jplace(tmp_path):
    """one epa-ng-amd process per visible GPU (>= 2), RCCL gather to rank 0: the jplace of the 1-process run
    (pquery order: by rank slice = file order)."""
    n = min(8, epa.device_count())
    if n < 2:
        pytest.skip("needs >= 2 GPUs (RCCL wants one device per rank)")
    base, load = _cli_case(tmp_path)
    ref_dir = tmp_path / "out_threads"
    ref_dir.mkdir()
    r = subprocess.run(base + ["-w", str(ref_dir)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    od = tmp_path / "out_ranks"
    od.mkdir()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen(base + ["-w", str(od), "--rank", str(k), "--world", str(n), "--device", str(k),
                                      "--comm-file", str(tmp_path / "uid")], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for k in range(n)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    a, b = load(ref_dir), load(od)
    key = lambda p: p["n"][0]
    assert sorted(a["placements"], key=key) == sorted(b["placements"], key=key)
    assert a["tree"] == b["tree"]


@pytest.mark.parametrize("world,rows_per_read,nreads", [(2, 8, 900), (3, 1, 900), (8, 1, 850), (8, 8, 5)])
def test_cli_n_processes_on_one_gpu_same_jplace(tmp_path, world, rows_per_read, nreads):
    """The 1-GPU twin of test_cli_one_process_per_gpu_same_jplace: `world` epa-ng-amd processes on device 0
    (host/place_ranks.cpp), the gather over the transport stand-in.  rows_per_read = 1 makes rows_cap smaller
    than a chunk's candidates (carry + several flush calls); 850 reads over 8 ranks in 100-read chunks leaves a
    short last slice; 5 reads over 8 ranks leaves ranks 5 .. 7 EMPTY (every post of theirs is an empty one).
    Same jplace as the threaded chunk loop."""
    import fake_rccl_util
    base, load = _cli_case(tmp_path, nreads=nreads)
    ref_dir = tmp_path / "out_threads"
    ref_dir.mkdir()
    r = subprocess.run(base + ["-w", str(ref_dir)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    od = tmp_path / "out_ranks"
    od.mkdir()
    env = fake_rccl_util.env(dict(os.environ))
    idf = tmp_path / "uid"
    idf.write_bytes(b"\0" * 152)   # a stale record of an "earlier run": must be ignored, not hung on
    procs, outs = _run_ranks([(base + ["-w", str(od), "--rank", str(k), "--world", str(world), "--device", "0",
                                       "--comm-file", str(idf), "--comm-rows-per-read", str(rows_per_read)], env)
                              for k in range(world)])
    assert all(p.returncode == 0 for p in procs), outs
    a, b = load(ref_dir), load(od)
    key = lambda p: p["n"][0]
    assert len(b["placements"]) == nreads
    assert sorted(a["placements"], key=key) == sorted(b["placements"], key=key)
    assert a["tree"] == b["tree"]
    assert not idf.exists()        # rank 0 removes the id file once the communicator exists


@pytest.mark.parametrize("world,standin", [(1, False), (3, True), (8, True)])
def test_cli_rank_mode_no_heur_same_jplace(tmp_path, world, standin):
    """--no-heur (every branch placed thoroughly, LWR over ALL branches + filter on the device) in the one-process-per-GPU
    mode: the kept placements travel as gather rows together with their like-weight ratios (epa_dev_place_all_rows /
    epa_dev_gather_rows; EPA_ROW_LWR rows) -- rows_per_read = 1 forces the carry path, so a placement and its LWR row
    arrive in different gathers.  Same jplace as the threaded chunk loop's --no-heur.  world = 1: real RCCL with rank 0
    sending to itself; world = 3 / 8: the transport stand-in (8 ranks over 60 reads in 10-read chunks: short and empty
    trailing slices)."""
    import fake_rccl_util
    base, load = _cli_case(tmp_path, nreads=60)
    base = [x if x != "100" else "10" for x in base] + ["--no-heur"]        # --chunk-size 10
    ref_dir = tmp_path / "out_threads"
    ref_dir.mkdir()
    r = subprocess.run(base + ["-w", str(ref_dir)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    od = tmp_path / "out_ranks"
    od.mkdir()
    env = dict(os.environ)
    extra = ["--comm-rows-per-read", "1"]
    if standin:
        env = fake_rccl_util.env(env)
    else:
        extra.append("--comm-self-send")
    procs, outs = _run_ranks([(base + ["-w", str(od), "--rank", str(k), "--world", str(world), "--device", "0",
                                       "--comm-file", str(tmp_path / "uid")] + extra, env) for k in range(world)])
    assert all(p.returncode == 0 for p in procs), outs
    a, b = load(ref_dir), load(od)
    key = lambda p: p["n"][0]
    assert len(b["placements"]) == 60
    assert sorted(a["placements"], key=key) == sorted(b["placements"], key=key)
