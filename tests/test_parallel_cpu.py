"""N>1 path on CPU: world_size-2 gloo processes shard the queries like the reference's
local_seq_package and gather per-pair results to rank 0 (the GPU run uses RCCL for the same
calls).  No GPU compute: results are synthetic functions of (branch, global seq id)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["EPA_ROOT"])
import torch.distributed as dist
import epa_ng_amd as epa
from epa_ng_amd import parallel
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N = 1001
off, cnt = parallel.local_seq_package(N, rank, world)
rng = np.random.RandomState(5)
nb = rng.randint(1, 4, N)                      # candidates per (global) query
pairs, res = [], []
for q in range(off, off + cnt):
    for j in range(nb[q]):
        pairs.append((7 * q % 13 + j, q - off))
pairs = np.array(pairs, dtype=epa.PAIR_DTYPE)
res = np.zeros(len(pairs), epa.RESULT_DTYPE)
res["lnl"] = -(pairs["branch_id"] * 1000.0 + pairs["seq_id"] + off)
out = parallel.gather_results(pairs, res, off, dist)
if rank == 0:
    assert out.shape == (nb.sum(), 5), out.shape
    assert np.array_equal(np.unique(out[:, 1]), np.arange(N))
    assert np.allclose(out[:, 2], -(out[:, 0] * 1000.0 + out[:, 1]))
    print("GATHER_OK", out.shape[0])
else:
    assert out is None
# overlapped exchange used by bench.py: three chunks through two slots
import torch
ag = parallel.AsyncResultGather(dist, rows_cap=64, device=torch.device("cpu"))
for step in range(3):
    n = 5 + 3 * rank + step
    p = torch.zeros((64, 2), dtype=torch.int32)
    r = torch.zeros((64, 3), dtype=torch.float64)
    p[:n, 0] = torch.arange(n, dtype=torch.int32) + 100 * step      # branch_id
    p[:n, 1] = torch.arange(n, dtype=torch.int32) + 1000 * rank     # seq_id
    r[:n, 0] = -(p[:n, 0].double() * 7 + p[:n, 1].double())
    ag.post(p, r, n, keep=True)
ag.finish(keep=True)
if rank == 0:
    assert len(ag.collected) == 3
    for step, parts in enumerate(ag.collected):
        assert len(parts) == world
        for rk, rows in enumerate(parts):
            assert rows.shape == (5 + 3 * rk + step, 4)
            b, s_, lnl, _, _ = parallel.unpack_rows(rows)
            assert np.array_equal(b, np.arange(len(b)) + 100 * step)
            assert np.array_equal(s_, np.arange(len(b)) + 1000 * rk)
            assert np.array_equal(lnl, -(b * 7.0 + s_))
    assert ag.carried_rows == 0
    print("ASYNC_OK")
# a rank whose chunk exceeds rows_cap carries the rest into its next gathers; finish() drains it
ag = parallel.AsyncResultGather(dist, rows_cap=8, device=torch.device("cpu"))
sent = []
for step in range(3):
    n = (21 if rank == 1 and step == 1 else 3) + step
    p = torch.zeros((32, 2), dtype=torch.int32)
    r = torch.zeros((32, 3), dtype=torch.float64)
    p[:n, 0] = torch.arange(n, dtype=torch.int32) + 100 * step
    p[:n, 1] = 1000 * rank
    r[:n, 0] = torch.arange(n, dtype=torch.float64) + 0.5 * step
    ag.post(p, r, n, keep=True)
    sent.append(n)
ag.finish(keep=True)
if rank == 0:
    for rk in range(world):
        rows = np.concatenate([parts[rk] for parts in ag.collected], 0)
        want = [(21 if rk == 1 and st == 1 else 3) + st for st in range(3)]
        assert rows.shape[0] == sum(want), (rows.shape, want)
        b, s_, lnl, _, _ = parallel.unpack_rows(rows)
        assert np.array_equal(b, np.concatenate([np.arange(n) + 100 * st for st, n in enumerate(want)]))   # order kept
        assert np.all(s_ == 1000 * rk)
    assert len(ag.collected) > 3                  # the drain rounds of finish()
    print("CARRY_OK")
elif rank == 1:
    assert ag.carried_rows > 0
dist.destroy_process_group()
'''


def test_local_seq_package_matches_reference_formula():
    from epa_ng_amd import parallel
    for n in (0, 1, 7, 8, 9, 1000, 1001):
        for world in (1, 2, 3, 8):
            slices = [parallel.local_seq_package(n, r, world) for r in range(world)]
            assert sum(c for _, c in slices) == n
            pos = 0
            for off, cnt in slices:
                assert off == min(pos, n)
                pos += cnt
            part = -(-n // world) if n else 0
            assert all(c <= part for _, c in slices)


def test_cpp_local_seq_package_equals_the_harness_formula():
    """the slice the one-process-per-GPU chunk loop of the product takes (epa_ng_amd/csrc/host/place_ranks.cpp) is the
    one the Python harness takes: both restate src/net/epa_mpi_util.cpp:10-30"""
    import ctypes as C
    from epa_ng_amd import hostlib, parallel
    L = hostlib.host_lib()
    L.epa_host_local_seq_package.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.epa_host_local_seq_package.restype = None
    for n in (0, 1, 7, 8, 9, 100, 1000003, 10 ** 7):
        for world in (1, 2, 3, 8):
            tot = 0
            for r in range(world):
                o, c = C.c_uint64(), C.c_uint64()
                L.epa_host_local_seq_package(n, r, world, C.byref(o), C.byref(c))
                assert (o.value, c.value) == parallel.local_seq_package(n, r, world)
                tot += c.value
            assert tot == n


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, EPA_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0] and "ASYNC_OK" in outs[0] and "CARRY_OK" in outs[0]


def test_transport_standin_builds_and_exports_what_the_product_binds():
    """tests/fake_rccl.cpp (test infrastructure for the N > 1 protocol tests on a 1-GPU box) exports exactly the
    ten entry points csrc/comm.hip resolves with dlsym"""
    import re
    import subprocess
    import fake_rccl_util
    so = fake_rccl_util.build()
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    have = set(re.findall(r" T (nccl\w+)", out))
    assert have == set(fake_rccl_util.SYMBOLS)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "epa_ng_amd", "csrc", "comm.hip")).read()
    assert set(re.findall(r'SYM\(\w+, "(nccl\w+)"\)', src)) == have
