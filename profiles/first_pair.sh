#!/bin/bash
# Why does every wave's FIRST pair of a Newton launch take ~92 us instead of ~41?  
# Builds thorough_dna.hip with -DTH_TIMING on the box (in-kernel s_memrealtime stamps per wave) and compares
#   A  a launch that follows other kernels (the chunk pipeline: preplacement + selection ran in between)
#   B  the same launch repeated back to back (nothing but a memset and the result copy in between)
#   C  after a deliberate L2 / I-cache disturbance: a 1 GB device memset between two launches
set -e
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I epa_ng_amd/csrc -Wno-unused-result -mllvm -amdgpu-sched-strategy=iterative-ilp -DTH_TIMING -c epa_ng_amd/csrc/thorough_dna.hip -o epa_ng_amd/build/thorough_dna.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o epa_ng_amd/libepa_dev.so epa_ng_amd/build/*.o
python - <<'PY' 2>&1 | grep -v "^  x " 
import sys, numpy as np, torch
sys.path.insert(0, "tests")
import epa_ng_amd as epa
from epa_ng_amd import hostlib, synth
root = synth.random_tree(512, 1)
rates = synth.gamma_rates(synth.CFG2_ALPHA)
labels, seqs = synth.simulate_msa(root, 1500, synth.CFG2_SUBST, synth.CFG2_FREQS, rates, 2)
ref = hostlib.Reference(synth.newick(root), labels, seqs, states=4, subst=synth.CFG2_SUBST, freqs=synth.CFG2_FREQS, rates=rates)
ev = ref.evaluator()
codes, wb, ws = synth.make_reads_compact(seqs, 100000, 150, 0.03, 3, 4)
dev = torch.device("cuda", 0)
dc, dwb, dws = torch.from_numpy(codes).to(dev), torch.from_numpy(wb.view(np.int32)).to(dev), torch.from_numpy(ws.view(np.int32)).to(dev)
cap = 100000 * 64
dp = torch.empty((cap, 2), dtype=torch.int32, device=dev); dr = torch.zeros((cap, 3), dtype=torch.float64, device=dev)
for i in range(4):
    print("== A%d: chunk body (preplacement + selection ran just before this Newton launch)" % i, file=sys.stderr, flush=True)
    ev.chunk_stage(0, dc, dwb, dws); ev.chunk_launch(0, max_span=150, max_pairs=cap, pairs_out=dp, results_out=dr, keep_on_device=True)
    n = ev.chunk_finish_device(0)
pairs = np.zeros(n, epa.PAIR_DTYPE); hp = dp[:n].cpu().numpy(); pairs["branch_id"], pairs["seq_id"] = hp[:, 0], hp[:, 1]
for i in range(4):
    print("== B%d: thorough alone, back to back" % i, file=sys.stderr, flush=True)
    ev.thorough(pairs, codes, wb, ws)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
for i in range(3):
    big.fill_(i); torch.cuda.synchronize()
    print("== C%d: thorough after a 1 GB fill (L2 contents replaced)" % i, file=sys.stderr, flush=True)
    ev.thorough(pairs, codes, wb, ws)
small = pairs[:13000]
for i in range(3):
    print("== D%d: 13k pairs back to back" % i, file=sys.stderr, flush=True)
    ev.thorough(small, codes, wb, ws)
PY
