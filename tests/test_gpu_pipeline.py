"""GPU tests of the whole path fed by the product's own host code (no oracle inputs): C++ host
precompute -> C-ABI -> HIP kernels, and the chunk loop down to the jplace file.  The oracle and
the golden vectors are the checkers."""
import json
import os

import numpy as np
import pytest

import epa_ng_amd as epa
from epa_ng_amd import hostlib, synth
from golden_util import GOLDEN, load_case
from oracle_lib import Oracle

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_host_fed_device_matches_oracle_512tips():
    # BASELINE configs[1] shape at reduced read count: 512 tips, W=1500, 150 bp reads
    w = synth.dna_workload(512, 1500, 1200, 150, (1, 2, 3))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    o = Oracle(w["newick"], w["labels"], w["seqs"], 4, w["subst"], w["freqs"], w["rates"])
    codes, wb, ws = epa.encode_queries(4, w["reads"])
    lnl = ev.preplace(codes, wb, ws)
    assert np.max(np.abs(lnl - o.preplace(w["reads"]))) < 1e-6
    pairs = ev.select(lnl, len(w["reads"]), 0.99999)
    hb, hs = hostlib.heuristic(lnl, "dynamic", 0.99999)   # device heuristic == host heuristic
    assert sorted(zip(hb.tolist(), hs.tolist())) == sorted(zip(pairs["branch_id"].tolist(), pairs["seq_id"].tolist()))
    res = ev.thorough(pairs, codes, wb, ws)
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], w["reads"])
    assert np.max(np.abs(res["lnl"] - tl)) < 1e-6
    assert np.max(np.abs(res["pendant_length"] - tp) / np.maximum(1.0, tp)) < 1e-6
    assert np.max(np.abs(res["distal_length"] - td)) < 1e-6
    assert ev.last_stats["rounds"] == o.last_stats["rounds"]
    # size-independent properties: best preplacement branch of an unmutated read is adjacent to
    # (or is) its source tip's branch is not guaranteed; but lnL must be finite and LWRs sum to 1
    assert np.all(np.isfinite(lnl))


def test_device_resident_buffers_roundtrip():
    import torch
    w = synth.dna_workload(32, 400, 300, 100, (51, 52, 53))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    codes, wb, ws = epa.encode_queries(4, w["reads"])
    Q = len(w["reads"])
    dev = torch.device("cuda", 0)
    dc = torch.from_numpy(codes).to(dev)
    dwb = torch.from_numpy(wb.view(np.int32)).to(dev)
    dws = torch.from_numpy(ws.view(np.int32)).to(dev)
    d_lnl = torch.empty((Q, ref.B), dtype=torch.float64, device=dev)
    ev.preplace(dc, dwb, dws, Q=Q, out=d_lnl)
    torch.cuda.synchronize()
    host = ev.preplace(codes, wb, ws)
    assert np.array_equal(d_lnl.cpu().numpy(), host)      # same kernel, same bits
    d_pairs = torch.empty((Q * 32, 2), dtype=torch.int32, device=dev)
    n = ev.select(d_lnl, Q, 0.99999, max_pairs=Q * 32, out=d_pairs)
    d_res = torch.empty((n, 3), dtype=torch.float64, device=dev)
    ev.thorough(d_pairs, dc, dwb, dws, Q=Q, n_pairs=n, out=d_res)
    torch.cuda.synchronize()
    pairs = np.zeros(n, epa.PAIR_DTYPE)
    pairs["branch_id"] = d_pairs[:n, 0].cpu().numpy().view(np.uint32)
    pairs["seq_id"] = d_pairs[:n, 1].cpu().numpy().view(np.uint32)
    res = ev.thorough(pairs, codes, wb, ws)
    assert np.array_equal(d_res.cpu().numpy()[:, 0], res["lnl"])


def test_chunk_loop_to_jplace(tmp_path):
    # bundled-shape plumbing case (BASELINE configs[0]) through the GPU path
    g = load_case("dna8_gtr_g_default")
    labels = [a for a, _ in g["msa"]]
    seqs = [b for _, b in g["msa"]]
    ref = hostlib.Reference(g["newick"], labels, seqs, model="GTR+G")
    qf = tmp_path / "q.fasta"
    with open(qf, "w") as f:
        for q in g["queries"]:
            f.write(">%s\n%s\n" % (q["name"], q["seq"]))
    nq, npairs = ref.place_file(str(qf), str(tmp_path), chunk_size=4)
    assert nq == len(g["queries"])
    jp = json.load(open(tmp_path / "epa_result.jplace"))
    assert jp["version"] == 3
    assert jp["fields"] == ["edge_num", "likelihood", "like_weight_ratio", "distal_length", "pendant_length"]
    assert jp["tree"].count("{") == 13
    assert [p["n"][0] for p in jp["placements"]] == [q["name"] for q in g["queries"]]
    for qi, p in enumerate(jp["placements"]):
        lw = [x[2] for x in p["p"]]
        assert lw == sorted(lw, reverse=True) and 1 <= len(lw) <= 7
        for edge, lnl, lwr, distal, pendant in p["p"]:
            gold = g["thorough"][qi][edge]
            assert abs(lnl - gold["lnl"]) < 1e-6
            assert abs(distal - gold["distal"]) < 1e-6 and abs(pendant - gold["pendant"]) < 1e-5 * max(1, gold["pendant"])
        # best edge: Rat -> 4, Carp -> 3
    assert jp["placements"][0]["p"][0][0] == 4 and jp["placements"][1]["p"][0][0] == 3


def test_fused_place_chunk_equals_three_calls():
    w = synth.dna_workload(96, 700, 900, 150, (61, 62, 63))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    codes, wb, ws = epa.encode_queries(4, w["reads"])
    lnl = ev.preplace(codes, wb, ws)
    pairs = ev.select(lnl, len(w["reads"]), 0.99999)
    res = ev.thorough(pairs, codes, wb, ws)
    p2, r2 = ev.place_chunk(codes, wb, ws)
    assert np.array_equal(p2, pairs)
    assert np.array_equal(r2["lnl"], res["lnl"]) and np.array_equal(r2["distal_length"], res["distal_length"])
    # mixed window lengths incl. long ones, max_span left to the library
    base = w["seqs"][5]
    W = len(base)
    qs = [base, "-" * 10 + base[10:400] + "-" * (W - 400), "-" * 300 + base[300:301] + "-" * (W - 301)] + w["reads"][:50]
    c2, b2, s2 = epa.encode_queries(4, qs)
    p3, r3 = ev.place_chunk(c2, b2, s2)
    o = Oracle(w["newick"], w["labels"], w["seqs"], 4, w["subst"], w["freqs"], w["rates"])
    tl, tp, td = o.thorough(p3["branch_id"], p3["seq_id"], qs)
    assert np.max(np.abs(r3["lnl"] - tl)) < 1e-6


def test_cli_binary_matches_golden(tmp_path):
    # the epa-ng-amd executable with the reference's flag names (src/main.cpp:96-270)
    import subprocess
    g = load_case("dna8_gtr_fu_g4")
    data = os.path.join(GOLDEN, "data")
    qf = tmp_path / "q.fasta"
    with open(qf, "w") as f:
        for q in g["queries"]:
            f.write(">%s\n%s\n" % (q["name"], q["seq"]))
    exe = hostlib.cli_exe()
    model = ("GTR{0.787874/1.821672/1.294006/0.698421/3.034135/1.0}+FU{0.256465/0.222535/0.308594/"
             "0.212406}+G4{0.478218}")
    r = subprocess.run([exe, "-t", os.path.join(data, "ref.tre"), "-s", os.path.join(data, "aln.fasta"),
                        "-q", str(qf), "-m", model, "-w", str(tmp_path), "--filter-max", "13",
                        "--filter-min-lwr", "0.0000001"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    jp = json.load(open(tmp_path / "epa_result.jplace"))
    assert len(jp["placements"]) == len(g["queries"])
    checked = 0
    for qi, p in enumerate(jp["placements"]):
        assert abs(sum(x[2] for x in p["p"]) - 1.0) < 1e-3 or len(p["p"]) < 13
        for edge, lnl, lwr, distal, pendant in p["p"]:
            gold = g["thorough"][qi][edge]
            assert abs(lnl - gold["lnl"]) < 1e-6
            checked += 1
    assert checked >= len(g["queries"])
    # --no-heur: every branch gets a thorough placement
    r = subprocess.run([exe, "-t", os.path.join(data, "ref.tre"), "-s", os.path.join(data, "aln.fasta"),
                        "-q", str(qf), "-m", model, "-w", str(tmp_path), "--no-heur", "--filter-max", "13",
                        "--filter-min-lwr", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    jp = json.load(open(tmp_path / "epa_result.jplace"))
    for qi, p in enumerate(jp["placements"]):
        assert len(p["p"]) <= 13
        for edge, lnl, lwr, distal, pendant in p["p"]:
            assert abs(lnl - g["thorough"][qi][edge]["lnl"]) < 1e-6


def test_cli_premasking_removes_all_gap_columns(tmp_path):
    """or_mask premasking of the reference front end (src/main.cpp:470-494, src/seq/MSA_Info.hpp:75-82,
    src/seq/MSA_Stream.cpp:26): a column that is all-gap in the reference file, or in the whole query
    file, is cut out of both alignments.  Files with such columns inserted must give the placements
    of the files without them (same edges, lnL, lengths); --no-pre-mask keeps the columns, and the
    reference-gap column (where the queries do carry characters) then shifts every lnL."""
    import subprocess
    g = load_case("dna8_gtr_fu_g4")
    data = os.path.join(GOLDEN, "data")
    ref = []
    name = None
    for line in open(os.path.join(data, "aln.fasta")):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:]
            ref.append([name, ""])
        elif line:
            ref[-1][1] += line
    rng = np.random.RandomState(5)

    def insert(seq, cols):   # cols: ascending list of (position in the ORIGINAL numbering, char)
        out, last = [], 0
        for pos, ch in cols:
            out.append(seq[last:pos])
            out.append(ch)
            last = pos
        out.append(seq[last:])
        return "".join(out)

    pos_refgap, pos_qrygap = 7, 123
    def write(dirp, padded):
        os.makedirs(dirp, exist_ok=True)
        with open(os.path.join(dirp, "r.fasta"), "w") as f:
            for n, s in ref:
                if padded:   # column A: gap in every reference row; column B: a real character
                    s = insert(s, [(pos_refgap, "-"), (pos_qrygap, "ACGT"[rng.randint(4)])])
                f.write(">%s\n%s\n" % (n, s))
        with open(os.path.join(dirp, "q.fasta"), "w") as f:
            for q in g["queries"]:
                s = q["seq"]
                if padded:   # column A: the queries carry a character; column B: gap in every query
                    s = insert(s, [(pos_refgap, "A"), (pos_qrygap, "-")])
                f.write(">%s\n%s\n" % (q["name"], s))

    exe = hostlib.cli_exe()
    model = ("GTR{0.787874/1.821672/1.294006/0.698421/3.034135/1.0}+FU{0.256465/0.222535/0.308594/"
             "0.212406}+G4{0.478218}")

    def run(dirp, *extra):
        r = subprocess.run([exe, "-t", os.path.join(data, "ref.tre"), "-s", os.path.join(dirp, "r.fasta"),
                            "-q", os.path.join(dirp, "q.fasta"), "-m", model, "-w", dirp, "--filter-max", "13",
                            "--filter-min-lwr", "0"] + list(extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        return json.load(open(os.path.join(dirp, "epa_result.jplace")))["placements"], r.stdout

    base, _ = run(str(tmp_path / "base"), *(write(str(tmp_path / "base"), False) or ()))
    padded, out = run(str(tmp_path / "pad"), *(write(str(tmp_path / "pad"), True) or ()))
    assert "Premasking: " in out
    assert len(base) == len(padded) == len(g["queries"])
    for a, b in zip(base, padded):
        assert a["n"] == b["n"] and len(a["p"]) == len(b["p"])
        for x, y in zip(a["p"], b["p"]):
            assert x[0] == y[0] and abs(x[1] - y[1]) < 1e-9 and abs(x[3] - y[3]) < 1e-9 and abs(x[4] - y[4]) < 1e-9
    nomask, _ = run(str(tmp_path / "pad"), "--no-pre-mask")
    d = [x[1] - y[1] for a, b in zip(base, nomask) for x, y in zip(a["p"], b["p"]) if x[0] == y[0]]
    assert d and min(abs(v) for v in d) > 0.1     # log(pi_A) and the extra reference column


def test_edge_cases_single_query_unsorted_pairs_many_branches():
    # B > 1024 exercises the 32-values-per-lane selection kernel; Q = 1; pairs in arbitrary order
    w = synth.dna_workload(600, 200, 3, 120, (81, 82, 83))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    assert ref.B == 1197
    ev = ref.evaluator()
    o = Oracle(w["newick"], w["labels"], w["seqs"], 4, w["subst"], w["freqs"], w["rates"])
    codes, wb, ws = epa.encode_queries(4, w["reads"][:1])
    lnl = ev.preplace(codes, wb, ws)
    assert lnl.shape == (1, ref.B)
    assert np.max(np.abs(lnl - o.preplace(w["reads"][:1]))) < 1e-6
    pairs, res = ev.place_chunk(codes, wb, ws)
    hb, hs = hostlib.heuristic(lnl, "dynamic", 0.99999)
    assert sorted(pairs["branch_id"].tolist()) == sorted(hb.tolist())
    rng = np.random.RandomState(1)
    codes3, wb3, ws3 = epa.encode_queries(4, w["reads"])
    pr = np.zeros(40, epa.PAIR_DTYPE)
    pr["branch_id"] = rng.randint(0, ref.B, 40)
    pr["seq_id"] = rng.randint(0, 3, 40)
    r2 = ev.thorough(pr, codes3, wb3, ws3)
    tl, tp, td = o.thorough(pr["branch_id"], pr["seq_id"], w["reads"])
    assert np.max(np.abs(r2["lnl"] - tl)) < 1e-6
    # threshold 1.0 selects every branch with non-zero LWR mass until the sum reaches 1
    p_all = ev.select(lnl, 1, 2.0)
    assert len(p_all) == ref.B and len(set(p_all["branch_id"].tolist())) == ref.B


def test_full_size_cfg2_properties():
    """BASELINE configs[1] at full size (512 tips, W=1500, 100k reads): too big for the oracle, so
    size-independent properties are checked: thorough lnL never below the preplacement lnL of
    the same pair (the optimiser starts from the preplacement lengths and only accepts
    improvements or reverts), sanity ranges, candidates == host heuristic on a sample, LWR
    normalisation, bit-identical repeat."""
    w = synth.dna_workload(512, 1500, 100000, 150, (1, 2, 3))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    codes, wb, ws = epa.encode_queries(4, w["reads"])
    Q = len(w["reads"])
    pairs, res = ev.place_chunk(codes, wb, ws, max_span=150)
    assert len(pairs) >= Q and set(np.unique(pairs["seq_id"])) == set(range(Q))
    assert np.all(np.isfinite(res["lnl"]))
    assert np.all(np.diff(pairs["branch_id"].astype(np.int64)) >= 0)
    bl = np.array([ref.branch(int(b))["length"] for b in range(ref.B)])
    assert np.all(res["pendant_length"] >= 1e-4 - 1e-12) and np.all(res["pendant_length"] <= 100.0)
    assert np.all(res["distal_length"] > 0) and np.all(res["distal_length"] < bl[pairs["branch_id"]])
    sub = slice(0, 4000)
    lnl = ev.preplace(codes[sub], wb[sub], ws[sub])
    m = pairs["seq_id"] < 4000
    pre = lnl[pairs["seq_id"][m], pairs["branch_id"][m]]
    assert np.all(res["lnl"][m] >= pre - 1e-7)
    hb, hs = hostlib.heuristic(lnl, "dynamic", 0.99999)
    assert sorted(zip(hb.tolist(), hs.tolist())) == sorted(zip(pairs["branch_id"][m].tolist(), pairs["seq_id"][m].tolist()))
    lw = np.exp(lnl - lnl.max(1, keepdims=True))
    lw /= lw.sum(1, keepdims=True)
    assert np.allclose(lw.sum(1), 1.0)
    # repeats: every launch of this size moves the XCDs' shares of the pair list toward the speeds the previous one
    # measured (epa_xcd_feedback) -- which wave places a pair must not show in any bit of the results
    shares, th_ms = [ev.xcd_shares()], []
    for _ in range(3):
        p2, r2 = ev.place_chunk(codes, wb, ws, max_span=150)
        assert np.array_equal(p2, pairs) and np.array_equal(r2, res)
        shares.append(ev.xcd_shares())
        th_ms.append(ev.kernel_ms("thorough"))
    # the invariants, whatever the box's speed: the shares sum to 1 and stay within the clamp
    for sh in shares:
        assert abs(sh.sum() - 1.0) < 1e-5 and np.all(sh > 0.125 * 0.84) and np.all(sh < 0.125 * 1.16)
    # learning is gated on the launch's MEASURED duration (>= 1 ms), not on what this box is expected to take
    if min(th_ms) >= 1.5:
        assert not np.array_equal(shares[-1], np.full(8, 0.125))
    assert 500.0 < ev.sclk_mhz() < 3000.0                               # the launch stamped its shader clock
    small = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"], freqs=w["freqs"],
                              rates=w["rates"]).evaluator()
    small.place_chunk(codes[:500], wb[:500], ws[:500], max_span=150)
    if small.kernel_ms("thorough") < 0.7:
        assert np.array_equal(small.xcd_shares(), np.full(8, 0.125))   # a launch below 1 ms teaches nothing


@pytest.mark.parametrize("states", [4, 20])
def test_compact_query_layout_gives_identical_results(states):
    """q_codes as compact window rows (epa_dev_set_query_layout) vs aligned Q x W rows: the three
    device calls return bit-identical tables / pairs / results (host and device buffers)."""
    import torch
    from epa_ng_amd import synth
    if states == 4:
        w = synth.dna_workload(32, 500, 700, 120, (41, 42, 43))
    else:
        w = synth.aa_workload(16, 260, 150, 70, (44, 45, 46))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=states, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    reads = list(w["reads"])
    reads[3] = reads[3].replace("A", "N", 2)
    full = epa.encode_queries(states, reads)
    comp = epa.encode_queries(states, reads, compact=True)
    assert comp[0].shape[1] < full[0].shape[1]
    t_full = ev.preplace(*full)
    t_comp = ev.preplace(*comp)
    assert np.array_equal(t_full, t_comp)
    pf, rf = ev.place_chunk(*full)
    pc, rc = ev.place_chunk(*comp)
    assert np.array_equal(pf, pc)
    for k in ("lnl", "pendant_length", "distal_length"):
        assert np.array_equal(rf[k], rc[k])
    r2 = ev.thorough(pc, *comp)
    assert np.array_equal(r2["lnl"], rc["lnl"])
    # device-resident compact buffers
    dc = torch.from_numpy(comp[0]).cuda()
    db = torch.from_numpy(comp[1].view(np.int32)).cuda()
    ds = torch.from_numpy(comp[2].view(np.int32)).cuda()
    t_dev = ev.preplace(dc, db, ds)
    assert np.array_equal(t_dev, t_full)


@pytest.mark.parametrize("states", [4, 20])
def test_mixed_window_lengths_dispatch_by_span_class(states):
    """One chunk whose windows span several kernel classes (DNA: 1..24 sites per lane and the
    HBM-slab kernel; 20 states: LDS and HBM slab): pairs are partitioned by class on the device,
    results come back in pair order and equal the oracle's."""
    from epa_ng_amd import synth
    rng = np.random.RandomState(9)
    if states == 4:
        W, lens = 1900, [20, 70, 130, 200, 300, 420, 700, 900, 1300, 1700]
        base = synth.dna_workload(10, W, 2, 100, (61, 62, 63))
    else:
        W, lens = 330, [30, 64, 100, 102, 103, 128, 200, 300]
        base = synth.aa_workload(10, W, 2, 50, (64, 65, 66))
    reads = []
    for i, n in enumerate(lens * 2):
        r, _ = synth.make_reads(base["seqs"], 1, n, 0.05, 700 + i, states=states)
        reads.append(r[0])
    o = Oracle(base["newick"], base["labels"], base["seqs"], states, base["subst"], base["freqs"], base["rates"])
    ref = hostlib.Reference(base["newick"], base["labels"], base["seqs"], states=states, subst=base["subst"],
                            freqs=base["freqs"], rates=base["rates"])
    ev = ref.evaluator()
    codes, wb, ws = epa.encode_queries(states, reads, compact=True)
    pairs, res = ev.place_chunk(codes, wb, ws)                      # class histogram from the select
    assert np.all(np.diff(pairs["branch_id"].astype(np.int64)) >= 0)
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], reads)
    assert np.max(np.abs(res["lnl"] - tl)) < 1e-6
    assert np.max(np.abs(res["distal_length"] - td)) < 1e-6
    rounds = o.last_stats["rounds"]
    assert ev.last_stats["rounds"] == rounds
    perm = rng.permutation(len(pairs))                              # caller-supplied pairs, any order
    r2 = ev.thorough(pairs[perm], codes, wb, ws)
    assert np.array_equal(r2["lnl"], res["lnl"][perm])
    assert ev.last_stats["rounds"] == rounds


def test_large_reference_more_than_4096_branches():
    """B = 4397 (> 64 x 64): candidate selection streams the row instead of holding it in
    registers (k_select_big); fused chunk vs the oracle and vs the host restatement."""
    from epa_ng_amd import synth
    w = synth.dna_workload(2200, 64, 48, 40, (71, 72, 73))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    assert ref.B == 2 * 2200 - 3
    ev = ref.evaluator()
    o = Oracle(w["newick"], w["labels"], w["seqs"], 4, w["subst"], w["freqs"], w["rates"])
    codes, wb, ws = epa.encode_queries(4, w["reads"], compact=True)
    lnl = ev.preplace(codes, wb, ws)
    assert np.max(np.abs(lnl - o.preplace(w["reads"]))) < 1e-6
    # 40-column reads against 4397 branches: flat likelihoods, > 100 candidates per read
    pairs, res = ev.place_chunk(codes, wb, ws, max_pairs=len(w["reads"]) * ref.B)
    hb, hs = hostlib.heuristic(lnl, "dynamic", 0.99999)
    assert sorted(zip(hb.tolist(), hs.tolist())) == sorted(zip(pairs["branch_id"].tolist(), pairs["seq_id"].tolist()))
    assert np.all(np.diff(pairs["branch_id"].astype(np.int64)) >= 0)
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], w["reads"])
    assert np.max(np.abs(res["lnl"] - tl)) < 1e-6


@pytest.mark.parametrize("states", [4, 20])
def test_device_reference_precompute_matches_host_clvs(states):
    """epa_dev_create_from_tree (all directional CLVs computed on the GPU from the tree and the tip
    sequences) vs epa_dev_create fed with the host-computed CLVs: tree lnL on every kind of
    branch, the preplacement table and the thorough results agree to rounding; golden tree lnL."""
    from epa_ng_amd import synth
    if states == 4:
        g = load_case("dna8_gtr_g_default")
        labels = [a for a, _ in g["msa"]]
        seqs = [b for _, b in g["msa"]]
        ref = hostlib.Reference(g["newick"], labels, seqs, states=4, subst=g["subst"], freqs=g["freqs"],
                                rates=g["gamma_rates"])
        ev = ref.evaluator()
        for b in range(ref.B):   # the reference's property: equal on every edge
            assert abs(ev.tree_logl(b) - (-4620.363834217626)) < 1e-7
        w = synth.dna_workload(200, 420, 300, 110, (81, 82, 83))   # deep enough to need rescaling? no, but many levels
    else:
        w = synth.aa_workload(40, 260, 120, 80, (84, 85, 86))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=states, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    e_dev = ref.evaluator(device_precompute=True)
    e_host = ref.evaluator(device_precompute=False)
    host_lnl = ref.tree_lnl(0)
    for b in (0, 1, ref.B // 2, ref.B - 1):
        assert abs(e_dev.tree_logl(b) - host_lnl) < 1e-6 * max(1.0, abs(host_lnl)) * 1e-3 + 1e-6
        assert abs(e_host.tree_logl(b) - host_lnl) < 1e-6 * max(1.0, abs(host_lnl)) * 1e-3 + 1e-6
    codes, wb, ws = epa.encode_queries(states, w["reads"], compact=True)
    t_dev, t_host = e_dev.preplace(codes, wb, ws), e_host.preplace(codes, wb, ws)
    assert np.max(np.abs(t_dev - t_host)) < 1e-8
    p_dev, r_dev = e_dev.place_chunk(codes, wb, ws)
    p_host, r_host = e_host.place_chunk(codes, wb, ws)
    assert np.array_equal(p_dev, p_host)
    assert np.max(np.abs(r_dev["lnl"] - r_host["lnl"])) < 1e-8
    assert np.max(np.abs(r_dev["distal_length"] - r_host["distal_length"])) < 1e-7


def test_device_reference_precompute_rescaling_deep_tree():
    """caterpillar-like 700-tip tree with long branches: the per-site 2^256 rescale triggers
    inside the device recursion; scaler counts must agree with the host precompute (tree lnL and
    preplacement equal)."""
    from epa_ng_amd import synth
    rng = np.random.RandomState(4)
    n = 700
    nwk = "(t0:0.9,t1:0.9"
    for i in range(2, n - 1):
        nwk = "(" + nwk + "):0.9,t%d:0.9" % i
    nwk = nwk + ",t%d:0.9);" % (n - 1)
    W = 96
    labels = ["t%d" % i for i in range(n)]
    seqs = ["".join(rng.choice(list("ACGT"), W)) for _ in range(n)]
    ref = hostlib.Reference(nwk, labels, seqs, states=4, subst=synth.CFG2_SUBST, freqs=synth.CFG2_FREQS,
                            rates=synth.gamma_rates(0.5))
    e_dev = ref.evaluator(device_precompute=True)
    e_host = ref.evaluator(device_precompute=False)
    host_lnl = ref.tree_lnl(0)
    assert host_lnl < -256 * np.log(2.0) * 10     # scaled many times over
    for b in (0, ref.B // 3, ref.B - 1):
        assert abs(e_dev.tree_logl(b) - host_lnl) < 1e-6
    reads = ["".join(rng.choice(list("ACGT"), W)) for _ in range(20)]
    codes, wb, ws = epa.encode_queries(4, reads, compact=True)
    assert np.max(np.abs(e_dev.preplace(codes, wb, ws) - e_host.preplace(codes, wb, ws))) < 1e-7


@pytest.mark.parametrize("states", [4, 20])
def test_prop_invariant_sites_device_vs_oracle(states):
    """+I through the whole device path (device reference precompute, lookup, pair / generic
    preplacement, first-phase vectors, thorough kernels incl. the long-window one): tree lnL,
    preplacement table and thorough results equal the oracle's +I restatement."""
    from epa_ng_amd import synth
    pinv = 0.23
    if states == 4:
        w = synth.dna_workload(24, 330, 60, 90, (91, 92, 93))
        const_cols = "ACGTTGCA" * 5
    else:
        w = synth.aa_workload(14, 230, 40, 60, (94, 95, 96))
        const_cols = "ARNDCQEGHI" * 3
    k = len(const_cols)
    seqs = [s[:-k] + const_cols for s in w["seqs"]]           # a block of invariant columns
    reads = [r[:-k] + (const_cols if i % 2 else "-" * k) for i, r in enumerate(w["reads"])]
    reads += [r[:60] + "-" * (len(r) - 60 - k) + const_cols for r in w["reads"][:6]]   # long windows
    if states == 4:
        reads[0] = reads[0].replace("A", "R", 1)                 # generic preplace kernel
    o = Oracle(w["newick"], w["labels"], seqs, states, w["subst"], w["freqs"], w["rates"], pinv=pinv)
    ref = hostlib.Reference(w["newick"], w["labels"], seqs, states=states, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"], pinv=pinv)
    for dev_pre in (True, False):
        ev = ref.evaluator(device_precompute=dev_pre)
        assert abs(ev.tree_logl(0) - o.tree_lnl(0)) < 1e-7
        codes, wb, ws = epa.encode_queries(states, reads, compact=True)
        lnl = ev.preplace(codes, wb, ws)
        assert np.max(np.abs(lnl - o.preplace(reads))) < 1e-6
        pairs, res = ev.place_chunk(codes, wb, ws, max_pairs=len(reads) * ref.B)
        tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], reads)
        assert np.max(np.abs(res["lnl"] - tl)) < 1e-6
        assert np.max(np.abs(res["distal_length"] - td)) < 1e-6
        assert np.max(np.abs(res["pendant_length"] - tp) / np.maximum(1.0, tp)) < 1e-6
        assert ev.last_stats["rounds"] == o.last_stats["rounds"]
    # and the +I term really matters on this data
    o0 = Oracle(w["newick"], w["labels"], seqs, states, w["subst"], w["freqs"], w["rates"])
    assert np.max(np.abs(o0.preplace(reads[:4]) - o.preplace(reads[:4]))) > 0.1


def test_prop_invariant_sites_long_window_kernel():
    """+I in k_thorough_dna_long (windows > 1536 sites)"""
    from epa_ng_amd import synth
    w = synth.dna_workload(10, 1800, 4, 1700, (97, 98, 99))
    seqs = [s[:-40] + "ACGT" * 10 for s in w["seqs"]]
    reads = [r[:-40] + "ACGT" * 10 for r in w["reads"]]
    o = Oracle(w["newick"], w["labels"], seqs, 4, w["subst"], w["freqs"], w["rates"], pinv=0.3)
    ref = hostlib.Reference(w["newick"], w["labels"], seqs, states=4, subst=w["subst"], freqs=w["freqs"],
                            rates=w["rates"], pinv=0.3)
    ev = ref.evaluator()
    codes, wb, ws = epa.encode_queries(4, reads, compact=True)
    assert ws.max() > 1536
    pairs = np.zeros(ref.B * len(reads), epa.PAIR_DTYPE)
    pairs["branch_id"] = np.repeat(np.arange(ref.B), len(reads))
    pairs["seq_id"] = np.tile(np.arange(len(reads)), ref.B)
    res = ev.thorough(pairs, codes, wb, ws)
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], reads)
    assert np.max(np.abs(res["lnl"] - tl)) < 1e-6
    assert np.max(np.abs(res["distal_length"] - td)) < 1e-6


def test_prop_invariant_sites_golden_on_device():
    """the +I golden case (independent brute force) straight through the device path"""
    g = load_case("dna8_gtr_fu_i_g4")
    labels = [a for a, _ in g["msa"]]
    seqs = [b for _, b in g["msa"]]
    ref = hostlib.Reference(g["newick"], labels, seqs, states=4, subst=g["subst"], freqs=g["freqs"],
                            rates=g["gamma_rates"], pinv=g["pinv"])
    assert abs(ref.tree_lnl(0) - g["tree_lnl"]) < 1e-8
    ev = ref.evaluator()
    assert abs(ev.tree_logl(3) - g["tree_lnl"]) < 1e-7
    qs = [q["seq"] for q in g["queries"]]
    codes, wb, ws = epa.encode_queries(4, qs)
    assert np.max(np.abs(ev.preplace(codes, wb, ws) - np.array(g["preplace"]))) < 1e-6
    pairs = np.zeros(ref.B * len(qs), epa.PAIR_DTYPE)
    pairs["branch_id"] = np.repeat(np.arange(ref.B), len(qs))
    pairs["seq_id"] = np.tile(np.arange(len(qs)), ref.B)
    res = ev.thorough(pairs, codes, wb, ws)
    for i, p in enumerate(pairs):
        e = g["thorough"][p["seq_id"]][p["branch_id"]]
        assert abs(res["lnl"][i] - e["lnl"]) < 1e-6
        assert abs(res["distal_length"][i] - e["distal"]) < 1e-6
        assert abs(res["pendant_length"][i] - e["pendant"]) < 1e-6 * max(1.0, e["pendant"])


def test_very_large_reference_streamed_select():
    """B = 16597 (> 16384): the candidate selection streams the row (k_select_big); checked against
    the host restatement of the dynamic heuristic on the same table."""
    from epa_ng_amd import synth
    w = synth.dna_workload(8300, 24, 6, 24, (75, 76, 77))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    assert ref.B == 2 * 8300 - 3
    ev = ref.evaluator()
    codes, wb, ws = epa.encode_queries(4, w["reads"], compact=True)
    lnl = ev.preplace(codes, wb, ws)
    pairs = ev.select(lnl, len(w["reads"]), 0.99999, max_pairs=len(w["reads"]) * ref.B)
    hb, hs = hostlib.heuristic(lnl, "dynamic", 0.99999)
    assert sorted(zip(hb.tolist(), hs.tolist())) == sorted(zip(pairs["branch_id"].tolist(), pairs["seq_id"].tolist()))
    assert np.all(np.diff(pairs["branch_id"].astype(np.int64)) >= 0)


@pytest.mark.parametrize("states,acc", [(4, False), (4, True), (20, False)])
def test_no_heur_all_pairs_lwr_and_filter_on_device(states, acc):
    """epa_dev_place_all (--no-heur: pairs generated, placed, LWR'd and filtered on the device) vs
    thorough placement of all pairs + the reference's compute_and_set_lwr / filter restated in
    numpy (support-threshold and accumulated mode, min / max clamps, the min - 1 top-up quirk)."""
    from epa_ng_amd import synth
    if states == 4:
        w = synth.dna_workload(40, 300, 90, 80, (111, 112, 113))
    else:
        w = synth.aa_workload(12, 200, 30, 60, (114, 115, 116))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=states, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    reads = list(w["reads"])
    codes, wb, ws = epa.encode_queries(states, reads, compact=True)
    Q, B = len(reads), ref.B
    pairs = np.zeros(B * Q, epa.PAIR_DTYPE)
    pairs["branch_id"] = np.repeat(np.arange(B), Q)
    pairs["seq_id"] = np.tile(np.arange(Q), B)
    full = ev.thorough(pairs, codes, wb, ws)
    lnl = full["lnl"].reshape(B, Q)
    for (thr, mn, mx) in ((0.01, 1, 7), (0.3, 3, 5), (0.9999, 1, 2)) if not acc else ((0.95, 1, 7), (0.5, 4, 6)):
        got = ev.place_all(codes, wb, ws, min_lwr=thr, acc=acc, filter_min=mn, filter_max=mx)
        assert ev.last_stats["pairs"] == B * Q
        for q in range(Q):
            col = lnl[:, q]
            e = np.exp(col - col.max())
            lw = e / e.sum()
            order = np.lexsort((np.arange(B), -col))
            if not acc:
                k = 0
                while k < B and lw[order[k]] > thr:
                    k += 1
                if k < mn:
                    k = min(B, mn)
                k = min(k, mx)
            else:
                k, s_ = 0, 0.0
                while k < B and k < mx and s_ < thr:
                    s_ += lw[order[k]]
                    k += 1
                if k + 1 < mn:
                    k = min(B, mn - 1)
            b_, l_, p_, d_, w_ = got[q]
            assert list(b_) == list(order[:k]), (q, thr, mn, mx, list(b_), list(order[:k + 1]), lw[order[:k + 2]])
            assert np.array_equal(l_, col[order[:k]])
            assert np.max(np.abs(w_ - lw[order[:k]])) < 1e-12
            assert np.array_equal(p_, full["pendant_length"].reshape(B, Q)[order[:k], q])


def test_cli_several_devices_same_jplace(tmp_path):
    """--devices a,b: one worker thread per GPU, chunks dealt in file order; the jplace must not
    depend on the device count (two contexts on GPU 0 stand in for two GPUs)."""
    import subprocess
    from epa_ng_amd import synth
    w = synth.dna_workload(30, 400, 900, 100, (121, 122, 123))
    tre, aln, qf = tmp_path / "r.tre", tmp_path / "r.fasta", tmp_path / "q.fasta"
    tre.write_text(w["newick"] + "\n")
    with open(aln, "w") as f:
        for l, s in zip(w["labels"], w["seqs"]):
            f.write(">%s\n%s\n" % (l, s))
    with open(qf, "w") as f:
        for i, s in enumerate(w["reads"]):
            f.write(">q%d\n%s\n" % (i, s))
    exe = hostlib.cli_exe()
    model = "GTR{%s}+FU{%s}+G4{0.478218}" % ("/".join(map(repr, w["subst"])), "/".join(map(repr, w["freqs"])))
    outs = []
    # the last run keeps the default --device-min-chunk: the 100-read chunks the user asked for are
    # read as one device chunk (the reference's --chunk-size is a memory / speed knob) -- same jplace
    for devs, extra, exact in (("0", [], True), ("0,0", [], True), ("0,0,0", ["--no-heur"], True),
                               ("0", ["--no-heur"], True), ("0", [], False)):
        od = tmp_path / ("out_" + devs.replace(",", "_") + ("_nh" if extra else "") + ("" if exact else "_merged"))
        od.mkdir()
        r = subprocess.run([exe, "-t", str(tre), "-s", str(aln), "-q", str(qf), "-m", model, "-w", str(od),
                            "--chunk-size", "100", "--devices", devs] + (["--device-min-chunk", "0"] if exact else []) + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        jp = json.load(open(od / "epa_result.jplace"))
        jp.pop("metadata", None)
        outs.append(jp)
        assert len(jp["placements"]) == 900
    assert outs[0] == outs[1]
    assert outs[2] == outs[3]
    assert outs[4] == outs[0]
    assert [p["n"] for p in outs[0]["placements"]] == [["q%d" % i] for i in range(900)]


def test_wrong_max_span_is_rejected():
    """the caller's max_span sizes kernel variants and packed rows: a longer window is an input
    error (EPA_ERR_QUERY_WIDTH), not a silently truncated sum -- also for device-resident inputs"""
    import torch
    from epa_ng_amd import synth
    w = synth.dna_workload(16, 400, 40, 180, (131, 132, 133))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    codes, wb, ws = epa.encode_queries(4, w["reads"], compact=True)
    dc = torch.from_numpy(codes).cuda()
    db = torch.from_numpy(wb.view(np.int32)).cuda()
    ds = torch.from_numpy(ws.view(np.int32)).cuda()
    pairs, res = ev.place_chunk(dc, db, ds, Q=len(wb), max_span=int(ws.max()))
    assert len(pairs) > 0
    with pytest.raises(epa.EpaError) as ei:
        ev.place_chunk(dc, db, ds, Q=len(wb), max_span=100)
    assert ei.value.code == -4


def test_cli_rooted_tree_preserve_rooting(tmp_path):
    """rooted reference tree through the CLI: with --preserve-rooting on (default) the jplace
    carries the rooted tree and every placement is mapped by rtree_mapper::in_rtree
    (src/io/jplace_util.cpp:20-26); off reports the unrooted working tree.  The two runs must agree
    placement by placement through the mapping (the mapping itself is pinned by the reference's
    literals in tests/test_host_cpu.py)."""
    import subprocess
    from epa_ng_amd import synth
    root = synth.random_tree(14, 141)
    rates = synth.gamma_rates(0.478218)
    labels, seqs = synth.simulate_msa(root, 300, synth.CFG2_SUBST, synth.CFG2_FREQS, rates, 142)
    reads, _ = synth.make_reads(seqs, 40, 120, 0.03, 143)
    k0, k1, k2 = root.kids

    def sub(node):
        r = synth.Node()
        r.kids = [node]
        return synth.newick(r)[1:-2]
    l2 = k2.length
    k2.length = 0.6 * l2
    rooted = "((%s,%s):%r,%s);" % (sub(k0), sub(k1), 0.4 * l2, sub(k2))
    k2.length = l2
    tre, aln, qf = tmp_path / "r.tre", tmp_path / "r.fasta", tmp_path / "q.fasta"
    tre.write_text(rooted + "\n")
    with open(aln, "w") as f:
        for l, s in zip(labels, seqs):
            f.write(">%s\n%s\n" % (l, s))
    with open(qf, "w") as f:
        for i, s in enumerate(reads):
            f.write(">q%d\n%s\n" % (i, s))
    exe = hostlib.cli_exe()
    model = "GTR{%s}+FU{%s}+G4{0.478218}" % ("/".join(map(repr, synth.CFG2_SUBST)), "/".join(map(repr, synth.CFG2_FREQS)))
    B = 2 * 14 - 3
    out = {}
    for mode in ("on", "off"):
        od = tmp_path / mode
        od.mkdir()
        r = subprocess.run([exe, "-t", str(tre), "-s", str(aln), "-q", str(qf), "-m", model, "-w", str(od),
                            "--no-heur", "--filter-max", str(B), "--filter-min-lwr", "0",
                            "--preserve-rooting", mode], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        out[mode] = json.load(open(od / "epa_result.jplace"))
    assert out["on"]["tree"].count("{") == B + 1 and out["off"]["tree"].count("{") == B
    ref = hostlib.Reference(rooted, labels, seqs, states=4, subst=synth.CFG2_SUBST, freqs=synth.CFG2_FREQS,
                            rates=rates)
    assert out["on"]["tree"] == ref.numbered_newick(10)
    sides = set()
    for pon, poff in zip(out["on"]["placements"], out["off"]["placements"]):
        assert pon["n"] == poff["n"] and len(pon["p"]) == len(poff["p"])
        for a, b in zip(pon["p"], poff["p"]):
            e, d = ref.in_rtree(b[0], b[3])
            assert a[0] == e and abs(a[3] - d) < 2e-10
            assert a[1] == b[1] and a[2] == b[2] and a[4] == b[4]
            if b[0] == B - 1:
                sides.add(e)
    assert sides == {B, ref.in_rtree(B - 1, 0.99 * l2)[0]}   # both halves of the former root edge got placements


@pytest.mark.parametrize("tips,width,reads,read_len", [(96, 300, 200, 90), (2200, 64, 24, 40), (8300, 24, 4, 24)])
def test_fixed_and_baseball_heuristics_on_device(tips, width, reads, read_len):
    """--fix-heur / --baseball-heur candidate selection on the device (epa_dev_set_heuristic) vs the
    host restatement of until_top_percent and baseball_heuristic (src/set_manipulators.cpp:82-88,
    src/core/heuristics.hpp:70-117) on the same table; all three selection kernels (row in one
    wave's registers, in a workgroup's registers, streamed)."""
    from epa_ng_amd import synth
    w = synth.dna_workload(tips, width, reads, read_len, (151, 152, 153))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    codes, wb, ws = epa.encode_queries(4, w["reads"], compact=True)
    lnl = ev.preplace(codes, wb, ws)
    Q = len(w["reads"])

    def as_set(b, s):
        return sorted(zip(np.asarray(b).tolist(), np.asarray(s).tolist()))
    for mode, param in (("fixed", 0.02), ("fixed", 0.3), ("fixed", 0.0004), ("baseball", 0.0), ("dynamic", 0.0)):
        ev.set_heuristic(mode, param)
        pairs = ev.select(lnl, Q, 0.99999, max_pairs=Q * ref.B)
        hb, hs = hostlib.heuristic(lnl, mode, param if mode == "fixed" else 0.99999)
        assert as_set(hb, hs) == as_set(pairs["branch_id"], pairs["seq_id"]), (mode, param)
        assert np.all(np.diff(pairs["branch_id"].astype(np.int64)) >= 0)
        if mode == "fixed":
            assert len(pairs) == Q * min(ref.B, int(np.ceil(param * ref.B)))
    # fused chunk with the baseball rule == select + thorough
    ev.set_heuristic("baseball")
    pairs = ev.select(lnl, Q, 0.99999, max_pairs=Q * ref.B)
    p2, r2 = ev.place_chunk(codes, wb, ws, max_pairs=Q * ref.B)
    assert np.array_equal(p2, pairs)
    res = ev.thorough(pairs, codes, wb, ws)
    assert np.array_equal(r2["lnl"], res["lnl"])
    ev.set_heuristic("dynamic")


def test_cli_fix_and_baseball_heuristics(tmp_path):
    """-G / --baseball-heur through the CLI run fused on the device; same jplace as with the host
    heuristic path (--host-heuristic keeps the table round trip for this cross-check)."""
    import subprocess
    from epa_ng_amd import synth
    w = synth.dna_workload(40, 300, 120, 100, (161, 162, 163))
    tre, aln, qf = tmp_path / "r.tre", tmp_path / "r.fasta", tmp_path / "q.fasta"
    tre.write_text(w["newick"] + "\n")
    with open(aln, "w") as f:
        for l, s in zip(w["labels"], w["seqs"]):
            f.write(">%s\n%s\n" % (l, s))
    with open(qf, "w") as f:
        for i, s in enumerate(w["reads"]):
            f.write(">q%d\n%s\n" % (i, s))
    exe = hostlib.cli_exe()
    model = "GTR{%s}+FU{%s}+G4{0.478218}" % ("/".join(map(repr, w["subst"])), "/".join(map(repr, w["freqs"])))
    for flags in (["-G", "0.1"], ["--baseball-heur"]):
        outs = []
        for host in (False, True):
            od = tmp_path / ("o_%s_%d" % (flags[0].strip("-"), host))
            od.mkdir()
            r = subprocess.run([exe, "-t", str(tre), "-s", str(aln), "-q", str(qf), "-m", model, "-w", str(od)] + flags
                               + (["--host-heuristic"] if host else []), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout + r.stderr
            jp = json.load(open(od / "epa_result.jplace"))
            jp.pop("metadata", None)
            outs.append(jp)
        assert outs[0] == outs[1], flags


def test_cli_model_file(tmp_path):
    """-m <file>: the RAxML 8 info file of the reference's test data parses to the cfg1 descriptor
    (test/src/parse_model.cpp:7-13), so the run must equal the one with the descriptor itself."""
    import subprocess
    g = load_case("dna8_gtr_fu_g4")
    data = os.path.join(GOLDEN, "data")
    qf = tmp_path / "q.fasta"
    with open(qf, "w") as f:
        for q in g["queries"]:
            f.write(">%s\n%s\n" % (q["name"], q["seq"]))
    exe = hostlib.cli_exe()
    desc = ("GTR{0.787874/1.821672/1.294006/0.698421/3.034135/1.000000}+FU{0.256465/0.222535/0.308594/"
            "0.212406}+G4{0.478218}")
    outs = []
    for m in (desc, os.path.join(data, "modelfiles", "rax8_dna")):
        od = tmp_path / ("o%d" % len(outs))
        od.mkdir()
        r = subprocess.run([exe, "-t", os.path.join(data, "ref.tre"), "-s", os.path.join(data, "aln.fasta"),
                            "-q", str(qf), "-m", m, "-w", str(od)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "GTR{0.787874/1.821672/1.294006/0.698421/3.034135/1.000000}" in r.stdout
        jp = json.load(open(od / "epa_result.jplace"))
        jp.pop("metadata", None)
        outs.append(jp)
    assert outs[0] == outs[1] and len(outs[0]["placements"]) == len(g["queries"])


def test_fourbit_wire_format_same_results():
    """queries in the 4-bit wire format (epa_dev_set_query_packing): expanded on the device, every
    entry point returns exactly what it returns for one-byte codes; aligned rows of odd length and
    compact rows, host arrays and device-resident buffers; rejected for 20 states"""
    import torch
    from epa_ng_amd import synth
    w = synth.dna_workload(24, 701, 300, 120, (171, 172, 173))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    reads = list(w["reads"])
    reads[3] = reads[3].replace("A", "R", 2).replace("C", "N", 1)   # ambiguity codes survive the nibbles
    for compact in (False, True):
        codes, wb, ws = epa.encode_queries(4, reads, compact=compact)
        packed = epa.pack_codes_4bit(codes)
        assert packed.data.shape[1] == (codes.shape[1] + 1) // 2
        lnl = ev.preplace(codes, wb, ws)
        assert np.array_equal(ev.preplace(packed, wb, ws), lnl)
        p1, r1 = ev.place_chunk(codes, wb, ws, max_span=int(ws.max()))
        p2, r2 = ev.place_chunk(packed, wb, ws, max_span=int(ws.max()))
        assert np.array_equal(p1, p2) and np.array_equal(r1, r2)
        dev = epa.Packed4(torch.from_numpy(packed.data).cuda(), packed.stride)
        p3, r3 = ev.place_chunk(dev, torch.from_numpy(wb.view(np.int32)).cuda(),
                                torch.from_numpy(ws.view(np.int32)).cuda(), Q=len(reads), max_span=int(ws.max()))
        assert np.array_equal(p1, p3) and np.array_equal(r1, r3)
        assert np.array_equal(ev.thorough(p1, packed, wb, ws), ev.thorough(p1, codes, wb, ws))
    wa = synth.aa_workload(8, 60, 4, 30, (174, 175, 176))
    ra = hostlib.Reference(wa["newick"], wa["labels"], wa["seqs"], states=20, subst=wa["subst"],
                           freqs=wa["freqs"], rates=wa["rates"])
    ea = ra.evaluator()
    with pytest.raises(epa.EpaError):
        ea._check(ea.L.epa_dev_set_query_packing(ea.h, 4))


def test_cli_bfast_queries(tmp_path):
    """-q <file>.bfast: the reference's own binary-fasta fixture (test/data/query.fasta.bin) gives
    the same jplace as its query.fasta"""
    import subprocess
    data = os.path.join(GOLDEN, "data")
    exe = hostlib.cli_exe()
    outs = []
    for q in ("query.fasta", "query.fasta.bin"):
        od = tmp_path / q.replace(".", "_")
        od.mkdir()
        r = subprocess.run([exe, "-t", os.path.join(data, "ref.tre"), "-s", os.path.join(data, "aln.fasta"),
                            "-q", os.path.join(data, q), "-m", "GTR+G", "-w", str(od)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        jp = json.load(open(od / "epa_result.jplace"))
        jp.pop("metadata", None)
        outs.append(jp)
    assert outs[0] == outs[1] and len(outs[0]["placements"]) == 2


@pytest.mark.parametrize("packed", [False, True])
def test_staged_chunk_pipeline_equals_place_chunk(packed):
    """epa_dev_chunk_stage / _launch / _finish (H2D of chunk k+1 and D2H of chunk k-1 on the copy
    stream while chunk k computes) must return, chunk for chunk, the bits of epa_dev_place_chunk;
    also: candidate overflow leaves the slot staged, state errors are loud, results can stay on
    the device."""
    import torch
    w = synth.dna_workload(64, 700, 2400, 150, (61, 62, 63))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    Q = 400
    chunks = []
    for c in range(6):
        codes, wb, ws = epa.encode_queries(4, w["reads"][c * Q:(c + 1) * Q], compact=True)
        chunks.append((epa.pack_codes_4bit(codes) if packed else codes, wb, ws))
    expect = [ev.place_chunk(*ch, max_span=150) for ch in chunks]
    cap = Q * 64
    with pytest.raises(epa.EpaError):
        ev.chunk_launch(0, max_span=150, max_pairs=cap)          # nothing staged
    ev.chunk_stage(0, *chunks[0])
    with pytest.raises(epa.EpaError):
        ev.chunk_launch(0, max_span=150, max_pairs=Q)            # candidate overflow: stays staged
    got = []
    for k in range(len(chunks)):
        ev.chunk_launch(k & 1, max_span=150, max_pairs=cap)
        if k:
            got.append(ev.chunk_finish((k - 1) & 1))
        if k + 1 < len(chunks):
            if k:
                with pytest.raises(epa.EpaError):
                    ev.chunk_stage(k & 1, *chunks[k + 1])        # that slot's launch is in flight
            ev.chunk_stage((k + 1) & 1, *chunks[k + 1])
    got.append(ev.chunk_finish((len(chunks) - 1) & 1))
    with pytest.raises(epa.EpaError):
        ev.chunk_finish(0)                                       # nothing in flight
    for (p, r), (ep, er) in zip(got, expect):
        assert np.array_equal(p, ep)
        assert np.array_equal(r["lnl"], er["lnl"]) and np.array_equal(r["pendant_length"], er["pendant_length"])
        assert np.array_equal(r["distal_length"], er["distal_length"])
    # results kept on the device in caller buffers (the send buffers of the N > 1 gather)
    dev = torch.device("cuda", 0)
    d_pairs = torch.zeros((cap, 2), dtype=torch.int32, device=dev)
    d_res = torch.zeros((cap, 3), dtype=torch.float64, device=dev)
    ev.chunk_stage(1, *chunks[2])
    ev.chunk_launch(1, max_span=150, max_pairs=cap, pairs_out=d_pairs, results_out=d_res, keep_on_device=True)
    n = ev.chunk_finish_device(1)
    torch.cuda.synchronize()
    assert n == len(expect[2][0])
    assert np.array_equal(d_pairs[:n, 0].cpu().numpy().view(np.uint32), expect[2][0]["branch_id"])
    assert np.array_equal(d_res[:n, 0].cpu().numpy(), expect[2][1]["lnl"])
    # the same with EPA_CHUNK_HOST_ORDERED (no stream-level ordering against the context's stream: rows read after finish)
    d_pairs.zero_(); d_res.zero_()
    torch.cuda.synchronize()
    ev.chunk_stage(0, *chunks[3])
    ev.chunk_launch(0, max_span=150, max_pairs=cap, pairs_out=d_pairs, results_out=d_res, keep_on_device=True,
                    host_ordered=True)
    n = ev.chunk_finish_device(0)
    assert n == len(expect[3][0])
    assert np.array_equal(d_pairs[:n, 0].cpu().numpy().view(np.uint32), expect[3][0]["branch_id"])
    assert np.array_equal(d_res[:n, 0].cpu().numpy(), expect[3][1]["lnl"])


@pytest.mark.parametrize("packed", [False, True])
def test_staged_chunks_in_hbm_and_two_half_launch_order(packed):
    """chunk_stage of DEVICE arrays reads them in place (no copy), and the two-half loop bench.py runs
    -- launch_begin(k); finish(k-1); stage(k+1); launch_end(k) -- returns, chunk for chunk, the bits
    of epa_dev_place_chunk, for host chunks (even row length: the flat 4-bit expansion; odd: byte by
    byte) and for HBM-resident ones; mixing host and device arrays in one stage call is an error"""
    import torch
    w = synth.dna_workload(48, 640, 1500, 150, (71, 72, 73))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    Q = 300
    cap = Q * 64
    dev = torch.device("cuda", 0)
    for odd in (False, True):
        chunks = []
        for c in range(5):
            codes, wb, ws = epa.encode_queries(4, w["reads"][c * Q:(c + 1) * Q], compact=True)
            if odd:                                   # rows of 151 codes: one trailing gap column
                codes = np.concatenate([codes, np.full((len(codes), 1), 15, np.uint8)], axis=1)
                codes = np.ascontiguousarray(codes)
            chunks.append((codes, wb, ws))
        expect = [ev.place_chunk(*ch, max_span=150) for ch in chunks]
        if packed:
            host = [(epa.pack_codes_4bit(c), b, s) for c, b, s in chunks]
            hbm = [(epa.Packed4(torch.from_numpy(p.data).to(dev), p.stride), torch.from_numpy(b.view(np.int32)).to(dev),
                    torch.from_numpy(s.view(np.int32)).to(dev)) for p, b, s in host]
        else:
            host = chunks
            hbm = [(torch.from_numpy(c).to(dev), torch.from_numpy(b.view(np.int32)).to(dev),
                    torch.from_numpy(s.view(np.int32)).to(dev)) for c, b, s in chunks]
        # host_ordered (EPA_CHUNK_HOST_ORDERED): the same loop without the library's stream-level ordering against the
        # context's stream -- legal here (inputs synchronised, rows read after finish), same bits
        for src, ho in ((host, False), (hbm, False), (host, True), (hbm, True)):
            if ho:
                torch.cuda.synchronize()
            got = []
            ev.chunk_stage(0, *src[0])
            for k in range(len(src)):
                ev.chunk_launch_begin(k & 1, max_span=150, max_pairs=cap, host_ordered=ho)
                if k:
                    got.append(ev.chunk_finish((k - 1) & 1))
                if k + 1 < len(src):
                    ev.chunk_stage((k + 1) & 1, *src[k + 1])
                ev.chunk_launch_end(k & 1)
            got.append(ev.chunk_finish((len(src) - 1) & 1))
            for (p, r), (ep, er) in zip(got, expect):
                assert np.array_equal(p, ep) and np.array_equal(r, er)
    with pytest.raises(AssertionError):
        ev.chunk_stage(0, hbm[0][0], chunks[0][1], chunks[0][2])
    with pytest.raises(epa.EpaError):                 # straight at the C-ABI: device codes, host windows
        ev._check(ev.L.epa_dev_chunk_stage(ev.h, 0, hbm[0][0].data.data_ptr() if packed else hbm[0][0].data_ptr(),
                                           chunks[0][1].ctypes.data, chunks[0][2].ctypes.data, Q))


def test_host_ordered_launch_as_first_call_on_a_cold_context():
    """the CLI's loop passes EPA_CHUNK_HOST_ORDERED on every launch and never builds
    the lookup beforehand, so the lazily built tables (k_build_lookup / k_build_lookup2 on the context's
    stream) must be complete before a slot's non-blocking stream reads them -- on BOTH slots' first calls.
    A large B x W makes the build long enough for a race to show; the rows must be the bits a warm context
    gives, for group launches as well."""
    w = synth.dna_workload(400, 2000, 600, 150, (91, 92, 93))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    Q = 200
    chunks = []
    for c in range(3):
        codes, wb, ws = epa.encode_queries(4, w["reads"][c * Q:(c + 1) * Q], compact=True)
        chunks.append((epa.pack_codes_4bit(codes), wb, ws))
    warm = ref.evaluator()
    expect = [warm.place_chunk(*ch, max_span=150) for ch in chunks]
    kw = dict(threshold=0.99999, max_span=150, max_pairs=Q * 64, host_ordered=True)
    for rep in range(3):
        ev = ref.evaluator()                      # cold: nothing has built the lookup
        ev.chunk_stage(0, *chunks[0])
        ev.chunk_stage(1, *chunks[1])
        ev.chunk_launch_begin(0, **kw)            # builds the lookup lazily ...
        ev.chunk_launch_begin(1, **kw)            # ... and the other slot's stream starts right behind
        ev.chunk_launch_end(0)
        ev.chunk_launch_end(1)
        got = [ev.chunk_finish(0), ev.chunk_finish(1)]
        for (p, r), (ep, er) in zip(got, expect):
            assert np.array_equal(p, ep) and np.array_equal(r, er)
        del ev
    ev = ref.evaluator()                          # cold again: a group launch as the first call
    for j in range(3):
        ev.chunk_stage(j, *chunks[j])
    ev.chunk_launch_many_begin([0, 1, 2], threshold=0.99999, max_span=150, max_pairs=3 * Q * 64, host_ordered=True)
    ev.chunk_launch_end(0)
    for j in range(3):
        p, r = ev.chunk_finish(j)
        assert np.array_equal(p, expect[j][0]) and np.array_equal(r, expect[j][1])


def test_five_slot_order_for_small_chunks():
    """the order bench.py's chunk5000 leg runs -- three chunks begun ahead on five slots, a chunk's
    Newton kernel queued while the previous chunk's still runs -- returns, chunk for chunk, the bits of
    epa_dev_place_chunk; slots beyond 5 are an error"""
    w = synth.dna_workload(40, 600, 1800, 150, (81, 82, 83))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    Q, S, A = 150, 5, 3
    chunks = []
    for c in range(12):
        codes, wb, ws = epa.encode_queries(4, w["reads"][c * Q:(c + 1) * Q], compact=True)
        chunks.append((epa.pack_codes_4bit(codes), wb, ws))
    expect = [ev.place_chunk(*ch, max_span=150) for ch in chunks]
    kw = dict(threshold=0.99999, max_span=150, max_pairs=Q * 64)
    n = len(chunks)
    got = [None] * n
    for k in range(A):
        ev.chunk_stage(k % S, *chunks[k])
        ev.chunk_launch_begin(k % S, **kw)
    with pytest.raises(epa.EpaError):
        ev.chunk_stage(0, *chunks[0])            # begun, not ended: the slot is busy
    for k in range(n):
        ev.chunk_launch_end(k % S)
        if k >= 2:
            got[k - 2] = ev.chunk_finish((k - 2) % S)
        if k + A < n:
            ev.chunk_stage((k + A) % S, *chunks[k + A])
            ev.chunk_launch_begin((k + A) % S, **kw)
    for k in range(n - 2, n):
        got[k] = ev.chunk_finish(k % S)
    for (p, r), (ep, er) in zip(got, expect):
        assert np.array_equal(p, ep) and np.array_equal(r, er)
    with pytest.raises(epa.EpaError):
        ev.chunk_stage(24, *chunks[0])


def test_group_launch_of_small_chunks_equals_per_chunk_launches():
    """epa_dev_chunk_launch_many: ONE chunk body (one preplacement, one selection, one Newton launch) over the
    concatenated queries of several staged slots, rows regrouped by chunk afterwards.  Every member's finish()
    returns the bits of epa_dev_place_chunk on that chunk alone -- branch-major order, chunk-local sequence ids --
    for groups of 4, 3 and 1 chunks of DIFFERENT sizes, host-staged (4-bit wire) and HBM-resident members mixed,
    members finished in any order; the bench's pipelined order (groups of four on twelve slots, two groups begun
    ahead); the Newton counters of a group are those of its members' own launches added up."""
    import torch
    w = synth.dna_workload(40, 600, 2600, 150, (81, 82, 83))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    sizes = [150, 97, 230, 1, 64, 200, 150, 33, 150, 150, 150, 150, 150, 150, 150, 150]
    chunks, lo = [], 0
    for q in sizes:
        codes, wb, ws = epa.encode_queries(4, w["reads"][lo:lo + q], compact=True)
        chunks.append((epa.pack_codes_4bit(codes), wb, ws))
        lo += q
    expect, rounds = [], []
    for ch in chunks:
        expect.append(ev.place_chunk(*ch, max_span=150))
        rounds.append((ev.last_stats["rounds"], ev.last_stats["newton_evals"]))
    kw = dict(threshold=0.99999, max_span=150, max_pairs=4 * 230 * 64)

    def same(got, k):
        p, r = got
        assert np.array_equal(p, expect[k][0]) and np.array_equal(r, expect[k][1])

    # --- a group of four (sizes 150 / 97 / 230 / 1), finished out of order
    for j in range(4):
        ev.chunk_stage(j, *chunks[j])
    ev.chunk_launch_many([0, 1, 2, 3], **kw)
    with pytest.raises(epa.EpaError):
        ev.chunk_stage(0, *chunks[0])                       # busy
    got_stats = {}
    for j in (2, 0, 3):
        same(ev.chunk_finish(j), j)
        got_stats[j] = dict(ev.last_stats)
    with pytest.raises(epa.EpaError):
        ev.chunk_stage(0, *chunks[4])                       # the leader's buffers still hold member 1's rows
    same(ev.chunk_finish(1), 1)
    got_stats[1] = dict(ev.last_stats)
    assert [got_stats[j]["pairs"] for j in range(4)] == [len(expect[j][0]) for j in range(4)]
    assert got_stats[0]["rounds"] == sum(rounds[j][0] for j in range(4))
    assert got_stats[0]["newton_evals"] == sum(rounds[j][1] for j in range(4))
    assert all(got_stats[j]["rounds"] == 0 for j in (1, 2, 3))
    # --- a group of three led by another slot, one member HBM-resident and unpacked; then a "group" of one
    dev = torch.device("cuda", 0)
    c5 = epa.encode_queries(4, w["reads"][sum(sizes[:5]):sum(sizes[:6])], compact=True)
    with pytest.raises(epa.EpaError):                       # mixed layouts in one group: refused, nothing launched
        ev.chunk_stage(7, *chunks[4]); ev.chunk_stage(5, *c5)
        ev.chunk_launch_many([7, 5], **kw)
    # (slot 7 holds chunk 4 packed, slot 5 chunk 5 unpacked: launch them one by one instead)
    ev.chunk_launch(7, **kw); same(ev.chunk_finish(7), 4)
    ev.chunk_launch(5, **kw); same(ev.chunk_finish(5), 5)
    for j, k in ((9, 4), (4, 5), (6, 6)):
        ev.chunk_stage(j, *chunks[k])
    ev.chunk_launch_many_begin([9, 4, 6], **kw)
    with pytest.raises(epa.EpaError):
        ev.chunk_launch_end(4)                              # a member, not the leader
    ev.chunk_launch_end(9)
    for j, k in ((9, 4), (4, 5), (6, 6)):
        same(ev.chunk_finish(j), k)
    ev.chunk_stage(2, *chunks[7])
    ev.chunk_launch_many([2], **kw)
    same(ev.chunk_finish(2), 7)
    # --- candidate overflow leaves every member staged: relaunch with room
    for j in range(3):
        ev.chunk_stage(j, *chunks[j])
    with pytest.raises(epa.EpaError) as ei:
        ev.chunk_launch_many([0, 1, 2], threshold=0.99999, max_span=150, max_pairs=100)
    assert ei.value.code == -9                          # EPA_ERR_PAIR_OVERFLOW
    ev.chunk_launch_many([0, 1, 2], **kw)
    for j in range(3):
        same(ev.chunk_finish(j), j)
    # --- the pipelined order of bench.py's chunk5000 leg: groups of four on twelve slots, two groups begun ahead
    groups = [list(range(8 + 4 * g, 12 + 4 * g)) for g in range(2)] + [[0, 1, 2, 3]]      # chunk indices per group
    slots_of = lambda g: [(g % 3) * 4 + j for j in range(4)]

    def begin(g):
        for sl, k in zip(slots_of(g), groups[g]):
            ev.chunk_stage(sl, *chunks[k])
        ev.chunk_launch_many_begin(slots_of(g), **kw)
    begin(0); begin(1)
    for g in range(len(groups)):
        ev.chunk_launch_end(slots_of(g)[0])
        if g >= 1:
            for sl, k in zip(slots_of(g - 1), groups[g - 1]):
                same(ev.chunk_finish(sl), k)
        if g + 2 < len(groups):
            begin(g + 2)
    for sl, k in zip(slots_of(len(groups) - 1), groups[-1]):
        same(ev.chunk_finish(sl), k)


def test_queued_thorough_launch_equals_host_launched(monkeypatch):
    """option queued_thorough (opt-in): the pair list and the Newton kernel are queued behind the selection before the
    host has seen the candidate count, guarded on the device by the read-back block the host checks afterwards
    (launch_thorough_queued).  Same bits as the host-launched order -- for a chunk of one read length (the queued
    kernel runs), for mixed read lengths (the queued kernel must exit, the ordinary per-class launches run), through
    the fused call and the five-slot staged order; a candidate overflow is still reported and still recoverable."""
    w = synth.dna_workload(40, 600, 1500, 150, (181, 182, 183))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    one_len = list(w["reads"][:900])
    mixed = []
    for i, r in enumerate(w["reads"][900:1500]):     # shorten every third read to 70 sites: span classes 11 and 10
        if i % 3 == 0:
            k = [j for j, ch in enumerate(r) if ch != "-"]
            r = "".join(ch if (j <= k[69] or ch == "-") else "-" for j, ch in enumerate(r)) if len(k) > 70 else r
        mixed.append(r)
    for reads in (one_len, mixed):
        codes, wb, ws = epa.encode_queries(4, reads, compact=True)
        ev.set_option("queued_thorough", 0)
        p0, r0 = ev.place_chunk(codes, wb, ws, max_span=150)
        st0 = dict(ev.last_stats)
        ev.set_option("queued_thorough", 1)
        p1, r1 = ev.place_chunk(codes, wb, ws, max_span=150)
        st1 = dict(ev.last_stats)
        assert len(p0) > len(reads) and np.array_equal(p0, p1) and np.array_equal(r0, r1)
        assert st0 == st1
        # staged order, three chunks begun ahead
        Q, S, A = 150, 5, 3
        chunks = []
        for c in range(len(reads) // Q):
            cc, cb, cs = epa.encode_queries(4, reads[c * Q:(c + 1) * Q], compact=True)
            chunks.append((epa.pack_codes_4bit(cc), cb, cs))
        ev.set_option("queued_thorough", 0)
        expect = [ev.place_chunk(*ch, max_span=150) for ch in chunks]
        ev.set_option("queued_thorough", 1)
        kw = dict(threshold=0.99999, max_span=150, max_pairs=Q * 64)
        n = len(chunks)
        got = [None] * n
        for k in range(min(A, n)):
            ev.chunk_stage(k % S, *chunks[k])
            ev.chunk_launch_begin(k % S, **kw)
        for k in range(n):
            ev.chunk_launch_end(k % S)
            if k >= 2:
                got[k - 2] = ev.chunk_finish((k - 2) % S)
            if k + A < n:
                ev.chunk_stage((k + A) % S, *chunks[k + A])
                ev.chunk_launch_begin((k + A) % S, **kw)
        for k in range(max(0, n - 2), n):
            got[k] = ev.chunk_finish(k % S)
        for (p, r), (ep, er) in zip(got, expect):
            assert np.array_equal(p, ep) and np.array_equal(r, er)
        # overflow: reported by launch_end, the slot stays staged, a larger max_pairs succeeds
        ev.chunk_stage(0, *chunks[0])
        ev.chunk_launch_begin(0, threshold=0.99999, max_span=150, max_pairs=8)
        with pytest.raises(epa.EpaError):
            ev.chunk_launch_end(0)
        ev.chunk_launch_begin(0, **kw)
        ev.chunk_launch_end(0)
        p, r = ev.chunk_finish(0)
        assert np.array_equal(p, expect[0][0]) and np.array_equal(r, expect[0][1])
        ev.set_option("queued_thorough", 0)


def test_selection_bitmap_and_sorted_staging_paths_agree(tmp_path):
    """the candidate list comes from a [B][Q] bitmap (default) or, for bitmaps over 64 MB, from
    staging rows + compaction + a stable device sort (the option select_sort forces that path): same pairs
    in the same (branch, query) order, same results, for the three selection rules"""
    import subprocess, sys
    script = tmp_path / "sel.py"
    script.write_text(
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import epa_ng_amd as epa\n"
        "from epa_ng_amd import hostlib, synth\n"
        "w = synth.dna_workload(96, 400, 700, 120, (91, 92, 93))\n"
        "ref = hostlib.Reference(w['newick'], w['labels'], w['seqs'], states=4, subst=w['subst'], freqs=w['freqs'], rates=w['rates'])\n"
        "ev = ref.evaluator()\n"
        "ev.set_option('select_sort', int(sys.argv[2]))\n"
        "codes, wb, ws = epa.encode_queries(4, w['reads'], compact=True)\n"
        "out = {}\n"
        "for mode, param in (('dynamic', 0.0), ('fixed', 0.05), ('baseball', 0.0)):\n"
        "    ev.set_heuristic(mode, param)\n"
        "    p, r = ev.place_chunk(codes, wb, ws, max_span=120, max_pairs=len(wb) * ref.B)\n"
        "    out[mode + '_p'] = p; out[mode + '_r'] = r\n"
        "np.savez(sys.argv[1], **out)\n" % (REPO, os.path.join(REPO, "tests")))
    outs = []
    for sort, name in ((0, "bitmap.npz"), (1, "sorted.npz")):
        r = subprocess.run([sys.executable, str(script), str(tmp_path / name), str(sort)], capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(np.load(tmp_path / name))
    for k in outs[0].files:
        assert len(outs[0][k]) > 0 and np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("states,tips,width,rl", [(4, 300, 900, 150), (20, 700, 260, 90), (4, 9, 200, 60)])
def test_selection_from_segment_maxima_equals_full_row_selection(states, tips, width, rl, monkeypatch):
    """The fused chunk body selects the dynamic rule's candidates from the per-(query, 64-branch segment) maxima the
    preplacement leaves behind (k_select_seg: only the segments near a row's maximum are read);
    the option select_full_rows sends the same chunk through the full-row kernel.  Same pairs in the same order, same
    results -- for a tree of 10 segments, one of 22 (20 states), one of a single partial segment, thresholds from
    lax to 1 - 1e-9, and reads with rare ambiguity codes (their rows have no maxima: rebuilt from the row)."""
    if states == 4:
        w = synth.dna_workload(tips, width, 600, rl, (131, 132, 133))
        reads = list(w["reads"])
        for i in range(0, len(reads), 17):     # rare codes: the generic preplacement kernel, no segment maxima
            r = list(reads[i])
            k = next(j for j, ch in enumerate(r) if ch != "-")
            r[k + 3] = "R"
            reads[i] = "".join(r)
    else:
        w = synth.aa_workload(tips, width, 300, rl, (134, 135, 136)) if hasattr(synth, "aa_workload") else None
        if w is None:
            subst, freqs = synth.aa_model(7)
            root = synth.random_tree(tips, 134)
            rates = synth.gamma_rates(0.6)
            labels, seqs = synth.simulate_msa(root, width, subst, freqs, rates, 135)
            rd, _ = synth.make_reads(seqs, 300, rl, 0.04, 136, states=20)
            w = dict(newick=synth.newick(root), labels=labels, seqs=seqs, subst=subst, freqs=freqs, rates=rates, reads=rd)
        reads = list(w["reads"])
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=states, subst=w["subst"], freqs=w["freqs"],
                            rates=w["rates"])
    ev = ref.evaluator()
    codes, wb, ws = epa.encode_queries(states, reads, compact=True)
    for thr in (0.9, 0.99999, 1.0 - 1e-9):
        ev.set_option("select_full_rows", 0)
        p_seg, r_seg = ev.place_chunk(codes, wb, ws, threshold=thr)
        ev.set_option("select_full_rows", 1)
        p_full, r_full = ev.place_chunk(codes, wb, ws, threshold=thr)
        ev.set_option("select_full_rows", 0)
        assert np.array_equal(p_seg, p_full), thr
        assert np.array_equal(r_seg["lnl"], r_full["lnl"])
    assert len(p_seg) >= len(reads)


def test_baseball_heuristic_counts_follow_the_reference_arithmetic():
    """baseball_heuristic (src/core/heuristics.hpp:74-117): hits = branches within 3.0 lnL of the
    best, then std::min(max_pitches - hits, max_strikes) more in size_t arithmetic -- 6 more when
    hits > 40 (the difference wraps), none at exactly 40, clamped at B (quirk D7)"""
    w = synth.dna_workload(40, 64, 4, 40, (71, 72, 73))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    B = ref.B
    assert B == 77
    rng = np.random.RandomState(2)
    hits = [1, 12, 34, 38, 40, 41, 50, 74, 77]
    lnl = np.empty((len(hits), B))
    for q, h in enumerate(hits):
        row = -1000.0 - 10.0 - rng.rand(B) * 50.0          # far outside the strike box
        idx = rng.permutation(B)[:h]
        row[idx] = -1000.0 - rng.rand(h) * 2.9              # inside it
        row[idx[0]] = -1000.0                               # the best
        lnl[q] = row
    ev.set_heuristic("baseball")
    pairs = ev.select(lnl, len(hits))
    hb, hs = hostlib.heuristic(lnl, "baseball")
    assert sorted(zip(hb.tolist(), hs.tolist())) == sorted(zip(pairs["branch_id"].tolist(), pairs["seq_id"].tolist()))
    got = np.bincount(pairs["seq_id"], minlength=len(hits))
    expect = [min(B, h + (6 if h > 40 else min(40 - h, 6))) for h in hits]
    assert got.tolist() == expect
    for q, h in enumerate(hits):                            # and they are the best ones
        chosen = set(pairs["branch_id"][pairs["seq_id"] == q].tolist())
        assert chosen == set(np.argsort(-lnl[q], kind="stable")[:expect[q]].tolist())


@pytest.mark.parametrize("states,model,flags,kw", [
    (20, "LG+G4{0.563473}", [], {}),                                 # BASELINE configs[2]'s model family
    (20, "WAG+F+G4{0.8}", ["--rate-scalers", "on"], {"rate_scalers": True}),
    (4, "GTR{0.7/1.8/1.2/0.6/3.0/1.0}+FU{0.25/0.23/0.30/0.22}+G8{0.5}", ["--raxml-blo"], {"raxml_blo": True}),
    (4, "HKY{1.0/3.5}+F+R3{0.2/1.0/3.0}{0.5/0.3/0.2}", ["--rate-scalers", "on", "--raxml-blo"],
     {"rate_scalers": True, "raxml_blo": True}),
])
def test_cli_named_models_rate_scalers_raxml_blo(tmp_path, states, model, flags, kw):
    """the model-string front end and the switches that used to be refused, end to end through the
    executable: named empirical AA matrices, named nucleotide models with symmetric rates, +F
    (frequencies counted on the reference MSA), +G8 / +R3, --rate-scalers, --raxml-blo.  The jplace
    must hold, per query, the placements the API gives for the same model and switches."""
    import subprocess
    subst, freqs = (synth.CFG2_SUBST, synth.CFG2_FREQS) if states == 4 else synth.aa_model(21)
    root = synth.random_tree(30, 91)
    labels, seqs = synth.simulate_msa(root, 200, subst, freqs, synth.gamma_rates(0.6), 92)
    reads, _ = synth.make_reads(seqs, 60, 100 if states == 4 else 70, 0.04, 93, states=states)
    nw = synth.newick(root)
    tre, aln, qf = tmp_path / "ref.tre", tmp_path / "ref.fasta", tmp_path / "q.fasta"
    tre.write_text(nw + "\n")
    with open(aln, "w") as f:
        for l, s in zip(labels, seqs):
            f.write(">%s\n%s\n" % (l, s))
    with open(qf, "w") as f:
        for i, s in enumerate(reads):
            f.write(">q%d\n%s\n" % (i, s))
    r = subprocess.run([hostlib.cli_exe(), "-t", str(tre), "-s", str(aln), "-q", str(qf), "-m", model,
                        "-w", str(tmp_path), "--filter-max", "30", "--filter-min-lwr", "1e-9"] + flags,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    jp = json.load(open(tmp_path / "epa_result.jplace"))
    assert len(jp["placements"]) == len(reads)
    # the executable premasks like the reference's front end (columns that are gaps in all of the
    # reference or in all of the 60 reads are cut out of both): the same columns for the API calls
    keep = hostlib.premask(aln, qf) == 0
    assert 0 < keep.sum() <= 200
    cut = lambda s: "".join(np.array(list(s))[keep])   # noqa: E731
    seqs, reads = [cut(s) for s in seqs], [cut(s) for s in reads]
    ref = hostlib.Reference(nw, labels, seqs, model=model)
    assert ref.s == states
    ev = ref.evaluator(**kw)
    codes, wb, ws = epa.encode_queries(states, reads, compact=True)
    pairs, res = ev.place_chunk(codes, wb, ws)
    for pq in jp["placements"]:
        q = int(pq["n"][0][1:])
        m = pairs["seq_id"] == q
        want = {int(b): (l, pl, dl) for b, l, pl, dl in zip(pairs["branch_id"][m], res["lnl"][m],
                                                            res["pendant_length"][m], res["distal_length"][m])}
        assert 1 <= len(pq["p"]) <= len(want)
        for edge, lnl, lwr, distal, pendant in pq["p"]:
            wl, wp, wd = want[edge]
            assert abs(lnl - wl) < 1e-6 * max(1.0, abs(wl)) and abs(pendant - wp) < 1e-8 and abs(distal - wd) < 1e-8
        assert abs(sum(p[2] for p in pq["p"]) - 1.0) < 1e-6 or len(pq["p"]) < len(want)
    # the same model through the oracle: parity of what the CLI computed
    from oracle_lib import Oracle
    mdl = ref.model()
    o = Oracle(nw, labels, seqs, states, ref.subst(), mdl["freqs"], mdl["rates"], weights=mdl["weights"],
               rate_scalers=kw.get("rate_scalers", False))
    o.set_raxml_blo(kw.get("raxml_blo", False))
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], reads)
    assert np.max(np.abs(res["lnl"] - tl)) < 1e-6


@pytest.mark.parametrize("scaling,gather", [("weak", "epa_comm"), ("strong", "epa_comm"), ("weak", "torch")])
def test_bench_two_ranks_on_one_gpu(scaling, gather):
    """bench.py's N > 1 branch end to end on the 1-GPU box: two ranks, both on device 0 (RCCL needs one
    GPU per rank, so gloo carries the launcher's barrier / id broadcast), query sharding with
    local_seq_package, both timed loops incl. the asynchronous result gather to rank 0, MAX-over-ranks
    timing, one JSON line from rank 0 whose aggregate value counts the reads of both ranks.  gather =
    "epa_comm": the default, the PRODUCT's gather (libepa_dev.so epa_comm_*, comm.hip) over the transport
    stand-in tests/fake_rccl.cpp; "torch": the torch.distributed harness of the same protocol."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, EPA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               WORLD_SIZE="2", LOCAL_RANK="0", EPA_BENCH_GATHER=gather)
    if gather == "epa_comm":
        import fake_rccl_util
        env = fake_rccl_util.env(env)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--chunk", "6000", "--tips", "64", "--width", "600", "--scaling", scaling, "--no-cpu-baseline"]
    # strong: a fixed job of 36 000 reads = 3 steps of 2 x 6000; weak: the 3 steps asked for, plus the
    # fixed-size `strong_cfg4` leg (48 000 reads = 4 steps per rank, cycling through the resident chunks)
    cmd += ["--reads", "36000", "--pool", "2"] if scaling == "strong" else ["--strong-reads", "48000", "--pool", "3"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in (1, 0)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert not [l for l in outs[0][0].splitlines() if l.startswith("{")]     # rank 1 prints no result line
    line = json.loads([l for l in outs[1][0].splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["steps"] == 3
    if scaling == "weak":
        sc = line["strong_cfg4"]
        assert sc["reads"] == 48000 and sc["steps_per_rank"] == 4 and sc["value"] > 0
        assert len(sc["per_rank_seconds"]) == 2
    else:
        assert line["strong_cfg4"] is None
    per_step = 12000          # every rank its own --chunk reads per step in both modes
    assert line["config"]["reads_per_step_whole_job"] == per_step
    assert line["config"]["reads_per_step_per_gpu"] == 6000
    assert len(line["per_rank_ms_per_step"]) == 2 and line["rccl_ranks"] == 0   # gloo here
    assert abs(line["value"] - 3 * per_step / (line["ms_per_step"] * 3e-3)) < 1e-3 * line["value"]
    assert line["pcie_inclusive"]["value"] > 0 and line["roofline"]["frac"] > 0
    g = line["gather"]
    if gather == "epa_comm":
        assert g["path"].startswith("epa_comm") and "stand-in" in g["transport"]
        assert g["rows_collected_rank0"] > 2 * 6000      # both loops' rows of both ranks reached rank 0
    else:
        assert g["path"].startswith("torch.distributed")


def test_bench_two_ranks_transport_that_hangs_falls_back_within_the_probe_timeout():
    """the first N > 1 run must not be able to lose the line.  The stand-in's fault knob
    (EPA_FAKE_RCCL_HANG_RECV: every ncclRecv waits for a message it never looks at -- "the transport cannot move
    data between these processes") makes the product's handshake (epa_comm_probe: a one-row gather + an all-reduce,
    every wait bounded) fail on every rank within the probe timeout; all ranks abort the communicator TOGETHER and
    run the torch.distributed harness; rank 0 prints a valid line that says so."""
    import socket
    import subprocess
    import sys
    import time
    import fake_rccl_util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, EPA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               WORLD_SIZE="2", LOCAL_RANK="0", EPA_BENCH_ONE_GPU="1", EPA_BENCH_PROBE_S="6", EPA_FAKE_RCCL_HANG_RECV="1")
    env = fake_rccl_util.env(env, timeout_s=300)     # the stand-in's own give-up is far away: the PROBE must end the wait
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--chunk", "3000", "--tips", "64", "--width", "600", "--no-cpu-baseline", "--strong-reads", "0", "--pool", "2"]
    t0 = time.time()
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in (1, 0)]
    outs = [p.communicate(timeout=300) for p in procs]
    took = time.time() - t0
    assert all(p.returncode == 0 for p in procs), outs
    line = json.loads([l for l in outs[1][0].splitlines() if l.startswith("{")][-1])
    g = line["gather"]
    assert g["path"].startswith("torch.distributed"), g
    assert g["fallback"] and "epa_comm gather unavailable" in g["fallback"][0] and "rank 0" in g["fallback"][0]
    assert "rank 1" in g["fallback"][0]              # BOTH ranks saw the handshake fail (rank 1 in the all-reduce)
    assert line["n_gpus"] == 2 and line["value"] > 0 and len(g["devices"]) == 2
    assert took < 200, took                          # two probes of 6 s + the run, not the stand-in's 300 s


def test_bench_two_ranks_line_names_devices_and_transport_library():
    """the healthy case of the same handshake: `gather.devices` = the PCI ids rank 0 RECEIVED through the product's
    gather (two ranks on the one GPU here: one distinct device), `gather.rccl_path` = the library the product bound"""
    import socket
    import subprocess
    import sys
    import fake_rccl_util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, EPA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               WORLD_SIZE="2", LOCAL_RANK="0", EPA_BENCH_ONE_GPU="1")
    env = fake_rccl_util.env(env)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--chunk", "3000", "--tips", "64", "--width", "600", "--no-cpu-baseline", "--strong-reads", "0", "--pool", "2"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in (1, 0)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    g = json.loads([l for l in outs[1][0].splitlines() if l.startswith("{")][-1])["gather"]
    assert g["path"].startswith("epa_comm") and g["fallback"] is None
    assert len(g["devices"]) == 2 and g["distinct_devices"] == 1 and g["devices"][0] == g["devices"][1]
    assert os.path.basename(g["rccl_path"]) == "libfake_rccl.so" and g["probe_seconds"] < 60


NCCL_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["EPA_ROOT"])
import torch
import torch.distributed as dist
from epa_ng_amd import parallel
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
rank, world = dist.get_rank(), dist.get_world_size()
ag = parallel.AsyncResultGather(dist, rows_cap=16, device=dev, host_copy=True)
want = []
for step in range(5):
    n = (40 if step == 2 and rank == world - 1 else 6) + step + rank     # one chunk overflows rows_cap: carry path
    p = torch.zeros((64, 2), dtype=torch.int32, device=dev)
    r = torch.zeros((64, 3), dtype=torch.float64, device=dev)
    p[:n, 0] = torch.arange(n, dtype=torch.int32, device=dev) + 100 * step
    p[:n, 1] = 1000 * rank + step
    r[:n, 0] = -(p[:n, 0].double() * 7 + p[:n, 1].double())
    r[:n, 1] = 0.25 * step
    ag.post(p, r, n, keep=True)
    want.append(n)
ag.finish(keep=True)
if rank == 0:
    for rk in range(world):
        rows = np.concatenate([parts[rk] for parts in ag.collected], 0)
        exp_n = [(40 if st == 2 and rk == world - 1 else 6) + st + rk for st in range(5)]
        assert rows.shape[0] == sum(exp_n), (rk, rows.shape, exp_n)
        b, s_, lnl, pen, _ = parallel.unpack_rows(rows)
        assert np.array_equal(b, np.concatenate([np.arange(n) + 100 * st for st, n in enumerate(exp_n)]))
        assert np.array_equal(s_, np.concatenate([np.full(n, 1000 * rk + st) for st, n in enumerate(exp_n)]))
        assert np.array_equal(lnl, -(b * 7.0 + s_))
    # the pinned host copy of the last retired gather holds the same valid rows as the device slot
    last = ag.counts[-1]
    slot = (ag.step - 1) % ag.depth
    for rk, k in enumerate(last):
        assert np.array_equal(ag.host[slot][rk, :k].numpy(), ag.collected[-1][rk])
    print("NCCL_GATHER_OK world=%d backend=%s" % (world, dist.get_backend()))
dist.destroy_process_group()
"""


def _run_ranks(script_path, n, extra_env=None, timeout=600):
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, EPA_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable] + script_path, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(n)]
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    return outs


def test_result_gather_on_the_rccl_backend(tmp_path):
    """AsyncResultGather over backend "nccl" (= RCCL): device send / receive slots, the gather on
    RCCL's stream, the side-stream host copy on rank 0, the carry path.  One rank per visible GPU (up
    to 8); on the 1-GPU box that is a 1-rank group -- no xGMI traffic, but every RCCL call of the
    exchange runs (device tensors, async gather, stream-level wait)."""
    n = max(1, min(8, epa.device_count()))
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    outs = _run_ranks([str(script)], n)
    assert "NCCL_GATHER_OK world=%d backend=nccl" % n in outs[0][0], outs[0]


def test_bench_over_rccl_on_all_visible_gpus():
    """bench.py's N > 1 branch on RCCL, one rank per GPU: runs wherever at least 2 GPUs are visible
    (the driver's 8-GPU box), skipped on the 1-GPU box."""
    n = min(8, epa.device_count())
    if n < 2:
        pytest.skip("needs >= 2 GPUs (RCCL wants one device per rank)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = _run_ranks([os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
                       "--chunk", "20000", "--strong-reads", str(20000 * n * 4), "--pool", "3",
                       "--no-cpu-baseline"], n, timeout=900)
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == n and line["rccl_ranks"] == n
    assert line["gather"]["path"].startswith("epa_comm") and line["gather"]["transport"] == "RCCL"
    assert line["config"]["reads_per_step_whole_job"] == 20000 * n
    assert line["strong_cfg4"]["steps_per_rank"] == 4 and line["strong_cfg4"]["value"] > 0
