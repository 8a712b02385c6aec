"""20-state (amino-acid) GPU parity: BASELINE configs[2] shape at test size + the golden AA case."""
import numpy as np
import pytest

import epa_ng_amd as epa
from epa_ng_amd import hostlib, synth
from golden_util import load_case
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu


def test_golden_aa_preplace_and_thorough():
    g = load_case("aa8_protgtr_g4")
    labels = [a for a, _ in g["msa"]]
    seqs = [b for _, b in g["msa"]]
    ref = hostlib.Reference(g["newick"], labels, seqs, states=20, subst=g["subst"], freqs=g["freqs"],
                            rates=g["gamma_rates"])
    ev = ref.evaluator()
    qs = [q["seq"] for q in g["queries"]]
    codes, wb, ws = epa.encode_queries(20, qs)
    lnl = ev.preplace(codes, wb, ws)
    assert np.max(np.abs(lnl - np.array(g["preplace"]))) < 1e-6
    pairs = np.zeros(ref.B * len(qs), epa.PAIR_DTYPE)
    pairs["branch_id"] = np.repeat(np.arange(ref.B), len(qs))
    pairs["seq_id"] = np.tile(np.arange(len(qs)), ref.B)
    res = ev.thorough(pairs, codes, wb, ws)
    nrev = 0
    for i, p in enumerate(pairs):
        gold = g["thorough"][p["seq_id"]][p["branch_id"]]
        assert abs(res["lnl"][i] - gold["lnl"]) < 1e-6, (p, res[i], gold)
        assert abs(res["distal_length"][i] - gold["distal"]) < 1e-6
        assert abs(res["pendant_length"][i] - gold["pendant"]) < 1e-6 * max(1.0, gold["pendant"])
        nrev += gold["reverted"]
    assert ev.last_stats["reverts"] == nrev


def test_synthetic_aa_vs_oracle():
    w = synth.aa_workload(48, 300, 250, 100, (71, 72, 73))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=20, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    o = Oracle(w["newick"], w["labels"], w["seqs"], 20, w["subst"], w["freqs"], w["rates"])
    codes, wb, ws = epa.encode_queries(20, w["reads"])
    pairs, res = ev.place_chunk(codes, wb, ws)
    lnl = ev.preplace(codes, wb, ws)
    assert np.max(np.abs(lnl - o.preplace(w["reads"]))) < 1e-6
    hb, hs = hostlib.heuristic(lnl, "dynamic", 0.99999)
    assert sorted(zip(hb.tolist(), hs.tolist())) == sorted(zip(pairs["branch_id"].tolist(), pairs["seq_id"].tolist()))
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], w["reads"])
    assert np.max(np.abs(res["lnl"] - tl)) < 1e-6
    assert np.max(np.abs(res["distal_length"] - td)) < 1e-6
    assert ev.last_stats["rounds"] == o.last_stats["rounds"]


@pytest.mark.parametrize("read_len,seeds", [(170, (81, 82, 83)), (37, (84, 85, 86))])
def test_synthetic_aa_window_lengths(read_len, seeds):
    """170-column windows: sumtable slab in HBM scratch, three 64-site passes (the LDS-resident
    slab covers windows up to 102 columns); 37-column windows: a single partly filled pass."""
    w = synth.aa_workload(24, 260, 60, read_len, seeds)
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=20, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    o = Oracle(w["newick"], w["labels"], w["seqs"], 20, w["subst"], w["freqs"], w["rates"])
    codes, wb, ws = epa.encode_queries(20, w["reads"])
    pairs, res = ev.place_chunk(codes, wb, ws)
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], w["reads"])
    assert np.max(np.abs(res["lnl"] - tl)) < 1e-6
    assert np.max(np.abs(res["distal_length"] - td)) < 1e-6
    assert np.max(np.abs(res["pendant_length"] - tp) / np.maximum(1.0, tp)) < 1e-6
    assert ev.last_stats["rounds"] == o.last_stats["rounds"]
    assert ev.last_stats["reverts"] == o.last_stats["reverts"]


def test_aa_preplace_site_path_bitwise_equals_generic(monkeypatch):
    """20-state fast path (k_preplace_sites: precomputed LDS offsets, 1024-query groups) vs the
    generic gather kernel vs the oracle: every window length mod 4, windows longer than one
    128-site chunk (accumulating variant), ambiguity codes (B, Z, X, gap) inside the window; then
    the single-chunk variant through place_chunk with a declared max_span."""
    w = synth.aa_workload(24, 520, 8, 60, (91, 92, 93))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=20, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    ev = ref.evaluator()
    o = Oracle(w["newick"], w["labels"], w["seqs"], 20, w["subst"], w["freqs"], w["rates"])
    rng = np.random.RandomState(5)
    W = 520
    aa = list("ARNDCQEGHILKMFPSTWYV")

    def make(spans, n):
        reads = []
        for i in range(n):
            span = int(rng.choice(spans))
            begin = int(rng.randint(0, W - span + 1))
            body = rng.choice(aa, span)
            if i % 3 == 0:
                k = rng.randint(0, span, max(1, span // 10))
                body[k] = rng.choice(list("BZX-"), len(k))
                body[0] = "A"; body[-1] = "C"
            reads.append("-" * begin + "".join(body) + "-" * (W - begin - span))
        return reads

    reads = make([1, 2, 3, 4, 5, 7, 37, 126, 127, 128, 129, 130, 131, 255, 257, 390, 519], 1300)
    for compact in (False, True):
        codes, wb, ws = epa.encode_queries(20, reads, compact=compact)
        assert len(set(ws % 4)) == 4 and ws.max() > 384
        fast = ev.preplace(codes, wb, ws)
        ev.set_option("preplace_generic", 1)
        generic = ev.preplace(codes, wb, ws)
        ev.set_option("preplace_generic", 0)
        assert np.array_equal(fast, generic)  # same association order, bit for bit
        assert np.max(np.abs(fast - o.preplace(reads))) < 1e-6
    short = make([1, 2, 3, 5, 6, 7, 64, 99, 100, 101, 102], 1300)
    codes, wb, ws = epa.encode_queries(20, short, compact=True)
    pf, rf = ev.place_chunk(codes, wb, ws, max_span=int(ws.max()))
    ev.set_option("preplace_generic", 1)
    pg, rg = ev.place_chunk(codes, wb, ws, max_span=int(ws.max()))
    ev.set_option("preplace_generic", 0)
    assert np.array_equal(pf, pg) and np.array_equal(rf, rg)
    lnl = ev.preplace(codes, wb, ws)
    hb, hs = hostlib.heuristic(lnl, "dynamic", 0.99999)
    assert sorted(zip(hb.tolist(), hs.tolist())) == sorted(zip(pf["branch_id"].tolist(), pf["seq_id"].tolist()))
