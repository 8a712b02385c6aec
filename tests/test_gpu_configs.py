"""GPU tests at the sizes of BASELINE.json's configs that are not the bench line: cfg3 (2000-tip
20-state reference, 50k queries) and the cfg5 shape (4000-tip DNA reference, --no-heur), plus the
switches and reference properties that had no device coverage (quirk D4, window == full width).

The oracle is the checker on read samples it finishes in seconds; at full size the tests fall back
on size-independent properties (thorough lnL >= preplacement lnL of the same pair, sanity ranges
of test/src/Tiny_Tree.cpp:39-48, candidates == host heuristic, bit-identical repeat).
Tolerances: per-branch lnL |delta| <= 1e-6 (BASELINE.json north_star); lengths 1e-6."""
import numpy as np
import pytest

import epa_ng_amd as epa
from epa_ng_amd import hostlib, synth
from golden_util import RAX8_PROT, parse_descriptor
from oracle_lib import Oracle, gamma_rates

pytestmark = pytest.mark.gpu

LNL_TOL = 1e-6


def _assert_thorough_parity(res, tl, tp, td):
    assert np.max(np.abs(res["lnl"] - tl)) < LNL_TOL
    assert np.max(np.abs(res["pendant_length"] - tp) / np.maximum(1.0, tp)) < 1e-6
    assert np.max(np.abs(res["distal_length"] - td)) < 1e-6


def test_cfg3_aa_2000_tips_oracle_sample_and_full_size_properties():
    """BASELINE configs[2] as SURVEY.md section 8d defines it: 2000 tips (B = 3997), W = 500, the
    PROTGTR+FU+G4 literal of test/src/parse_model.cpp:26-47, 50 000 queries of 100 residues,
    seeds 11/12/13.  Oracle parity on the first 2000 reads, properties on all 50 000."""
    subst, freqs, alpha = parse_descriptor(RAX8_PROT)
    rates = synth.gamma_rates(alpha)
    root = synth.random_tree(2000, 11)
    labels, seqs = synth.simulate_msa(root, 500, subst, freqs, rates, 12)
    nw = synth.newick(root)
    reads, _ = synth.make_reads(seqs, 50000, 100, 0.03, 13, states=20)
    ref = hostlib.Reference(nw, labels, seqs, states=20, subst=subst, freqs=freqs, rates=rates)
    assert ref.B == 3997 and ref.W == 500
    ev = ref.evaluator()
    ns = 2000
    hostlib.configure_threads()
    o = Oracle(nw, labels, seqs, 20, subst, freqs, rates)
    assert abs(ev.tree_logl(0) - o.tree_lnl(0)) < 1e-6 * abs(o.tree_lnl(0)) * 1e-3 + 1e-6
    sample = reads[:ns]
    codes, wb, ws = epa.encode_queries(20, sample, compact=True)
    lnl = ev.preplace(codes, wb, ws)
    assert np.max(np.abs(lnl - o.preplace(sample))) < LNL_TOL
    pairs = ev.select(lnl, ns, 0.99999)
    hb, hs = hostlib.heuristic(lnl, "dynamic", 0.99999)
    assert sorted(zip(hb.tolist(), hs.tolist())) == sorted(zip(pairs["branch_id"].tolist(), pairs["seq_id"].tolist()))
    res = ev.thorough(pairs, codes, wb, ws)
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], sample)
    _assert_thorough_parity(res, tl, tp, td)
    assert ev.last_stats["rounds"] == o.last_stats["rounds"]
    assert ev.last_stats["reverts"] == o.last_stats["reverts"]
    del o
    # ---- full size, fused chunk (the table is Q x 3997 doubles = 1.6 GB in HBM)
    codes, wb, ws = epa.encode_queries(20, reads, compact=True)
    Q = len(reads)
    pairs_f, res_f = ev.place_chunk(codes, wb, ws, max_span=100)
    assert set(np.unique(pairs_f["seq_id"])) == set(range(Q))
    assert np.all(np.isfinite(res_f["lnl"]))
    assert np.all(np.diff(pairs_f["branch_id"].astype(np.int64)) >= 0)
    bl = np.array([ref.branch(int(b))["length"] for b in range(ref.B)])
    assert np.all(res_f["pendant_length"] >= 1e-4 - 1e-12) and np.all(res_f["pendant_length"] <= 100.0)
    assert np.all(res_f["distal_length"] > 0) and np.all(res_f["distal_length"] < bl[pairs_f["branch_id"]])
    m = pairs_f["seq_id"] < ns
    # the sample's part of the full-size run == the separate calls above, pair for pair
    key_f = pairs_f["branch_id"][m].astype(np.int64) * Q + pairs_f["seq_id"][m]
    key_s = pairs["branch_id"].astype(np.int64) * Q + pairs["seq_id"]
    assert np.array_equal(np.sort(key_f), np.sort(key_s))
    of, os_ = np.argsort(key_f), np.argsort(key_s)
    assert np.array_equal(res_f["lnl"][m][of], res["lnl"][os_])
    pre = lnl[pairs_f["seq_id"][m], pairs_f["branch_id"][m]]
    assert np.all(res_f["lnl"][m] >= pre - 1e-7)
    p2, r2 = ev.place_chunk(codes, wb, ws, max_span=100)
    assert np.array_equal(p2, pairs_f) and np.array_equal(r2["lnl"], res_f["lnl"])


def _numpy_lwr_filter(lnl_row, min_lwr=0.01, mn=1, mx=7):
    """compute_and_set_lwr + discard_by_support_threshold (src/set_manipulators.cpp:43-69,131-163)"""
    lw = np.exp(lnl_row - lnl_row.max())
    lw /= lw.sum()
    order = np.lexsort((np.arange(len(lw)), -lnl_row))
    keep = []
    for b in order:
        if len(keep) >= mx:
            break
        if not lw[b] > min_lwr and len(keep) >= mn:
            break
        keep.append(b)
    return np.array(keep), lw


def test_cfg5_shape_4000_tips_default_heuristic_and_no_heur_vs_oracle():
    """BASELINE configs[4] shape: 4000 tips (B = 7997), W = 1500, 150 bp reads.  Default dynamic
    heuristic on 1500 reads against the oracle (preplacement table, candidates, thorough), then
    --no-heur (epa_dev_place_all: every one of the 7997 branches gets the NR optimisation, LWR and
    filter on the device) on 6 reads against the oracle's thorough placement of all 47 982 pairs
    and the numpy restatement of compute_and_set_lwr + filter."""
    w = synth.dna_workload(4000, 1500, 1500, 150, (21, 22, 23))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=4, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    assert ref.B == 7997
    ev = ref.evaluator()
    hostlib.configure_threads()
    o = Oracle(w["newick"], w["labels"], w["seqs"], 4, w["subst"], w["freqs"], w["rates"])
    reads = w["reads"]
    codes, wb, ws = epa.encode_queries(4, reads, compact=True)
    lnl = ev.preplace(codes, wb, ws)
    assert np.max(np.abs(lnl - o.preplace(reads))) < LNL_TOL
    pairs, res = ev.place_chunk(codes, wb, ws, max_span=150)
    hb, hs = hostlib.heuristic(lnl, "dynamic", 0.99999)
    assert sorted(zip(hb.tolist(), hs.tolist())) == sorted(zip(pairs["branch_id"].tolist(), pairs["seq_id"].tolist()))
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], reads)
    _assert_thorough_parity(res, tl, tp, td)
    assert ev.last_stats["rounds"] == o.last_stats["rounds"]
    # ---- per-rate scalers: what the reference (and epa-ng-amd) turn on above 2000 tips
    # (src/io/file_io.cpp:211-214); on ordinary data the numbers equal the per-site ones
    o_rs = Oracle(w["newick"], w["labels"], w["seqs"], 4, w["subst"], w["freqs"], w["rates"], rate_scalers=True)
    ev_rs = ref.evaluator(rate_scalers=True)
    nrs = 300
    c3, b3, s3 = epa.encode_queries(4, reads[:nrs], compact=True)
    lnl_rs = ev_rs.preplace(c3, b3, s3)
    assert np.max(np.abs(lnl_rs - o_rs.preplace(reads[:nrs]))) < LNL_TOL
    assert np.max(np.abs(lnl_rs - lnl[:nrs])) < 1e-7
    p_rs, r_rs = ev_rs.place_chunk(c3, b3, s3, max_span=150)
    tl_rs, tp_rs, td_rs = o_rs.thorough(p_rs["branch_id"], p_rs["seq_id"], reads[:nrs])
    _assert_thorough_parity(r_rs, tl_rs, tp_rs, td_rs)
    assert ev_rs.last_stats["rounds"] == o_rs.last_stats["rounds"]
    del o_rs, ev_rs
    # ---- --no-heur on a handful of reads
    nq = 6
    sub = reads[:nq]
    c2, b2, s2 = epa.encode_queries(4, sub, compact=True)
    got = ev.place_all(c2, b2, s2, min_lwr=0.01, filter_min=1, filter_max=7, max_span=150)
    assert ev.last_stats["pairs"] == nq * ref.B
    pb = np.repeat(np.arange(ref.B), nq)
    ps = np.tile(np.arange(nq), ref.B)
    al, ap, ad = o.thorough(pb, ps, sub)
    assert ev.last_stats["rounds"] == o.last_stats["rounds"]
    al, ap, ad = (x.reshape(ref.B, nq) for x in (al, ap, ad))
    for q in range(nq):
        keep, lw = _numpy_lwr_filter(al[:, q])
        bid, l, pen, dis, lwr = got[q]
        assert np.array_equal(bid, keep)
        assert np.max(np.abs(l - al[keep, q])) < LNL_TOL
        assert np.max(np.abs(pen - ap[keep, q]) / np.maximum(1.0, ap[keep, q])) < 1e-6
        assert np.max(np.abs(dis - ad[keep, q])) < 1e-6
        assert np.max(np.abs(lwr - lw[keep])) < 1e-6


def test_aa_x_as_n_quirk_on_device():
    """Quirk D4: the reference preplaces an AA 'X' in the 'N' = asparagine
    column (src/core/Lookup_Store.hpp:63-66).  epa_ref_desc.aa_x_as_n = 1 reproduces it; the
    default scores 'X' as "any".  Both against the oracle (orc_set_aa_x_quirk); the thorough
    placement uses pll_map_aa in either mode, i.e. 'X' = any."""
    w = synth.aa_workload(24, 120, 40, 60, (31, 32, 33))
    reads = []
    rng = np.random.RandomState(5)
    for r in w["reads"]:
        r = list(r)
        idx = [k for k, ch in enumerate(r) if ch != "-"]
        for k in rng.choice(idx[1:-1], 6, replace=False):
            r[k] = "X"
        reads.append("".join(r))
    ref = hostlib.Reference(w["newick"], w["labels"], w["seqs"], states=20, subst=w["subst"],
                            freqs=w["freqs"], rates=w["rates"])
    o = Oracle(w["newick"], w["labels"], w["seqs"], 20, w["subst"], w["freqs"], w["rates"])
    tables = {}
    for quirk in (False, True):
        ev = ref.evaluator(aa_x_as_n=quirk)
        o.set_aa_x_quirk(quirk)
        for compact in (False, True):
            codes, wb, ws = epa.encode_queries(20, reads, aa_x_as_n=quirk, compact=compact)
            lnl = ev.preplace(codes, wb, ws)
            assert np.max(np.abs(lnl - o.preplace(reads))) < LNL_TOL
        tables[quirk] = lnl
        pairs = ev.select(lnl, len(reads), 0.99999)
        res = ev.thorough(pairs, codes, wb, ws)
        tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], reads)
        _assert_thorough_parity(res, tl, tp, td)
    # the switch is not a no-op: 'X' as asparagine scores strictly lower than 'X' as any state
    assert np.all(tables[True] < tables[False])


def test_windowed_equals_full_width_on_device():
    """The reference's shift_partition_focus_logtest (test/src/pll_util.cpp:270-335) through the
    device: all tips are gap outside columns [2, 6), so scoring a query over its window (premasking
    on) and over the full width (off) must give the same numbers -- lookup sums, thorough lnL and
    branch lengths."""
    freqs = [0.17, 0.19, 0.25, 0.39]
    labels = ["a", "t", "g"]
    seqs = ["--AAAA----", "--TTTT----", "--GGGG----"]
    nw = "(a:0.123,t:0.123,g:0.123);"
    rates = gamma_rates(1.0)
    ref = hostlib.Reference(nw, labels, seqs, states=4, subst=[1.0] * 6, freqs=freqs, rates=rates)
    o = Oracle(nw, labels, seqs, 4, [1.0] * 6, freqs, rates)
    ev = ref.evaluator()
    for b in range(ref.B):
        assert abs(ev.tree_logl(b) - o.tree_lnl(b)) < 1e-10
    qs = ["--ACGT----", "--TTTT----", "--GNRA----"]
    out = {}
    for premask in (True, False):
        codes, wb, ws = epa.encode_queries(4, qs, premasking=premask)
        lnl = ev.preplace(codes, wb, ws)
        assert np.max(np.abs(lnl - o.preplace(qs, premask=premask))) < 1e-10
        pairs = np.zeros(ref.B * len(qs), epa.PAIR_DTYPE)
        pairs["branch_id"] = np.repeat(np.arange(ref.B), len(qs))
        pairs["seq_id"] = np.tile(np.arange(len(qs)), ref.B)
        res = ev.thorough(pairs, codes, wb, ws)
        tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], qs, premask=premask)
        _assert_thorough_parity(res, tl, tp, td)
        out[premask] = (lnl, res)
    assert np.max(np.abs(out[True][0] - out[False][0])) < 1e-12
    assert np.max(np.abs(out[True][1]["lnl"] - out[False][1]["lnl"])) < 1e-12
    assert np.max(np.abs(out[True][1]["pendant_length"] - out[False][1]["pendant_length"])) < 1e-9
