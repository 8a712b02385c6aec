"""Pins the CPU oracle (oracle/epa_oracle.c) against (a) the independent brute-force golden
vectors of tests/gen_golden.py and (b) the reference's own literal vectors for this path
(edge numbering, test/src/pll_util.cpp:134-143).  CPU only."""
import numpy as np
import pytest

from golden_util import CASES, load_case
from oracle_lib import Oracle, gamma_rates, lib


def make_oracle(g):
    labels = [a for a, _ in g["msa"]]
    seqs = [b for _, b in g["msa"]]
    return Oracle(g["newick"], labels, seqs, g["states"], g["subst"], g["freqs"],
                  g["gamma_rates"])


@pytest.mark.parametrize("case", CASES)
def test_gamma_rates(case):
    g = load_case(case)
    r = gamma_rates(g["alpha"], 4)
    assert np.allclose(r, g["gamma_rates"], rtol=1e-11, atol=0)
    assert abs(r.mean() - 1.0) < 1e-12


def test_gamma_alpha1_known_values():
    # closed form for alpha = 1 (exponential): SURVEY.md section 8c quotes these
    r = gamma_rates(1.0, 4)
    assert np.allclose(r, [0.13695378, 0.47675186, 1.0, 2.38629436], atol=5e-9)


def test_edge_numbering_reference_literal():
    # literal from the reference's test/src/pll_util.cpp:134-137 (ref.tre with all lengths 1.0)
    g = load_case("dna8_gtr_g_default")
    import re
    nw = re.sub(r":[0-9.]+", ":1.0", g["newick"])
    labels = [a for a, _ in g["msa"]]
    seqs = [b for _, b in g["msa"]]
    o = Oracle(nw, labels, seqs, 4, g["subst"], g["freqs"], g["gamma_rates"])
    assert o.numbered_newick(2) == (
        "(A:1.00{0},(B:1.00{1},(C:1.00{2},(D:1.00{3},(E:1.00{4},(F:1.00{5},G:1.00{6}):"
        "1.00{7}):1.00{8}):1.00{9}):1.00{10}):1.00{11},H:1.00{12});")


@pytest.mark.parametrize("case", CASES)
def test_tree_lnl_all_edges(case):
    # invariant (i) of the reference's test/src/epa_pll_util.cpp:82-121 + golden value
    g = load_case(case)
    o = make_oracle(g)
    assert o.B == len(g["branch_lengths"])
    assert o.numbered_newick(2) == g["numbered_newick_p2"]
    for b in range(o.B):
        assert abs(o.tree_lnl(b) - g["tree_lnl"]) < 1e-8
        assert abs(o.branch_info(b)[0] - g["branch_lengths"][b]) < 1e-15


@pytest.mark.parametrize("case", CASES)
def test_preplace_matches_bruteforce(case):
    g = load_case(case)
    o = make_oracle(g)
    qs = [q["seq"] for q in g["queries"]]
    got = o.preplace(qs)
    exp = np.array(g["preplace"])
    assert got.shape == exp.shape
    assert np.max(np.abs(got - exp)) < 1e-8
    # lookup-sum == direct edge lnL (SURVEY 8c invariant)
    for qi in (0, len(qs) - 1):
        for b in (0, o.B // 2, o.B - 1):
            assert abs(o.direct_default_lnl(b, qs[qi]) - got[qi, b]) < 1e-9


@pytest.mark.parametrize("case", CASES)
def test_thorough_matches_bruteforce(case):
    g = load_case(case)
    o = make_oracle(g)
    qs = [q["seq"] for q in g["queries"]]
    pb, ps = [], []
    for b in range(o.B):
        for qi in range(len(qs)):
            pb.append(b)
            ps.append(qi)
    lnl, pen, dis = o.thorough(pb, ps, qs)
    nrev = 0
    for i, (b, qi) in enumerate(zip(pb, ps)):
        e = g["thorough"][qi][b]
        assert abs(lnl[i] - e["lnl"]) < 1e-7, (b, qi, lnl[i], e)
        # lengths: relative (flat optima at pendant ~ 40 are only tol-determined)
        assert abs(pen[i] - e["pendant"]) < 1e-7 * max(1.0, e["pendant"]), (b, qi, pen[i], e)
        assert abs(dis[i] - e["distal"]) < 1e-7, (b, qi, dis[i], e)
        nrev += e["reverted"]
    assert o.last_stats["reverts"] == nrev
    assert o.last_stats["rounds"] == sum(e["rounds"] for row in g["thorough"] for e in row)
    # sanity ranges of the reference's test/src/Tiny_Tree.cpp:39-48
    for i, b in enumerate(pb):
        assert np.isfinite(lnl[i]) and lnl[i] != 0.0
        assert 0.0 < dis[i] < g["branch_lengths"][b]
        assert pen[i] > 0.0


def test_derivatives_match_finite_differences():
    g = load_case("dna8_gtr_fu_g4")
    o = make_oracle(g)
    q = g["queries"][0]["seq"]
    t, h = 0.07, 1e-5
    f, df, l0 = o.pendant_derivatives(3, q, t)
    _, _, lp = o.pendant_derivatives(3, q, t + h)
    _, _, lm = o.pendant_derivatives(3, q, t - h)
    assert abs(f - (-(lp - lm) / (2 * h))) < 1e-5 * max(1.0, abs(f))
    assert abs(df - (-(lp - 2 * l0 + lm) / (h * h))) < 2e-3 * max(1.0, abs(df))


def test_windowed_equals_full_when_outside_is_gap():
    """The reference's shift_partition_focus_logtest (test/src/pll_util.cpp:270-335) restated on
    its own numbers: tips "--AAAA----" / "--TTTT----" / "--GGGG----", freqs {.17,.19,.25,.39},
    all exchangeabilities 1, Gamma(1.0) x 4, branch lengths 0.123.  Every column outside [2, 6) is
    all-gap at every tip, so it contributes log(sum_i pi_i) = 0 and the lnL over the window must
    EQUAL the lnL over the full width (EXPECT_DOUBLE_EQ = 4 ulps), for the edge lnL of the tree and
    for a query scored on every branch (lookup sums and direct evaluation)."""
    from oracle_lib import Oracle, gamma_rates
    freqs = [0.17, 0.19, 0.25, 0.39]
    labels = ["a", "t", "g"]
    seqs = ["--AAAA----", "--TTTT----", "--GGGG----"]
    inner = [s_[2:6] for s_ in seqs]
    nw = "(a:0.123,t:0.123,g:0.123);"
    rates = gamma_rates(1.0)
    full = Oracle(nw, labels, seqs, 4, [1.0] * 6, freqs, rates)
    win = Oracle(nw, labels, inner, 4, [1.0] * 6, freqs, rates)

    def double_eq(x, y):     # gtest's EXPECT_DOUBLE_EQ: within 4 ulps
        return abs(x - y) <= 4 * np.spacing(max(abs(x), abs(y)))
    for b in range(full.B):
        assert double_eq(full.tree_lnl(b), win.tree_lnl(b))
    q = "--ACGT----"
    ranged = full.preplace([q], premask=True)
    whole = full.preplace([q], premask=False)
    cut = win.preplace([q[2:6]], premask=False)
    for b in range(full.B):
        assert double_eq(ranged[0, b], whole[0, b]) and double_eq(ranged[0, b], cut[0, b])
        assert double_eq(full.direct_default_lnl(b, q, True), full.direct_default_lnl(b, q, False))
        assert abs(full.direct_default_lnl(b, q, True) - ranged[0, b]) < 1e-10   # lookup == direct
    # and on the bundled data: a windowed query scored over the full width differs from its
    # windowed score exactly by the reference-only columns, which are not all-gap there
    g = load_case("dna8_gtr_g_default")
    o = make_oracle(g)
    qq = g["queries"][2]["seq"]           # window 100..250
    assert np.all(o.preplace([qq], premask=False) < o.preplace([qq], premask=True))


def test_error_codes():
    g = load_case("dna8_gtr_g_default")
    o = make_oracle(g)
    with pytest.raises(RuntimeError):
        o.preplace(["-" * o.W])
    with pytest.raises(RuntimeError):
        o.preplace(["!" + "A" * (o.W - 1)])
    assert lib().orc_char_column(4, b"u", 0) == 1 and lib().orc_char_column(4, b"x", 0) == 0
    assert lib().orc_char_column(20, b"X", 1) == 11 and lib().orc_char_column(20, b"X", 0) == 21


def test_prop_invariant_sites_matches_bruteforce():
    """+I (GTR+FU+I{0.2}+G4 on the reference's data): the oracle's restatement of libpll's
    invariant-site handling (rates / (1-p) in P(t), (1-p) L + p pi_inv, invariant sites from the
    reference tips only) against the independent brute force of tests/gen_golden.py."""
    g = load_case("dna8_gtr_fu_i_g4")
    assert g["pinv"] == 0.2 and sum(1 for v in g["invariant_state"] if v >= 0) > 50
    labels = [a for a, _ in g["msa"]]
    seqs = [b for _, b in g["msa"]]
    o = Oracle(g["newick"], labels, seqs, 4, g["subst"], g["freqs"], g["gamma_rates"], pinv=g["pinv"])
    for b in range(o.B):
        assert abs(o.tree_lnl(b) - g["tree_lnl"]) < 1e-8
    qs = [q["seq"] for q in g["queries"]]
    assert np.max(np.abs(o.preplace(qs) - np.array(g["preplace"]))) < 1e-8
    pb = [b for b in range(o.B) for _ in qs]
    ps = [qi for _ in range(o.B) for qi in range(len(qs))]
    lnl, pen, dis = o.thorough(pb, ps, qs)
    for i, (b, qi) in enumerate(zip(pb, ps)):
        e = g["thorough"][qi][b]
        assert abs(lnl[i] - e["lnl"]) < 1e-7, (b, qi, lnl[i], e)
        assert abs(pen[i] - e["pendant"]) < 1e-7 * max(1.0, e["pendant"])
        assert abs(dis[i] - e["distal"]) < 1e-7
    assert o.last_stats["rounds"] == sum(e["rounds"] for row in g["thorough"] for e in row)
    assert o.last_stats["reverts"] == sum(e["reverted"] for row in g["thorough"] for e in row)


# ---- round 3: the evaluator apart from the optimiser (orc_score_at) and the oracle's rounding siblings

def _sweep_oracle(seed):
    import sweep_util as su
    c = su.make_case(seed)
    o = Oracle(c["newick"], c["labels"], c["seqs"], c["states"], c["subst"], c["freqs"], c["rates"], pinv=c["pinv"])
    B, Q = o.B, c["nreads"]
    return c, o, np.repeat(np.arange(B), Q), np.tile(np.arange(Q), B)


def test_score_at_returned_lengths_is_the_returned_lnl():
    """Tiny_Tree::place returns the edge lnL of the triplet at the lengths it returns (also after the
    'worse -> restore lengths, keep old lnL' exit, optimize.cpp:224-232): orc_score_at at the
    optimiser's output reproduces its lnL, and at the starting lengths it is the preplacement value."""
    c, o, pb, ps = _sweep_oracle(7)
    tl, tp, td = o.thorough(pb, ps, c["reads"])
    at = o.score_at(pb, ps, c["reads"], tp, td)
    assert np.max(np.abs(at - tl)) < 1e-9
    assert o.last_stats["reverts"] > 0            # the restore exit is exercised
    orig = np.array([o.branch_info(b)[0] for b in range(o.B)])
    pre = o.preplace(c["reads"])
    at0 = o.score_at(pb, ps, c["reads"], np.full(len(pb), -np.log(0.9)), orig[pb] / 2)
    assert np.max(np.abs(at0 - pre[ps, pb])) < 1e-8


def test_trace_rows_follow_the_optimiser():
    c, o, pb, ps = _sweep_oracle(7)
    rows, l, p, d = o.trace_pair(3, c["reads"][0])
    tl, tp, td = o.thorough([3], [0], c["reads"])
    assert (l, p, d) == (tl[0], tp[0], td[0])
    kinds = rows[:, 0]
    assert kinds[0] == 1 and kinds[-1] == 3 and set(kinds) <= {1.0, 2.0, 3.0}
    assert (kinds == 3).sum() == o.last_stats["rounds"]
    assert (kinds != 3).sum() == o.last_stats["newton_evals"]
    assert rows[0, 1] == -np.log(0.9)             # first row: the pendant solve starts at -ln 0.9


def test_rounding_siblings_agree_where_the_optimum_is_well_conditioned():
    """a sibling (terms of the dot products moved by a few ulp, stationary eigenvalue exactly 0) walks
    the same path on an ordinary configuration: same lengths, lnL within 1e-9"""
    c, o, pb, ps = _sweep_oracle(3)
    tl, tp, td = o.thorough(pb, ps, c["reads"])
    import sweep_util as su
    for v in (1, 2, 0x801, 0x1003, (4 << 16) | 5):
        o.set_rounding_variant(v)
        l2, p2, d2 = o.thorough(pb, ps, c["reads"])
        same = ~su.lengths_differ(p2, d2, tp, td)
        assert same.mean() > 0.99
        assert np.max(np.abs(l2 - tl)[same]) < 1e-9
    o.set_rounding_variant(0)
    assert np.array_equal(o.thorough(pb, ps, c["reads"])[0], tl)


def test_oracle_disagrees_with_itself_on_the_known_bimodal_seeds():
    """Seeds 1070 and 2692 of the randomised sweep (reads of 1 and 2 sites): the pendant likelihood
    has two local optima and the first bisection lands where f is 0+ and f' is rounding noise.  The
    ORACLE ALONE, evaluated by its faithfully rounded siblings, ends in either of them -- with the
    lnL gaps round 2 saw between device and oracle (0.766 and 1.95) -- while the evaluator
    (score_at at whatever lengths came out) stays exact.  This is the CPU-side half of the rule the GPU
    sweep applies to diverging pairs (tests/test_gpu_parity.py)."""
    import sweep_util as su
    for seed, gap in ((1070, 0.766), (2692, 1.95)):
        c, o, pb, ps = _sweep_oracle(seed)
        tl, tp, td = o.thorough(pb, ps, c["reads"])
        worst = 0.0
        for v in (1, 2, 3, 4, 5, 6, 0x801, 0x802):
            o.set_rounding_variant(v)
            l2, p2, d2 = o.thorough(pb, ps, c["reads"])
            o.set_rounding_variant(0)
            at = o.score_at(pb, ps, c["reads"], p2, d2)
            assert np.max(np.abs(at - l2)) < 1e-9      # the evaluator does not care which optimum
            fl = su.lengths_differ(p2, d2, tp, td)
            if fl.any():
                worst = max(worst, float(np.abs(l2 - tl)[fl].max()))
        assert abs(worst - gap) < 0.01, (seed, worst)


def test_flat_pair_rule_names_the_decision_where_a_sibling_leaves_the_oracles_path():
    """The optimiser-path rule of the GPU sweep (sweep_util.reproduce_flat_pairs) exercised on CPU with a
    rounded sibling of the oracle standing in for the device: on the bimodal seeds every pair whose
    lengths differ is reproduced with amplitude 0 (the stationary-eigenvalue variant), the two traces
    agree up to a point and part at one of the solver's three decisions; and a "device" that no sibling
    reproduces (lengths moved by hand) is reported as unreproduced."""
    import sweep_util as su
    for seed in (1070, 2692):
        c, o, pb, ps = _sweep_oracle(seed)
        tl, tp, td = o.thorough(pb, ps, c["reads"])
        o.set_rounding_variant(0x801)
        l2, p2, d2 = o.thorough(pb, ps, c["reads"])
        o.set_rounding_variant(0)
        res = {"lnl": l2, "pendant_length": p2, "distal_length": d2}
        flat = su.lengths_differ(p2, d2, tp, td)
        assert flat.any()
        rep = su.reproduce_flat_pairs(o, c["reads"], pb, ps, res, flat)
        assert rep["flat_reproduced"] == rep["flat_pairs"] == int(flat.sum()) and not rep["unreproduced"]
        assert rep["max_amplitude_log2_ulp"] == 0 and rep["stationary_mode"] == rep["flat_pairs"]
        assert sum(rep["decisions"].values()) == rep["flat_pairs"] and set(rep["decisions"]) <= set(su.DECISIONS)
        # a result nobody computes: not reproduced
        k = int(np.nonzero(flat)[0][0])
        bad = {"lnl": l2.copy(), "pendant_length": p2.copy(), "distal_length": d2.copy()}
        bad["pendant_length"][k] *= 1.37
        only = np.zeros(len(pb), bool)
        only[k] = True
        rep = su.reproduce_flat_pairs(o, c["reads"], pb, ps, bad, only)
        assert rep["unreproduced"] == [(int(pb[k]), int(ps[k]))]
    # trace comparison on hand-made rows
    a = [(1, 0.1, 2.0, 5.0), (1, 0.5, -1e-18, 3.0), (1, 0.7, 0.1, 3.0), (3, 10.0, 11.0, 0.0)]
    assert su.first_divergence(a, a) == (None, None)
    assert su.first_divergence(a, a[:2] + [(1, 0.3, 0.1, 3.0)] + a[3:]) == (2, "newton_branch")
    assert su.first_divergence(a, a[:2] + a[3:]) == (2, "newton_termination")
    assert su.first_divergence(a, a[:3] + [(3, 10.0, 11.0, 1.0)]) == (3, "round_decision")
    assert su.first_divergence(a, a + [(1, 0.1, 2.0, 5.0)]) == (4, "round_decision")
