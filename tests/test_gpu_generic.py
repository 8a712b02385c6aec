"""GPU parity of the general thorough kernel (k_thorough_generic) and of the setup kernels in the
modes the tuned kernels do not serve: any number of rate categories (+G8, +R5, ...), per-rate
scalers (PLL_ATTRIB_RATE_SCALERS) and --raxml-blo.  Checker: the oracle in the same mode.
Tolerances: per-branch lnL |delta| <= 1e-6; lengths 1e-6."""
import numpy as np
import pytest

import epa_ng_amd as epa
from epa_ng_amd import hostlib, synth
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu

LNL_TOL = 1e-6


def all_pairs(B, Q):
    p = np.zeros(B * Q, epa.PAIR_DTYPE)
    p["branch_id"] = np.repeat(np.arange(B), Q)
    p["seq_id"] = np.tile(np.arange(Q), B)
    return p


def check_against_oracle(ev, o, reads, states, pairs=None, compact=True):
    codes, wb, ws = epa.encode_queries(states, reads, compact=compact)
    lnl = ev.preplace(codes, wb, ws)
    assert np.max(np.abs(lnl - o.preplace(reads))) < LNL_TOL
    if pairs is None:
        pairs = ev.select(lnl, len(reads), 0.99999)
    res = ev.thorough(pairs, codes, wb, ws)
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], reads)
    assert np.max(np.abs(res["lnl"] - tl)) < LNL_TOL
    assert np.max(np.abs(res["pendant_length"] - tp) / np.maximum(1.0, tp)) < 1e-6
    assert np.max(np.abs(res["distal_length"] - td)) < 1e-6
    assert ev.last_stats["rounds"] == o.last_stats["rounds"]
    assert ev.last_stats["newton_evals"] == o.last_stats["newton_evals"]
    return lnl, pairs, res


@pytest.mark.parametrize("states,cats,pinv", [(4, 3, 0.0), (4, 5, 0.0), (4, 8, 0.2), (4, 16, 0.0), (20, 6, 0.0),
                                              (20, 3, 0.15)])
def test_any_number_of_rate_categories(states, cats, pinv):
    """+G8 / +R5-style models: `cats` categories with unequal weights (free rates), optional +I"""
    rng = np.random.RandomState(100 + cats)
    rates = np.sort(rng.gamma(0.7, 1.5, cats)) + 1e-3
    weights = rng.dirichlet(np.full(cats, 4.0))
    rates = rates / np.sum(rates * weights)            # mean rate 1 (Model.cpp:405-455)
    subst, freqs = (synth.CFG2_SUBST, synth.CFG2_FREQS) if states == 4 else synth.aa_model(3)
    root = synth.random_tree(40, 7 + cats)
    labels, seqs = synth.simulate_msa(root, 260, subst, freqs, synth.gamma_rates(0.7), 8)
    nw = synth.newick(root)
    reads, _ = synth.make_reads(seqs, 48, 150 if states == 4 else 90, 0.05, 9, states=states)
    ref = hostlib.Reference(nw, labels, seqs, states=states, subst=subst, freqs=freqs, rates=rates,
                            weights=weights, pinv=pinv)
    o = Oracle(nw, labels, seqs, states, subst, freqs, rates, weights=weights, pinv=pinv)
    for device_precompute in (True, False):
        ev = ref.evaluator(device_precompute=device_precompute)
        assert abs(ev.tree_logl(2) - o.tree_lnl(2)) < 1e-7
        check_against_oracle(ev, o, reads, states)
    # the fused chunk body and --no-heur run on the same kernel
    codes, wb, ws = epa.encode_queries(states, reads, compact=True)
    p, r = ev.place_chunk(codes, wb, ws)
    tl, _, _ = o.thorough(p["branch_id"], p["seq_id"], reads)
    assert np.max(np.abs(r["lnl"] - tl)) < LNL_TOL


@pytest.mark.parametrize("cats,pinv", [(3, 0.0), (5, 0.0), (8, 0.0), (8, 0.25), (12, 0.0), (16, 0.0)])
def test_category_group_kernel_equals_general_kernel(cats, pinv, monkeypatch):
    """Nucleotide models with 3 / 5 .. 16 rate categories run on k_thorough_dna with one wave per group
    of four categories (padded with weight-0 copies of the last category); the option thorough_generic sends the
    same context shape to k_thorough_generic.  Windows of every single-wave class (incl. the lengths
    the four-category kernel serves with half-chunk tails) and one beyond them (general kernel in both
    contexts): same pairs, lnL to 1e-9, lengths to 1e-9, identical round / Newton-evaluation counters;
    and the oracle on the mixed chunk."""
    rng = np.random.RandomState(300 + cats)
    rates = np.sort(rng.gamma(0.6, 1.5, cats)) + 1e-3
    weights = rng.dirichlet(np.full(cats, 3.0))
    rates = rates / np.sum(rates * weights)
    root = synth.random_tree(30, 50 + cats)
    labels, seqs = synth.simulate_msa(root, 400, synth.CFG2_SUBST, synth.CFG2_FREQS, synth.gamma_rates(0.7), 51)
    nw = synth.newick(root)
    reads = []
    for k, rl in enumerate((40, 64, 80, 100, 128, 150, 170, 192, 250)):
        r, _ = synth.make_reads(seqs, 4, rl, 0.04, 60 + k, states=4)
        reads += list(r)
    ref = hostlib.Reference(nw, labels, seqs, states=4, subst=synth.CFG2_SUBST, freqs=synth.CFG2_FREQS,
                            rates=rates, weights=weights, pinv=pinv)
    codes, wb, ws = epa.encode_queries(4, reads, compact=True)
    pairs = all_pairs(ref.B, len(reads))[::3].copy()
    ev = ref.evaluator()
    res = ev.thorough(pairs, codes, wb, ws)
    st = dict(ev.last_stats)
    evg = ref.evaluator()
    evg.set_option("thorough_generic", 1)
    resg = evg.thorough(pairs, codes, wb, ws)
    assert np.max(np.abs(res["lnl"] - resg["lnl"])) < 1e-9
    assert np.max(np.abs(res["pendant_length"] - resg["pendant_length"])) < 1e-9
    assert np.max(np.abs(res["distal_length"] - resg["distal_length"])) < 1e-9
    assert st["rounds"] == evg.last_stats["rounds"] and st["newton_evals"] == evg.last_stats["newton_evals"]
    o = Oracle(nw, labels, seqs, 4, synth.CFG2_SUBST, synth.CFG2_FREQS, rates, weights=weights, pinv=pinv)
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], reads)
    assert np.max(np.abs(res["lnl"] - tl)) < LNL_TOL
    assert st["rounds"] == o.last_stats["rounds"]


@pytest.mark.parametrize("cats,pinv,blo", [(4, 0.0, "sliding"), (5, 0.0, "sliding"), (8, 0.0, "sliding"), (8, 0.2, "sliding"),
                                           (8, 0.0, "raxml"), (4, 0.0, "raxml")])
def test_aa_matrix_core_kernel_8_categories_and_long_windows(cats, pinv, blo, monkeypatch):
    """20-state models on k_thorough_aa_mfma beyond {4 categories, 192 sites}: 8 categories (+G8; +R5 padded
    with weight-0 copies of its last category) as NC = 8 instantiations, windows of up to 384 (4 categories)
    / 256 (8) residues on 8-wave workgroups; longer ones in the same chunk go to the lane = site / general
    kernel.  The option thorough_generic sends the same context shape to k_thorough_generic: same pairs, lnL and lengths
    to 1e-9, identical round / Newton-evaluation counters; and the oracle on the mixed chunk."""
    rng = np.random.RandomState(700 + cats)
    rates = np.sort(rng.gamma(0.6, 1.5, cats)) + 1e-3
    weights = rng.dirichlet(np.full(cats, 3.0))
    rates = rates / np.sum(rates * weights)
    subst, freqs = synth.aa_model(9)
    root = synth.random_tree(24, 90 + cats)
    labels, seqs = synth.simulate_msa(root, 420, subst, freqs, synth.gamma_rates(0.7), 91)
    nw = synth.newick(root)
    reads = []
    for k, rl in enumerate((30, 64, 65, 100, 128, 129, 192, 193, 250, 256, 257, 300, 384, 385, 410)):
        r, _ = synth.make_reads(seqs, 2, rl, 0.05, 160 + k, states=20)
        reads += list(r)
    kw = dict(raxml_blo=True) if blo == "raxml" else {}
    ref = hostlib.Reference(nw, labels, seqs, states=20, subst=subst, freqs=freqs, rates=rates, weights=weights, pinv=pinv)
    codes, wb, ws = epa.encode_queries(20, reads, compact=True)
    pairs = all_pairs(ref.B, len(reads))[::2].copy()
    ev = ref.evaluator(**kw)
    res = ev.thorough(pairs, codes, wb, ws)
    st = dict(ev.last_stats)
    evg = ref.evaluator(**kw)
    evg.set_option("thorough_generic", 1)
    resg = evg.thorough(pairs, codes, wb, ws)
    assert np.max(np.abs(res["lnl"] - resg["lnl"])) < 1e-9
    assert np.max(np.abs(res["pendant_length"] - resg["pendant_length"])) < 1e-9
    assert np.max(np.abs(res["distal_length"] - resg["distal_length"])) < 1e-9
    assert st["rounds"] == evg.last_stats["rounds"] and st["newton_evals"] == evg.last_stats["newton_evals"]
    o = Oracle(nw, labels, seqs, 20, subst, freqs, rates, weights=weights, pinv=pinv)
    if blo == "raxml":
        o.set_raxml_blo(True)
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], reads)
    assert np.max(np.abs(res["lnl"] - tl)) < LNL_TOL
    assert np.max(np.abs(res["distal_length"] - td)) < 1e-6
    assert st["rounds"] == o.last_stats["rounds"]


@pytest.mark.parametrize("states", [4, 20])
def test_per_rate_scalers_equal_per_site_scalers_on_ordinary_data(states):
    """where nothing underflows the two scaling schemes give the same numbers (the device in
    either mode, the oracle in either mode); a deep tree makes sure scalers are non-zero"""
    subst, freqs = (synth.CFG2_SUBST, synth.CFG2_FREQS) if states == 4 else synth.aa_model(5)
    root = synth.random_tree(300, 41, mean_bl=0.4, lo=0.05, hi=2.0)
    rates = synth.gamma_rates(0.4)
    labels, seqs = synth.simulate_msa(root, 96, subst, freqs, rates, 42)
    nw = synth.newick(root)
    reads, _ = synth.make_reads(seqs, 20, 60, 0.05, 43, states=states)
    ref = hostlib.Reference(nw, labels, seqs, states=states, subst=subst, freqs=freqs, rates=rates)
    out = {}
    for rs in (False, True):
        o = Oracle(nw, labels, seqs, states, subst, freqs, rates, rate_scalers=rs)
        ev = ref.evaluator(rate_scalers=rs)
        assert abs(ev.tree_logl(0) - o.tree_lnl(0)) < 1e-6
        out[rs] = check_against_oracle(ev, o, reads, states)
    assert np.max(np.abs(out[True][0] - out[False][0])) < 1e-7
    assert np.array_equal(out[True][1], out[False][1])
    assert np.max(np.abs(out[True][2]["lnl"] - out[False][2]["lnl"])) < 1e-7


def test_per_rate_scalers_where_per_site_scaling_underflows():
    """the designed case of tests/test_rate_scalers_cpu.py (one column on which per-site scaling
    provably loses the dominant rate category) on the device: per-rate mode == the oracle's
    per-rate mode, which that file pins against log-space pruning; the per-site device context
    reproduces the oracle's per-site failure (tree lnL no longer equal on every edge)"""
    import test_rate_scalers_cpu as T
    root, labels, seqs = T.designed_case()
    nw = synth.newick(root)
    ref = hostlib.Reference(nw, labels, seqs, states=4, subst=T.SUBST, freqs=T.FREQS, rates=T.RATES)
    o = Oracle(nw, labels, seqs, 4, T.SUBST, T.FREQS, T.RATES, rate_scalers=True)
    ev = ref.evaluator(rate_scalers=True)
    for b in (0, 9, ref.B - 1):
        assert abs(ev.tree_logl(b) - o.tree_lnl(b)) < 1e-7 * abs(o.tree_lnl(b))
    qs = ["C" + seqs[T.N_CLADE + 3][1:], "A" + seqs[5][1:], "G" + seqs[T.N_CLADE + 20][1:]]
    check_against_oracle(ev, o, qs, 4, pairs=all_pairs(ref.B, len(qs))[::7].copy())
    ps = Oracle(nw, labels, seqs, 4, T.SUBST, T.FREQS, T.RATES)
    evs = ref.evaluator(rate_scalers=False)
    assert abs(evs.tree_logl(ref.B - 1) - ps.tree_lnl(ref.B - 1)) < 1e-7 * abs(ps.tree_lnl(ref.B - 1))
    assert abs(evs.tree_logl(ref.B - 1) - ev.tree_logl(ref.B - 1)) > 10.0


@pytest.mark.parametrize("states,rs", [(4, False), (20, False), (4, True)])
def test_raxml_blo_local_optimisation(states, rs):
    """--raxml-blo (src/core/pll/optimize.cpp:274-279): radius-1 local optimisation of the three
    branches of the triplet instead of the sliding rule"""
    subst, freqs = (synth.CFG2_SUBST, synth.CFG2_FREQS) if states == 4 else synth.aa_model(11)
    root = synth.random_tree(48, 51)
    rates = synth.gamma_rates(0.6)
    labels, seqs = synth.simulate_msa(root, 300, subst, freqs, rates, 52)
    nw = synth.newick(root)
    reads, _ = synth.make_reads(seqs, 40, 120 if states == 4 else 80, 0.04, 53, states=states)
    ref = hostlib.Reference(nw, labels, seqs, states=states, subst=subst, freqs=freqs, rates=rates)
    o = Oracle(nw, labels, seqs, states, subst, freqs, rates, rate_scalers=rs)
    o.set_raxml_blo(True)
    ev = ref.evaluator(raxml_blo=True, rate_scalers=rs)
    _, pairs, res = check_against_oracle(ev, o, reads, states)
    bl = np.array([ref.branch(int(b))["length"] for b in range(ref.B)])
    assert np.all(res["distal_length"] >= 0) and np.all(res["distal_length"] <= bl[pairs["branch_id"]] + 1e-12)
    ev2 = ref.evaluator()
    codes, wb, ws = epa.encode_queries(states, reads, compact=True)
    sliding = ev2.thorough(pairs, codes, wb, ws)
    # three free lengths instead of one sliding split: better on average; both optimisers stop
    # once a round moves lnL by less than OPT_BRANCH_EPSILON = 0.1, so "never worse" only holds
    # up to a few stopping tolerances
    assert np.mean(res["lnl"] - sliding["lnl"]) > 0.0
    assert np.all(res["lnl"] >= sliding["lnl"] - 0.5)


@pytest.mark.parametrize("states,pinv", [(4, 0.0), (4, 0.15), (20, 0.0), (20, 0.2)])
def test_raxml_blo_tuned_kernels_equal_general_kernel(monkeypatch, states, pinv):
    """--raxml-blo runs on the LOCAL instantiations of k_thorough_dna (register sumtable, 1 - 8 waves per
    pair; with +I since round 5) and k_thorough_aa_mfma (windows up to 192 sites; +I included); the general kernel
    (option thorough_generic) is the cross-check.  The window lengths cover every span class of the tuned
    kernels and, for 20 states, the hand-over to the general kernel beyond 192 sites."""
    root = synth.random_tree(40, 61)
    rates = synth.gamma_rates(0.5)
    subst, freqs = (synth.CFG2_SUBST, synth.CFG2_FREQS) if states == 4 else synth.aa_model(17)
    labels, seqs = synth.simulate_msa(root, 700, subst, freqs, rates, 62)
    nw = synth.newick(root)
    reads = []
    for k, rl in enumerate((20, 64, 150, 192, 300, 500) if states == 4 else (20, 64, 100, 128, 160, 192)):
        r, _ = synth.make_reads(seqs, 6, rl, 0.03, 63 + k, states=states)
        reads += list(r)
    ref = hostlib.Reference(nw, labels, seqs, states=states, subst=subst, freqs=freqs, rates=rates, pinv=pinv)
    o = Oracle(nw, labels, seqs, states, subst, freqs, rates, pinv=pinv)
    o.set_raxml_blo(True)
    ev = ref.evaluator(raxml_blo=True)
    _, pairs, res = check_against_oracle(ev, o, reads, states)
    tuned_stats = dict(ev.last_stats)
    evg = ref.evaluator(raxml_blo=True)
    evg.set_option("thorough_generic", 1)
    codes, wb, ws = epa.encode_queries(states, reads, compact=True)
    gen = evg.thorough(pairs, codes, wb, ws)
    assert np.max(np.abs(gen["lnl"] - res["lnl"])) < 1e-8
    assert np.max(np.abs(gen["pendant_length"] - res["pendant_length"])) < 1e-7
    assert np.max(np.abs(gen["distal_length"] - res["distal_length"])) < 1e-7
    assert evg.last_stats["rounds"] == tuned_stats["rounds"]


@pytest.mark.parametrize("states,cats", [(4, 4), (20, 4), (4, 5)])
def test_recollected_optimiser_constants_are_runtime_parameters(states, cats):
    """PLLMOD_OPT_MIN_BRANCH_LEN and the two variable details of pllmod_opt_minimize_newton are
    recollections (the pll-modules source is not in the reference's tree): they are parameters of
    the C-ABI (epa_ref_desc.blo_min_branch, EPA_FLAG_NEWTON_*), and under every setting the device
    follows the oracle configured the same way -- so pinning parity later is a matter of flags.
    profiles/r2_sensitivity.md holds what each of them changes."""
    subst, freqs = (synth.CFG2_SUBST, synth.CFG2_FREQS) if states == 4 else synth.aa_model(13)
    root = synth.random_tree(40, 61)
    rates = synth.gamma_rates(0.5, cats) if cats != 4 else synth.gamma_rates(0.5)
    labels, seqs = synth.simulate_msa(root, 300, subst, freqs, synth.gamma_rates(0.5), 62)
    nw = synth.newick(root)
    reads, _ = synth.make_reads(seqs, 60, 150 if states == 4 else 90, 0.04, 63, states=states)
    ref = hostlib.Reference(nw, labels, seqs, states=states, subst=subst, freqs=freqs, rates=rates)
    o = Oracle(nw, labels, seqs, states, subst, freqs, rates)
    base = None
    moved = 0
    for nv, mn in ((0, 1e-4), (1, 1e-4), (2, 1e-4), (3, 1e-6), (0, 1e-6)):
        o.set_blo(min_branch=mn, newton_variant=nv)
        ev = ref.evaluator(newton_variant=nv, blo_min_branch=mn)
        _, pairs, res = check_against_oracle(ev, o, reads, states)
        if base is None:
            base = res
        else:
            moved += int((np.abs(res["lnl"] - base["lnl"]) > 1e-6).sum())
    assert moved > 0      # the switches are not no-ops


@pytest.mark.parametrize("states", [4, 20])
def test_keep_eigenvalues_switch_runs_every_term_and_agrees_on_ordinary_data(states):
    """EPA_FLAG_KEEP_EIGENVALUES: the stationary eigenvalue out of the eigen-solver (1e-17-ish) is used verbatim on
    the general kernel instead of being set to exactly 0 -- the switch a maintainer with a real libpll build
    needs to A/B the snap.  On ordinary data (no saturated lengths) both contexts and the oracle agree."""
    subst, freqs = (synth.CFG2_SUBST, synth.CFG2_FREQS) if states == 4 else synth.aa_model(5)
    root = synth.random_tree(40, 77)
    rates = synth.gamma_rates(0.6)
    labels, seqs = synth.simulate_msa(root, 300, subst, freqs, rates, 78)
    nw = synth.newick(root)
    reads, _ = synth.make_reads(seqs, 32, 120 if states == 4 else 80, 0.04, 79, states=states)
    ref = hostlib.Reference(nw, labels, seqs, states=states, subst=subst, freqs=freqs, rates=rates)
    o = Oracle(nw, labels, seqs, states, subst, freqs, rates)
    ev_keep = ref.evaluator(keep_eigenvalues=True)
    ev_snap = ref.evaluator()
    _, pairs, res_keep = check_against_oracle(ev_keep, o, reads, states)
    codes, wb, ws = epa.encode_queries(states, reads, compact=True)
    res_snap = ev_snap.thorough(pairs, codes, wb, ws)
    assert np.max(np.abs(res_keep["lnl"] - res_snap["lnl"])) < 1e-8
    assert np.max(np.abs(res_keep["pendant_length"] - res_snap["pendant_length"])) < 1e-8
    # +I needs the exact zero mode: refused, not silently snapped
    refi = hostlib.Reference(nw, labels, seqs, states=states, subst=subst, freqs=freqs, rates=rates, pinv=0.2)
    with pytest.raises(Exception):
        refi.evaluator(keep_eigenvalues=True)


@pytest.mark.parametrize("states,cats", [(4, 4), (4, 5), (4, 8), (20, 4), (20, 6)])
def test_descriptor_path_with_caller_supplied_per_rate_scaler_rows(states, cats):
    """epa_dev_create (the binding of INTEGRATION.md: libpll's CLVs and scale buffers handed over as they are) with
    EPA_FLAG_RATE_SCALERS: scaler rows are uint32 [W][c] per branch side (PLL_ATTRIB_RATE_SCALERS,
    src/tree/tiny_util.cpp:37-44).  Producer: the oracle in per-rate mode on a deep tree (non-zero, category-
    dependent counts).  The device transposes the rows, aligns every category to the site's minimum once per
    branch side (k_add_scaler, k_align_rates) and must reproduce the oracle's tree lnL, preplacement and thorough
    results -- with 4 categories on the tuned kernels, with 5 / 6 / 8 on the general kernel (caller rows are
    not padded)."""
    rng = np.random.RandomState(900 + cats)
    rates = np.sort(rng.gamma(0.5, 1.5, cats)) + 1e-3
    weights = np.full(cats, 1.0 / cats)
    rates = rates / np.sum(rates * weights)
    subst, freqs = (synth.CFG2_SUBST, synth.CFG2_FREQS) if states == 4 else synth.aa_model(4)
    root = synth.random_tree(260, 33 + cats, mean_bl=0.5, lo=0.05, hi=3.0)
    labels, seqs = synth.simulate_msa(root, 90, subst, freqs, synth.gamma_rates(0.4), 34)
    nw = synth.newick(root)
    reads, _ = synth.make_reads(seqs, 16, 60, 0.05, 35, states=states)
    o = Oracle(nw, labels, seqs, states, subst, freqs, rates, weights=weights, rate_scalers=True)
    ev_, u, ui = o.eigen()
    pc, ps, dc, ds, bl = [], [], [], [], []
    for b in range(o.B):
        cp, sp, cd, sd = o.branch_sides(b)
        pc.append(cp); ps.append(sp); dc.append(cd); ds.append(sd)
        bl.append(o.branch_info(b)[0])
    assert max(int(x.max()) for x in ps) > 0                       # the tree is deep enough to rescale
    assert any(len(set(x.reshape(-1, cats)[w])) > 1 for x in ps for w in range(0, 90, 7))   # ... differently per category
    ev = epa.Evaluator(states, rates, weights, ev_, u, ui, freqs, bl, pc, dc, ps, ds, flags=0x2)
    for b in (0, o.B // 2, o.B - 1):
        assert abs(ev.tree_logl(b) - o.tree_lnl(b)) < 1e-7 * abs(o.tree_lnl(b))
    check_against_oracle(ev, o, reads, states)
