"""Pins oracle/epa_oracle.c against the REAL reference wherever it can be built: `make -C oracle ref`
(libpll-2, pll-modules and genesis installed) leaves oracle/_ref/epa_ref_driver -- the reference's own
Tree / Tiny_Tree / Lookup_Store classes compiled from /root/reference/src where they lie plus
oracle/ref_driver.cpp -- and this test diffs every (branch, query) pair of the reference's bundled data
against the oracle: tree lnL, Lookup_Store preplacement sums, thorough lnL / pendant / distal.

In this image the reference is unbuildable (its submodules are empty, no pll.h anywhere): the test SKIPS
and the parity grade stays "unpinned" -- it is the one-command path to pinning it elsewhere.  CPU only."""
import os
import subprocess

import numpy as np
import pytest

from golden_util import GOLDEN, load_case
from oracle_lib import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "epa_ref_driver")

needs_ref = pytest.mark.skipif(not os.path.exists(DRIVER),
                               reason="oracle/_ref/epa_ref_driver not built: the reference's libraries "
                                      "(libpll-2, pll-modules, genesis) are not installed here -- make -C oracle ref")


def _run(case, model, extra=()):
    g = load_case(case)
    d = os.path.join(GOLDEN, "data")
    query = os.path.join(d, "query.fasta" if g["states"] == 4 else "AA_query.fasta")
    out = subprocess.run([DRIVER, os.path.join(d, g["tree_file"]), os.path.join(d, g["aln_file"]), query, model, *extra],
                         capture_output=True, text=True, timeout=600, check=True).stdout
    tree_lnl, pre, tho = None, {}, {}
    for line in out.splitlines():
        f = line.split()
        if f[0] == "L":
            tree_lnl = float(f[1])
        elif f[0] == "P":
            pre[(int(f[1]), int(f[2]))] = float(f[3])
        elif f[0] == "T":
            tho[(int(f[1]), int(f[2]))] = tuple(float(x) for x in f[3:6])
    return g, query, tree_lnl, pre, tho


@needs_ref
@pytest.mark.parametrize("case,model,raxml_blo", [("dna8_gtr_g_default", "GTR+G", False),
                                                  ("dna8_gtr_g_default", "GTR+G", True)])
def test_oracle_equals_the_reference_on_its_bundled_data(case, model, raxml_blo):
    from golden_util import read_fasta
    g, query, tree_lnl, pre, tho = _run(case, model, ("--raxml-blo",) if raxml_blo else ())
    labels = [a for a, _ in g["msa"]]
    seqs = [b for _, b in g["msa"]]
    o = Oracle(g["newick"], labels, seqs, g["states"], g["subst"], g["freqs"], g["gamma_rates"])
    o.set_raxml_blo(raxml_blo)
    reads = [s for _, s in read_fasta(query)]
    B = o.B
    assert abs(o.tree_lnl(0) - tree_lnl) < 1e-6 * max(1.0, abs(tree_lnl))
    lnl = o.preplace(reads)
    worst_pre = max(abs(lnl[q, b] - v) for (b, q), v in pre.items())
    assert len(pre) == B * len(reads) and worst_pre < 1e-6
    pb = np.repeat(np.arange(B), len(reads))
    ps = np.tile(np.arange(len(reads)), B)
    tl, tp, td = o.thorough(pb, ps, reads)
    for i, (b, q) in enumerate(zip(pb, ps)):
        rl, rp, rd = tho[(int(b), int(q))]
        assert abs(tl[i] - rl) < 1e-6, (b, q, tl[i], rl)
        assert abs(tp[i] - rp) < 1e-6 * max(1.0, rp) and abs(td[i] - rd) < 1e-6, (b, q)


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("case,model,raxml_blo", [("dna8_gtr_g_default", "GTR+G", False),
                                                  ("dna8_gtr_g_default", "GTR+G", True)])
def test_device_equals_the_reference_on_its_bundled_data(case, model, raxml_blo):
    """The PRODUCT against the real reference: the same driver output, diffed against what
    libepa_dev.so returns through the C-ABI for every (branch, query) pair of the reference's bundled data -- tree lnL,
    preplacement table, thorough lnL / pendant / distal -- so that `make -C oracle ref && pytest tests/test_ref_pin.py`
    on a box with libpll and a GPU turns "parity: unpinned" into a statement about the device, not only about the
    checker.  oracle/_ref/ travels to the GPU box with the snapshot (git-ignored, not gpurun-ignored)."""
    import epa_ng_amd as epa
    from epa_ng_amd import hostlib
    from golden_util import read_fasta
    g, query, tree_lnl, pre, tho = _run(case, model, ("--raxml-blo",) if raxml_blo else ())
    labels = [a for a, _ in g["msa"]]
    seqs = [b for _, b in g["msa"]]
    ref = hostlib.Reference(g["newick"], labels, seqs, states=g["states"], subst=g["subst"], freqs=g["freqs"],
                            rates=g["gamma_rates"])
    ev = ref.evaluator(raxml_blo=raxml_blo)
    reads = [s for _, s in read_fasta(query)]
    B = ref.B
    assert abs(ev.tree_logl(0) - tree_lnl) < 1e-6 * max(1.0, abs(tree_lnl))
    codes, wb, ws = epa.encode_queries(g["states"], reads, premasking=False)
    lnl = ev.preplace(codes, wb, ws)
    assert len(pre) == B * len(reads) and max(abs(lnl[q, b] - v) for (b, q), v in pre.items()) < 1e-6
    pairs = np.zeros(B * len(reads), epa.PAIR_DTYPE)
    pairs["branch_id"] = np.repeat(np.arange(B), len(reads))
    pairs["seq_id"] = np.tile(np.arange(len(reads)), B)
    res = ev.thorough(pairs, codes, wb, ws)
    for i, (b, q) in enumerate(zip(pairs["branch_id"], pairs["seq_id"])):
        rl, rp, rd = tho[(int(b), int(q))]
        assert abs(res["lnl"][i] - rl) < 1e-6, (b, q, res["lnl"][i], rl)
        assert abs(res["pendant_length"][i] - rp) < 1e-6 * max(1.0, rp) and abs(res["distal_length"][i] - rd) < 1e-6, (b, q)


def test_the_recipe_says_why_it_cannot_run_here():
    """`make -C oracle ref` either builds the driver or stops with the message naming the missing library --
    it never fabricates one"""
    if os.path.exists(DRIVER):
        pytest.skip("the reference driver is built")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], capture_output=True, text=True)
    assert r.returncode != 0 and "unbuildable here" in r.stdout + r.stderr
    assert not os.path.exists(DRIVER)
