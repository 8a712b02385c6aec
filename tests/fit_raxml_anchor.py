"""External anchor: RAxML 8.2.4's EPA result on the reference's own fixture.

`test/data/raxml_output.jplace` of the reference checkout (copied to tests/golden/data/) is the output
of `raxmlHPC -f v -s aln.phy -t ref.tre -m GTRGAMMA` -- the original EPA, which EPA-ng reimplements --
on the classic 10-taxon mtDNA alignment (`aln.phy`, rows in the order Cow Carp Chicken Human Loach
Mouse Rat Seal Whale Frog, anonymised as A..J; `aln.fasta` / `query.fasta` hold the same sequences).
No test of the reference reads it, and its taxon names do not match the anonymised files, so it
looked unusable; matching the rows by content shows it is a coherent data set:

    tree (with RAxML's optimised branch lengths) : the "tree" field of the jplace, real names
    reference rows                               : aln.phy A C D E F H I J  (8 taxa)
    queries                                      : Carp = row B, Rat = row G

RAxML does not print the GTR rates and alpha it estimated.  This script recovers them by maximising
the 8-taxon tree log-likelihood at the jplace's branch lengths (6 free parameters, empirical base
frequencies over all 10 rows as RAxML computes them) with the CPU oracle, and prints the constants
that tests/test_external_anchor.py holds.  Run:  python tests/fit_raxml_anchor.py
"""
import json
import os
import re
import sys

import numpy as np
from scipy.optimize import minimize

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from oracle_lib import Oracle, gamma_rates  # noqa: E402

DATA = os.path.join(HERE, "golden", "data")
REAL_NAME_OF_ROW = dict(A="Cow", B="Carp", C="Chicken", D="Human", E="Loach", F="Mouse", G="Rat", H="Seal",
                        I="Whale", J="Frog")


def load():
    """-> newick with the row letters as labels and RAxML's branch lengths, reference labels / rows,
    queries {name: row}, base frequencies, the jplace placements"""
    phy = open(os.path.join(DATA, "aln.phy")).read().split("\n")
    n = int(phy[0].split()[0])
    rows = [l for l in phy[1:] if l.strip()]
    names, seqs = [], {}
    for r in rows[:n]:
        nm, sq = r.split(None, 1)
        names.append(nm)
        seqs[nm] = sq.replace(" ", "")
    for k, r in enumerate(rows[n:]):
        seqs[names[k % n]] += r.replace(" ", "")
    jp = json.load(open(os.path.join(DATA, "raxml_output.jplace")))
    nw = re.sub(r"\{\d+\}", "", jp["tree"])
    for row, real in REAL_NAME_OF_ROW.items():
        nw = nw.replace(real + ":", row + ":")
    labels = [r for r in names if REAL_NAME_OF_ROW[r] not in ("Carp", "Rat")]
    counts = np.array([sum(seqs[r].count(ch) for r in names) for ch in "ACGT"], float)
    queries = {REAL_NAME_OF_ROW[r]: seqs[r] for r in names if REAL_NAME_OF_ROW[r] in ("Carp", "Rat")}
    return nw, labels, [seqs[r] for r in labels], queries, counts / counts.sum(), jp["placements"]


def main():
    nw, labels, ref, queries, freqs, placements = load()

    def negll(x):
        o = Oracle(nw, labels, ref, 4, np.append(np.exp(x[:5]), 1.0), freqs, gamma_rates(float(np.exp(x[5]))))
        return -o.tree_lnl(0)

    x = np.zeros(6)
    for _ in range(3):   # restarts of the simplex from its own optimum
        r = minimize(negll, x, method="Nelder-Mead", options=dict(xatol=1e-9, fatol=1e-10, maxiter=6000))
        x = r.x
    print("tree lnL %.6f" % -r.fun)
    print("RATES =", [float("%.8f" % v) for v in np.exp(x[:5])] + [1.0])
    print("ALPHA = %.8f" % float(np.exp(x[5])))
    print("FREQS =", [float("%.8f" % v) for v in freqs])
    print(placements)


if __name__ == "__main__":
    main()
