#!/usr/bin/env python3
"""Golden-vector generator for the placement hot path (run in the build container, output
committed under tests/golden/*.json).

This is deliberately NOT a transcription of oracle/epa_oracle.c: it is an independent fp64
brute-force evaluator used to pin the oracle's arithmetic, because the reference itself cannot be
built here (its numerics are in un-vendored submodules):

  * P(t) = scipy.linalg.expm(Q r t) on the rate matrix itself (no eigen-decomposition);
  * discrete-Gamma mean rates from scipy.stats.gamma (ppf / cdf);
  * preplacement log-likelihoods by WHOLE-TREE Felsenstein pruning of the (n+1)-taxon tree with
    the query physically inserted at the branch midpoint (no directional CLVs, no lookup table);
  * thorough placement: the control flow of opt_branch_lengths_pplacer / pllmod_opt_minimize_newton
    (reference src/core/pll/optimize.cpp:60-248) driven by derivatives obtained from Q P(t)
    matrix products on a 3-taxon star (no sumtable).

Data fixtures (tests/golden/data/) are the reference's own test data files (test/data/ref.tre,
aln.fasta, query.fasta, AA_aln.fasta, AA_query.fasta) plus aa_ref.tre authored here because the
reference's AA alignment has no matching tree.
"""
import json
import math
import os
import sys

import numpy as np
from scipy.linalg import expm
from scipy.stats import gamma as gamma_dist

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")

DEFAULT_BL = -math.log(0.9)
MIN_BL, MAX_BL, DEF_OPT_BL = 1e-4, 100.0, 0.1

NT = {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12, "M": 3,
      "B": 14, "D": 13, "H": 11, "V": 7, "N": 15, "O": 15, "X": 15, "-": 15, "?": 15, ".": 15}
AA_ORDER = "ARNDCQEGHILKMFPSTWYV"


def char_vec(s, ch):
    ch = ch.upper()
    v = np.zeros(s)
    if s == 4:
        m = NT[ch]
        for i in range(4):
            if (m >> i) & 1:
                v[i] = 1.0
        return v
    if ch in AA_ORDER:
        v[AA_ORDER.index(ch)] = 1.0
    elif ch == "B":
        v[2] = v[3] = 1.0
    elif ch == "Z":
        v[5] = v[6] = 1.0
    elif ch in "X-?*":
        v[:] = 1.0
    else:
        raise ValueError(ch)
    return v


def read_fasta(path):
    out, name = [], None
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        if line[0] == ">":
            name = line[1:].split()[0]
            out.append([name, ""])
        else:
            out[-1][1] += line.upper()
    return out


# ---------------------------------------------------------------- tree
class Node:
    def __init__(self):
        self.kids, self.label, self.length, self.parent = [], None, 0.0, None


def parse_newick(s):
    pos = [0]

    def skip():
        while s[pos[0]].isspace():
            pos[0] += 1

    def lab_len(n):
        skip()
        b = pos[0]
        while s[pos[0]] not in ":,();":
            pos[0] += 1
        n.label = s[b:pos[0]].strip() or None
        if s[pos[0]] == ":":
            pos[0] += 1
            b = pos[0]
            while s[pos[0]] not in ",();":
                pos[0] += 1
            n.length = float(s[b:pos[0]])

    def sub():
        skip()
        n = Node()
        if s[pos[0]] == "(":
            pos[0] += 1
            while True:
                k = sub()
                k.parent = n
                n.kids.append(k)
                skip()
                if s[pos[0]] == ",":
                    pos[0] += 1
                    continue
                assert s[pos[0]] == ")"
                pos[0] += 1
                break
        lab_len(n)
        return n

    root = sub()
    for n in walk(root):
        if n.length == 0.0 and n is not root:
            n.length = DEFAULT_BL  # zero == missing (pll_util.cpp:13-39)
    return root


def walk(n):
    for k in n.kids:
        yield from walk(k)
    yield n


def branches_postorder(root):
    """edge i <-> node below it, in the order pinned by test/src/pll_util.cpp:134-143"""
    return [n for n in walk(root) if n is not root]


def numbered_newick(root, prec):
    idx = [0]

    def rec(n):
        if n.kids:
            inner = ",".join(rec(k) for k in n.kids)
            r = "(%s):%.*f{%d}" % (inner, prec, n.length, idx[0])
        else:
            r = "%s:%.*f{%d}" % (n.label, prec, n.length, idx[0])
        idx[0] += 1
        return r

    return "(" + ",".join(rec(k) for k in root.kids) + ");"


# ---------------------------------------------------------------- model
class Model:
    def __init__(self, s, subst, freqs, alpha, cats=4, rates=None, pinv=0.0):
        self.s = s
        self.pinv = pinv          # +I: rates / (1 - p) in P(t); site lk = (1-p) L + p pi_inv
        self.cinv = None          # [W] p * pi_inv per site (set by make_case from the reference tips)
        self.freqs = np.asarray(freqs, float)
        R = np.zeros((s, s))
        R[np.triu_indices(s, 1)] = subst
        R = R + R.T
        Q = R * self.freqs[None, :]
        np.fill_diagonal(Q, 0.0)
        np.fill_diagonal(Q, -Q.sum(1))
        Q /= -(self.freqs * np.diag(Q)).sum()
        self.Q = Q
        self.alpha = alpha
        if rates is None:
            g = gamma_dist(alpha, scale=1.0 / alpha)
            g1 = gamma_dist(alpha + 1.0, scale=1.0 / alpha)
            cuts = [0.0] + [g.ppf(i / cats) for i in range(1, cats)] + [np.inf]
            rates = [cats * (g1.cdf(cuts[i + 1]) - g1.cdf(cuts[i])) for i in range(cats)]
        self.rates = np.asarray(rates, float)
        self.weights = np.full(len(self.rates), 1.0 / len(self.rates))

    def P(self, t):
        return np.stack([expm(self.Q * (r * t / (1.0 - self.pinv))) for r in self.rates])  # [c][i][j]


# ---------------------------------------------------------------- likelihood (log-normalised)
def tip_partial(model, seq):
    v = np.stack([char_vec(model.s, ch) for ch in seq])  # [W][s]
    return np.repeat(v[:, None, :], len(model.rates), 1), np.zeros(len(seq))  # [W][c][s], logscale


def prune(model, node, seqs):
    """partial of the subtree below `node` (excluding its own branch): ([W][c][s], log factor[W])"""
    if not node.kids:
        return tip_partial(model, seqs[node.label])
    acc, logf = None, 0.0
    for k in node.kids:
        pk, lk = prune(model, k, seqs)
        P = model.P(k.length)
        term = np.einsum("cij,wcj->wci", P, pk)
        acc = term if acc is None else acc * term
        logf = logf + lk
    mx = acc.max(axis=(1, 2))
    return acc / mx[:, None, None], logf + np.log(mx)


def mix_inv(model, l0, logf):
    """+I: true site likelihood (1 - p) L + p pi_inv; l0 is normalised by exp(logf)"""
    if model.pinv == 0.0:
        return np.log(l0) + logf
    return np.log((1.0 - model.pinv) * l0 * np.exp(logf) + model.cinv[:len(l0)])


def root_lnl_sites(model, partial, logf):
    site = np.einsum("c,wci,i->w", model.weights, partial, model.freqs)
    return mix_inv(model, site, logf)


def tree_lnl(model, root, seqs):
    p, lf = prune(model, root, seqs)
    return float(root_lnl_sites(model, p, lf).sum())


def reroot_partials(model, root, seqs, node):
    """(distal partial = subtree below node, proximal partial = rest of tree seen from node's
    parent end), both excluding the branch above `node`."""
    dist = prune(model, node, seqs)

    def up(n):  # partial of everything except subtree(n), seen at n.parent, excluding n's branch
        par = n.parent
        acc, logf = None, 0.0
        for k in par.kids:
            if k is n:
                continue
            pk, lk = prune(model, k, seqs)
            term = np.einsum("cij,wcj->wci", model.P(k.length), pk)
            acc = term if acc is None else acc * term
            logf = logf + lk
        if par.parent is not None:
            pu, lu = up(par)
            term = np.einsum("cij,wcj->wci", model.P(par.length), pu)
            acc = acc * term
            logf = logf + lu
        mx = acc.max(axis=(1, 2))
        return acc / mx[:, None, None], logf + np.log(mx)

    return dist, up(node)


def valid_range(q):
    lo, hi = 0, len(q)
    while lo < hi and q[lo] == "-":
        lo += 1
    while hi > lo and q[hi - 1] == "-":
        hi -= 1
    return lo, hi - lo


def star_terms(model, tipp, dist, prox, tp, td, tx, order=(0, 0, 0)):
    """per-site likelihood of the 3-taxon star with (optionally differentiated) branches.
    order[i] = derivative order w.r.t. pendant / distal / proximal length."""
    def branch(p, t, k):
        P = model.P(t)
        for _ in range(k):
            P = np.einsum("c,ij,cjk->cik", model.rates / (1.0 - model.pinv), model.Q, P)
        return np.einsum("cij,wcj->wci", P, p)

    a = branch(tipp, tp, order[0])
    b = branch(dist[0], td, order[1])
    c = branch(prox[0], tx, order[2])
    return np.einsum("c,wci,i->w", model.weights, a * b * c, model.freqs)


def star_lnl(model, tipp, dist, prox, tp, td, tx, lo, n):
    l0 = star_terms(model, tipp, dist, prox, tp, td, tx)
    return float(mix_inv(model, l0, dist[1] + prox[1])[lo:lo + n].sum())


def newton(x1, xguess, x2, tol, max_iters, deriv):
    rts = min(max(xguess, x1), x2)
    f, df = deriv(rts)
    if df >= 0.0 and abs(f) < tol:
        return rts
    if f < 0.0:
        xl, xh = rts, x2
    else:
        xh, xl = rts, x1
    for i in range(1, max_iters + 1):
        if df <= 0.0 or (((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0):
            dx = 0.5 * (xh - xl)
            rts = xl + dx
            if xl == rts:
                return rts
        else:
            dx = f / df
            temp = rts
            rts -= dx
            if temp == rts:
                return rts
        if abs(dx) < tol or i == max_iters:
            return rts
        if rts < x1:
            rts = x1
        f, df = deriv(rts)
        if df > 0.0 and abs(f) < tol:
            return rts
        if f < 0.0:
            xl = rts
        else:
            xh = rts
    raise RuntimeError("newton overflow")


def thorough(model, tipp, dist, prox, orig, lo, n):
    tp, td, tx = DEFAULT_BL, orig / 2.0, orig / 2.0
    sl = slice(lo, lo + n)

    def d_of(which):
        def deriv(t):
            args = [tp, td, tx]
            args[which] = t
            o1 = [0, 0, 0]; o1[which] = 1
            o2 = [0, 0, 0]; o2[which] = 2
            l0 = star_terms(model, tipp, dist, prox, *args)[sl]
            l1 = star_terms(model, tipp, dist, prox, *args, order=tuple(o1))[sl]
            l2 = star_terms(model, tipp, dist, prox, *args, order=tuple(o2))[sl]
            if model.pinv > 0.0:   # derivatives of (1-p) L + p pi_inv: only L depends on t
                sc = np.exp((dist[1] + prox[1])[sl])
                l0 = (1.0 - model.pinv) * l0 * sc + model.cinv[sl]
                l1 = (1.0 - model.pinv) * l1 * sc
                l2 = (1.0 - model.pinv) * l2 * sc
            d1 = -l1 / l0
            return float(d1.sum()), float((d1 * d1 - l2 / l0).sum())
        return deriv

    negll = -star_lnl(model, tipp, dist, prox, tp, td, tx, lo, n)
    smoothings, rounds, reverted = 32, 0, False
    while smoothings:
        old_td, old_tp = td, tp
        xmin, xmax = MIN_BL, MAX_BL
        xguess = tp if xmin <= tp <= xmax else DEF_OPT_BL
        tp = newton(xmin, xguess, xmax, xmin / 10.0, 30, d_of(0))
        xmin = min(MIN_BL / 2.0, orig / 2.0)
        xtol = xmin / 10.0
        xmax = orig - xtol
        xguess = td if xmin <= td <= xmax else orig / 2.0
        td = newton(xmin, xguess, xmax, xtol, 30, d_of(1))   # proximal P still the old one
        tx = orig - td
        new = -star_lnl(model, tipp, dist, prox, tp, td, tx, lo, n)
        rounds += 1
        if new - negll > new * 1e-14:
            tp, td, tx = old_tp, old_td, orig - old_td
            reverted = True
            break
        smoothings -= 1
        if abs(new - negll) < 0.1:
            smoothings = 0
        negll = new
    return {"lnl": -negll, "pendant": tp, "distal": (orig / (td + tx)) * td, "rounds": rounds,
            "reverted": reverted}


# ---------------------------------------------------------------- cases
def make_case(name, tree_file, aln_file, queries, s, subst, freqs, alpha, full_pairs=True, pinv=0.0):
    newick = open(os.path.join(DATA, tree_file)).read().strip()
    root = parse_newick(newick)
    seqs = dict(read_fasta(os.path.join(DATA, aln_file)))
    model = Model(s, subst, freqs, alpha, pinv=pinv)
    brs = branches_postorder(root)
    W = len(next(iter(seqs.values())))
    # sites invariant over the reference tips (a tip's ambiguity / gap does not break it): the
    # intersection of the tips' state sets is a single state
    inter = np.ones((W, s), bool)
    for sq in seqs.values():
        inter &= np.stack([char_vec(s, ch) for ch in sq]) > 0
    inv_state = np.where(inter.sum(1) == 1, inter.argmax(1), -1)
    model.cinv = np.where(inv_state >= 0, pinv * model.freqs[np.maximum(inv_state, 0)], 0.0)
    out = {"name": name, "tree_file": tree_file, "aln_file": aln_file, "states": s,
           "subst": list(subst), "freqs": list(freqs), "alpha": alpha, "pinv": pinv,
           "invariant_state": inv_state.tolist(),
           "gamma_rates": model.rates.tolist(),
           "numbered_newick_p2": numbered_newick(root, 2),
           "tree_lnl": tree_lnl(model, root, seqs),
           "branch_lengths": [b.length for b in brs],
           "queries": [{"name": qn, "seq": qs} for qn, qs in queries],
           "preplace": [], "thorough": []}
    for qn, qs in queries:
        assert len(qs) == W
        lo, n = valid_range(qs)
        tipp = tip_partial(model, qs)[0]
        row_pre, row_thr = [], []
        for bi, b in enumerate(brs):
            dist, prox = reroot_partials(model, root, seqs, b)
            # the reference keeps a tip end DISTAL (Tiny_Tree.cpp:64-74); in post-order edge
            # enumeration the node below an edge is already the only end that can be a tip.
            row_pre.append(star_lnl(model, tipp, dist, prox, DEFAULT_BL, b.length / 2.0,
                                    b.length / 2.0, lo, n))
            # whole-tree cross-check of the star formulation: insert the query for real
            if bi % 5 == 0 and n == W:
                par = b.parent
                mid, tipn = Node(), Node()
                tipn.label, tipn.length = "__query__", DEFAULT_BL
                mid.length = b.length / 2.0
                mid.kids = [b, tipn]
                idx = par.kids.index(b)
                par.kids[idx] = mid
                old = b.length
                b.length = old / 2.0
                seqs["__query__"] = qs
                full = tree_lnl(model, root, seqs)
                del seqs["__query__"]
                b.length = old
                par.kids[idx] = b
                assert abs(full - row_pre[-1]) < 1e-8, (full, row_pre[-1])
            if full_pairs:
                row_thr.append(thorough(model, tipp, dist, prox, b.length, lo, n))
        out["preplace"].append(row_pre)
        out["thorough"].append(row_thr)
    return out


def derive_queries(base):
    """window / ambiguity variants of the bundled queries (exercise premasking + tip codes)"""
    out = []
    for name, seq in base:
        out.append((name, seq))
    n0, s0 = base[0]
    W = len(s0)
    w1 = "-" * 100 + s0[100:250] + "-" * (W - 250)
    out.append((n0 + "_win100_250", w1))
    w2 = list("-" * 300 + base[1][1][300:480] + "-" * (W - 480))
    for i in range(330, 340):
        w2[i] = "-"          # internal gap stays inside the window
    w2[350], w2[351], w2[352], w2[353] = "N", "R", "Y", "K"
    out.append((base[1][0] + "_win300_480_amb", "".join(w2)))
    w3 = s0[:60] + "-" * (W - 60)
    out.append((n0 + "_head60", w3))
    w4 = "-" * (W - 1) + s0[-1]
    out.append((n0 + "_last1", w4))
    return out


def main():
    os.makedirs(os.path.join(HERE, "golden"), exist_ok=True)
    # authored fixture: a tree over the AA alignment's labels
    aa_tree = ("(Cow:0.11,(Whale:0.09,Seal:0.13):0.04,((Human:0.15,Mouse:0.21):0.03,"
               "(Chicken:0.31,(Frog:0.36,Loach:0.42):0.08):0.12):0.05);")
    with open(os.path.join(DATA, "aa_ref.tre"), "w") as f:
        f.write(aa_tree + "\n")

    q = read_fasta(os.path.join(DATA, "query.fasta"))
    dnaq = derive_queries([(a, b) for a, b in q])
    only = set(sys.argv[1:])   # optional: regenerate just the named cases
    cases = []
    _make = make_case

    def make_case_sel(name, *a, **kw):
        return _make(name, *a, **kw) if (not only or name in only) else None
    # GTR+G defaults of raxml::Model("GTR+G") (Model.cpp:190-193,470,487-488)
    cases.append(make_case_sel("dna8_gtr_g_default", "ref.tre", "aln.fasta", dnaq, 4,
                           [0.5, 0.5, 0.5, 0.5, 0.5, 1.0], [0.25] * 4, 1.0))
    # model string pinned by test/src/parse_model.cpp:10-11
    cases.append(make_case_sel("dna8_gtr_fu_g4", "ref.tre", "aln.fasta", dnaq, 4,
                           [0.787874, 1.821672, 1.294006, 0.698421, 3.034135, 1.0],
                           [0.256465, 0.222535, 0.308594, 0.212406], 0.478218))
    # +I on the same data (model GTR+FU+I{0.2}+G4): pins the oracle's restatement of libpll's
    # invariant-site handling against the brute force (no scaling occurs on 8 taxa)
    cases.append(make_case_sel("dna8_gtr_fu_i_g4", "ref.tre", "aln.fasta", dnaq[:4], 4,
                           [0.787874, 1.821672, 1.294006, 0.698421, 3.034135, 1.0],
                           [0.256465, 0.222535, 0.308594, 0.212406], 0.478218, pinv=0.2))
    # 20-state: a deterministic pseudo-random PROTGTR (190 rates) + non-uniform freqs
    rng = np.random.RandomState(7)
    aasub = np.round(rng.gamma(1.0, 2.0, 190) + 0.01, 6).tolist()
    aaf = rng.dirichlet(np.full(20, 8.0))
    aaf = np.round(aaf / aaf.sum(), 6)
    aaf[-1] = round(1.0 - aaf[:-1].sum(), 6)
    aq = read_fasta(os.path.join(DATA, "AA_query.fasta"))
    aaq = [(a, b) for a, b in aq]
    s0 = aaq[0][1]
    W = len(s0)
    w = list("-" * 200 + s0[200:320] + "-" * (W - 320))
    w[210], w[211], w[212] = "X", "B", "Z"
    aaq.append((aaq[0][0] + "_win200_320_amb", "".join(w)))
    cases.append(make_case_sel("aa8_protgtr_g4", "aa_ref.tre", "AA_aln.fasta", aaq, 20, aasub,
                           aaf.tolist(), 0.563473))
    for c in [c for c in cases if c is not None]:
        path = os.path.join(HERE, "golden", c["name"] + ".json")
        with open(path, "w") as f:
            json.dump(c, f, indent=0)
        rev = sum(t["reverted"] for row in c["thorough"] for t in row)
        tot = sum(len(row) for row in c["thorough"])
        print(c["name"], "tree lnL", c["tree_lnl"], "pairs", tot, "reverted", rev, file=sys.stderr)


if __name__ == "__main__":
    main()
