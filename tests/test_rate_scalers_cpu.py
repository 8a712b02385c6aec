"""Per-rate scalers (PLL_ATTRIB_RATE_SCALERS, src/tree/tiny_util.cpp:37-44; auto-on above 2000 tips,
src/io/file_io.cpp:211-214) in the CPU oracle, pinned against an independent log-space evaluator:
Felsenstein pruning done separately per rate category with logsumexp (no scaling scheme at all).

The alignment holds one designed column on which per-SITE scaling provably fails: a 640-tip clade
that is constant at that column (the slow categories keep likelihood ~1 there, so the site is never
rescaled, while a fast category decays to ~1e-346 and is flushed to zero), followed by 70 tips that
all disagree (which costs the slow categories ~1e-460 but the fast one only ~1e-42).  The fast
category dominates the true likelihood -- only per-rate scaling keeps it."""
import numpy as np
import pytest
from scipy.special import logsumexp

from epa_ng_amd import synth
from oracle_lib import Oracle

RATES = np.array([0.5e-9, 1.0e-9, 1.5, 2.5])
RATES = RATES / RATES.mean()
FREQS = [0.25, 0.25, 0.25, 0.25]
SUBST = [1.0] * 6
N_CLADE, N_TOP, W, BL = 640, 70, 6, 0.6


def designed_case():
    rng = np.random.RandomState(3)
    clade = synth.random_tree(N_CLADE, 5, mean_bl=BL, lo=BL, hi=BL)     # all branches = BL
    node = synth.Node()
    node.kids = clade.kids[:2]
    node.length = BL
    inner = synth.Node()
    inner.kids = [node, clade.kids[2]]
    inner.length = BL
    cur = inner
    tops = []
    for i in range(N_TOP):
        t = synth.Node("x%d" % i)
        t.length = BL
        tops.append(t)
    for t in tops[:-2]:
        p = synth.Node()
        p.kids = [cur, t]
        p.length = BL
        cur = p
    root = synth.Node()
    root.kids = [cur, tops[-2], tops[-1]]
    labels, seqs = [], []
    for i in range(N_CLADE):
        labels.append("t%d" % i)
        seqs.append("A" + "".join(rng.choice(list("ACGT"), W - 1)))
    for i in range(N_TOP):
        labels.append("x%d" % i)
        seqs.append("CGT"[i % 3] + "".join(rng.choice(list("ACGT"), W - 1)))
    return root, labels, seqs


def log_pruning(root, labels, seqs, extra=None):
    """per-category log-space pruning; extra = (branch node, distal, pendant, query string) inserts
    a query tip on the branch above `node` at `distal` from it.  -> per-site lnL [W]"""
    Q = synth.rate_matrix(SUBST, FREQS)
    seq_of = dict(zip(labels, seqs))
    out = np.zeros((len(RATES), W))
    lw = np.log(np.full(len(RATES), 1.0 / len(RATES)))

    def tipvec(s):
        v = np.full((W, 4), -np.inf)
        for w, ch in enumerate(s):
            if ch in "ACGT":
                v[w, "ACGT".index(ch)] = 0.0
            else:
                v[w, :] = 0.0
        return v
    for k, r in enumerate(RATES):
        def up(node):  # log CLV [W][4] at the top of node's own branch start (node itself)
            stack = [(node, False)]
            res = {}
            while stack:
                nd, done = stack.pop()
                if not nd.kids:
                    res[id(nd)] = tipvec(seq_of[nd.label])
                    continue
                if not done:
                    stack.append((nd, True))
                    for kd in nd.kids:
                        stack.append((kd, False))
                    continue
                acc = np.zeros((W, 4))
                for kd in nd.kids:
                    acc += through(res.pop(id(kd)), kd.length, kd)
                res[id(nd)] = acc
            return res[id(node)]

        def through(v, t, kd):
            if extra is not None and kd is extra[0]:
                lo = prop(v, extra[1])
                qv = prop(tipvec(extra[3]), extra[2])
                return prop(lo + qv, t - extra[1])
            return prop(v, t)

        def prop(v, t):
            P = synth.pmatrix(Q, FREQS, t * r)
            with np.errstate(divide="ignore"):
                lp = np.log(P)
            return logsumexp(lp[None, :, :] + v[:, None, :], axis=2)
        top = up(root)
        out[k] = logsumexp(top + np.log(np.asarray(FREQS))[None, :], axis=1)
    return logsumexp(out + lw[:, None], axis=0)


def test_per_rate_scalers_match_log_space_pruning_where_per_site_scaling_fails():
    root, labels, seqs = designed_case()
    nw = synth.newick(root)
    exact = log_pruning(root, labels, seqs)
    per_rate = Oracle(nw, labels, seqs, 4, SUBST, FREQS, RATES, rate_scalers=True)
    per_site = Oracle(nw, labels, seqs, 4, SUBST, FREQS, RATES)
    for b in (0, 7, per_rate.B - 1):
        assert abs(per_rate.tree_lnl(b) - exact.sum()) < 1e-7 * abs(exact.sum())
    # the designed column is what separates the schemes.  Per-site scaling loses the fast category
    # when the clade is aggregated BEFORE the disagreeing tips (the CLV pointing out of the clade,
    # used by the edges near the root); evaluated from inside the clade the order is reversed and
    # even per-site scaling copes -- with it the tree lnL is no longer the same on every edge
    assert abs(per_site.tree_lnl(per_site.B - 1) - exact.sum()) > 10.0
    assert abs(per_site.tree_lnl(0) - exact.sum()) < 1e-7 * abs(exact.sum())
    assert abs(per_rate.tree_lnl(0) - per_rate.tree_lnl(per_rate.B - 1)) < 1e-7
    # the ordinary columns alone: both schemes agree with the exact value
    ok_cols = [s_[1:] for s_ in seqs]
    a = Oracle(nw, labels, ok_cols, 4, SUBST, FREQS, RATES, rate_scalers=True)
    b_ = Oracle(nw, labels, ok_cols, 4, SUBST, FREQS, RATES)
    assert abs(a.tree_lnl(3) - exact[1:].sum()) < 1e-7 * abs(exact[1:].sum())
    assert abs(a.tree_lnl(3) - b_.tree_lnl(3)) < 1e-7


def test_per_rate_scalers_placement_matches_log_space_pruning():
    """preplacement (lookup sums) and the thorough result under per-rate scaling, checked by
    inserting the query into the tree at the lengths the oracle reports"""
    root, labels, seqs = designed_case()
    nw = synth.newick(root)
    o = Oracle(nw, labels, seqs, 4, SUBST, FREQS, RATES, rate_scalers=True)
    base = log_pruning(root, labels, seqs)
    q = "C" + seqs[N_CLADE + 3][1:]
    lnl = o.preplace([q])[0]
    # branch ids: utree_query_branches order == post-order over the three subtrees of the root
    order = []

    def post(nd):
        stack, outp = [nd], []
        while stack:
            x = stack.pop()
            outp.append(x)
            for kd in x.kids:
                stack.append(kd)
        return outp[::-1]
    # recursive post-order would overflow Python's stack on the caterpillar; emulate it
    def post_order(nd):
        res, stack = [], [(nd, 0)]
        while stack:
            x, i = stack.pop()
            if i < len(x.kids):
                stack.append((x, i + 1))
                stack.append((x.kids[i], 0))
            else:
                res.append(x)
        return res
    for kd in root.kids:
        order += post_order(kd)
    assert len(order) == o.B
    pend = -np.log(0.9)
    for b in (0, 11, o.B - 1, o.B - 2):
        nd = order[b]
        ins = log_pruning(root, labels, seqs, extra=(nd, nd.length / 2, pend, q))
        assert abs(ins.sum() - lnl[b]) < 1e-7 * abs(lnl[b])
    best = int(np.argmax(lnl))
    tl, tp, td = o.thorough([best], [0], [q])
    nd = order[best]
    ins = log_pruning(root, labels, seqs, extra=(nd, float(td[0]), float(tp[0]), q))
    assert abs(ins.sum() - tl[0]) < 1e-7 * abs(tl[0])
    assert tl[0] >= lnl[best] - 1e-9
    del base
