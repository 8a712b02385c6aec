"""The randomised odd-shape generator of the parity sweep (tests/test_gpu_parity.py) as a function,
so that the sweep, the named outlier cases and the CPU self-consistency tests draw the SAME
configurations from a seed."""
import numpy as np

from epa_ng_amd import synth

# seeds of the 3000-seed hand run of round 2 (profiles/r2_sweep_3000.log) on which a pair ended in a
# different local optimum than the oracle's with |dlnL| > 1e-5, plus seed 2233 (1.2 % flat pairs)
OUTLIER_SEEDS = (231, 1070, 1457, 1671, 1882, 2297, 2409, 2623, 2692, 2695, 2869, 2233)


def make_case(seed):
    """4..90 tips, 12..500 columns, branch lengths 1e-8..20, alpha 0.05..50, +I, both alphabets,
    1-site to full-length reads with ambiguity codes"""
    rng = np.random.RandomState(seed)
    states = 4 if rng.rand() < 0.7 else 20
    tips = int(rng.choice([4, 5, 9, 17, 40, 90]))
    W = int(rng.choice([12, 64, 65, 130, 260, 500]))
    mean_bl = float(rng.choice([1e-5, 1e-3, 0.05, 0.5, 3.0]))
    hi = float(rng.choice([1.0, 20.0]))
    lo = float(rng.choice([1e-8, 1e-6, 1e-4]))
    pinv = float(rng.choice([0.0, 0.0, 0.35]))
    alpha = float(rng.choice([0.05, 0.5, 2.0, 50.0]))
    root = synth.random_tree(tips, seed, mean_bl=mean_bl, lo=lo, hi=hi)
    rates = synth.gamma_rates(alpha)
    subst, freqs = (synth.CFG2_SUBST, synth.CFG2_FREQS) if states == 4 else synth.aa_model(seed)
    labels, seqs = synth.simulate_msa(root, W, subst, freqs, rates, seed + 1)
    nreads = 24
    rl = min(int(rng.choice([1, 2, 3, 7, min(W, 64), min(W, 65), min(W, 129), W])), W)
    reads, _ = synth.make_reads(seqs, nreads, rl, float(rng.choice([0.0, 0.03, 0.5])), seed + 2, states)
    amb = "RYKMSWBDHVN-" if states == 4 else "BZX-"
    reads = list(reads)
    for i in range(0, nreads, 3):
        r = list(reads[i])
        idx = [k for k, ch in enumerate(r) if ch != "-"]
        if len(idx) > 2:
            for k in rng.choice(idx[1:-1], max(1, len(idx) // 6)):
                r[k] = amb[rng.randint(len(amb))]
        reads[i] = "".join(r)
    return dict(seed=seed, states=states, tips=tips, W=W, rl=rl, pinv=pinv, alpha=alpha,
                newick=synth.newick(root), labels=labels, seqs=seqs, subst=subst, freqs=freqs,
                rates=rates, reads=reads, nreads=nreads)


def lengths_differ(p_a, d_a, p_b, d_b):
    """the sweep's 'flat-optimum pair' predicate: optimised lengths differ by more than 1e-6"""
    return (np.abs(p_a - p_b) > 1e-6 * np.maximum(1.0, p_b)) | (np.abs(d_a - d_b) > 1e-6)
