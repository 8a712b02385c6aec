import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import epa_ng_amd as epa
from epa_ng_amd import hostlib
import sweep_util as su
from oracle_lib import Oracle
lo, hi = int(sys.argv[1]), int(sys.argv[2])
tot = bad = flat = ncfg = 0
worst = 0.0
for seed in range(lo, hi):
    c = su.make_case(seed)
    if c["states"] != 4:
        continue
    pinv = 0.35 if seed % 2 == 0 else 0.1
    o = Oracle(c["newick"], c["labels"], c["seqs"], 4, c["subst"], c["freqs"], c["rates"], pinv=pinv)
    o.set_raxml_blo(True)
    ref = hostlib.Reference(c["newick"], c["labels"], c["seqs"], states=4, subst=c["subst"], freqs=c["freqs"], rates=c["rates"], pinv=pinv)
    ev = ref.evaluator(raxml_blo=True)
    reads = c["reads"]
    codes, wb, ws = epa.encode_queries(4, reads, compact=True)
    B = ref.B
    pairs = np.zeros(B * len(reads), epa.PAIR_DTYPE)
    pairs["branch_id"] = np.repeat(np.arange(B), len(reads)); pairs["seq_id"] = np.tile(np.arange(len(reads)), B)
    res = ev.thorough(pairs, codes, wb, ws)
    tl, tp, td = o.thorough(pairs["branch_id"], pairs["seq_id"], reads)
    dl = np.abs(res["lnl"] - tl)
    f = su.lengths_differ(res["pendant_length"], res["distal_length"], tp, td)
    nb = int(np.sum((dl > 1e-6) & ~f))
    ncfg += 1; tot += len(pairs); bad += nb; flat += int(f.sum()); worst = max(worst, float(dl[~f].max()) if (~f).any() else 0.0)
    if nb or ev.last_stats["rounds"] != o.last_stats["rounds"] and not f.any():
        print("seed", seed, "same-path pairs off", nb, "rounds", ev.last_stats["rounds"], o.last_stats["rounds"], flush=True)
print("raxml-blo +I nucleotide: %d configurations, %d pairs; same-path pairs with |dlnL| > 1e-6: %d (max %.3g); pairs on another path: %d (max |dlnL| among them see log)" % (ncfg, tot, bad, worst, flat))
