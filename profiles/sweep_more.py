#!/usr/bin/env python3
"""Hand run of the randomised parity sweep (tests/test_gpu_parity.py check_sweep_case: evaluator parity at the
device's lengths for every pair, same-path pairs to 1e-6, every other pair reproduced by a rounding sibling of
the oracle) over NEW seeds, one log line per configuration:
    EPA_SWEEP_LOG=gpurun_out/r5_sweep_8200_8799.log python profiles/sweep_more.py 8200 8800
Bimodal configurations are logged, not failed (EPA_SWEEP_NO_BOUNDS); a violation of the rule stops the run."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("EPA_SWEEP_NO_BOUNDS", "1")
import test_gpu_parity as T  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
bad = []
for seed in range(lo, hi):
    try:
        T.check_sweep_case(seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:300]))
        print("VIOLATION seed %d: %s" % (seed, str(e)[:300]), flush=True)
print("seeds %d .. %d: %d configurations, %d violations, %.0f s" % (lo, hi - 1, hi - lo, len(bad), time.time() - t0))
sys.exit(1 if bad else 0)
