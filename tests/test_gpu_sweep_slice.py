"""A fixed slice of the randomised parity sweep on seeds nobody has tuned a clause on.

The 192 configurations of test_gpu_parity.py (seeds 0 .. 191 + the named outliers) are the ones the flat-pair rule
of tests/sweep_util.py was written against; the builder's hand runs over ~9000 more seeds are builder-box logs.  This
module makes the DRIVER run further configurations -- seeds 20000 .. 20199, the first half of a range (20000 .. 20399)
that was fixed before any of its seeds had ever been run; the whole range ran once by hand with this module's rule
(profiles/r6_sweep_slice_20000_20399.log: 400 configurations, no violation, ONE over the per-configuration bounds: seed
20210, 16 of 4248 pairs in another local optimum 1.0 lnL away), the suite keeps the half that fits the driver's GPU
test time (~5 min) -- through the same check_sweep_case():
  1. evaluator parity at the device's own lengths, every pair, 1e-6            (unconditional)
  2. pairs on the oracle's path: lnL to 1e-6                                   (unconditional)
  3a/b. every other pair reproduced by a rounding sibling of the oracle (<= 2^8 ulp, 2^12 for lnL-equal pairs)
        at a named solver decision                                             (unconditional)
  3c. the per-configuration bounds (<= 1 % of the pairs off the oracle's path, each within 1e-4 lnL) cannot be
      asserted seed by seed without naming outliers after the fact -- about one random configuration in 300 is
      bimodal.  Stated up front instead: at most 1 % of the slice's configurations (2 of 200)
      may exceed them, and the last test of the module prints which did.
Twenty seeds per test item: 10 items of ~30 s."""
import os

import pytest

import test_gpu_parity as T

pytestmark = pytest.mark.gpu

FIRST, COUNT, PER_ITEM = 20000, 200, 20
_seen = {"configs": 0, "over_bounds": []}


@pytest.mark.parametrize("block", range(COUNT // PER_ITEM))
def test_untuned_seed_slice_of_the_parity_sweep(block, monkeypatch):
    import sweep_util as su
    monkeypatch.setenv("EPA_SWEEP_NO_BOUNDS", "1")      # (c) is accounted for below, over the whole slice
    for seed in range(FIRST + block * PER_ITEM, FIRST + (block + 1) * PER_ITEM):
        assert seed not in su.OUTLIER_BOUNDS
        info = T.check_sweep_case(seed)                  # asserts 1, 2, 3a, 3b
        _seen["configs"] += 1
        if info and (info["nflat"] > info["max_flat"] or info["dflat"] > info["max_dlnl"]):
            _seen["over_bounds"].append((seed, info["nflat"], info["pairs"], info["dflat"]))


def test_untuned_seed_slice_bimodal_fraction():
    if _seen["configs"] < COUNT:
        pytest.skip("the slice did not run completely in this session (-k / -x)")
    print("sweep slice %d .. %d: %d configurations, over the per-configuration bounds: %s"
          % (FIRST, FIRST + COUNT - 1, _seen["configs"], _seen["over_bounds"] or "none"))
    assert len(_seen["over_bounds"]) <= COUNT // 100, _seen["over_bounds"]
