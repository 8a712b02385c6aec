#!/bin/bash
# same-box A/B of the 20-state kernel: one barrier per Newton evaluation (product) against the two-barrier form
# (variant library exp/libepa_dev_aa2b.so = -DAAM_PRIVATE_TABLES=0), cfg3 shape, three interleaved rounds
out=gpurun_out/r6_aa_ab.txt; : > $out
AA="--workload aa --tips 2000 --width 500 --read-len 100 --chunk 50000 --pool 3 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
for rep in 1 2 3; do
  for v in base aa2b; do
    unset EPA_DEV_SO
    [ $v = aa2b ] && export EPA_DEV_SO=$PWD/exp/libepa_dev_aa2b.so
    python bench.py $AA > /tmp/v.log 2>/tmp/v.err
    echo "== $v (rep $rep)" >> $out
    python - >> $out <<PY
import json
d=json.loads(open('/tmp/v.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['kernel_ms_per_step'], d['roofline']['frac'], d['roofline']['sclk_mhz'], d['roofline']['rounds_per_pair'], d['roofline']['newton_iters_per_solve'])
PY
    tail -1 /tmp/v.err | cut -c1-200 >> $out
  done
done
cat $out
