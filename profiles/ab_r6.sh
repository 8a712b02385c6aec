#!/bin/bash
# A/B on one box: base / spec-step variant library / staggered preplacement bursts, two interleaved rounds
out=gpurun_out/d_ab.txt; : > $out
for rep in 1 2 3; do
  for v in base spec stag; do
    unset EPA_DEV_SO EPA_BENCH_OPTS
    [ $v = spec ] && export EPA_DEV_SO=$PWD/exp/libepa_dev_spec.so
    [ $v = stag ] && export EPA_BENCH_OPTS=preplace_stagger=1
    python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extras > /tmp/v.log 2>/tmp/v.err
    echo "== $v (rep $rep)" >> $out
    python - >> $out <<PY
import json
d=json.loads(open('/tmp/v.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['schedule']['ms_per_step_serialised_two_slot_order'], d['config']['kernel_ms_per_step'], d['roofline']['sclk_mhz'], d['roofline']['rounds_per_pair'], d['roofline']['newton_iters_per_solve'])
PY
    tail -1 /tmp/v.err | cut -c1-200 >> $out
  done
done
cat $out
