#!/bin/bash
tag=${1:-ab}
R=$PWD
mkdir -p gpurun_out
out=$R/gpurun_out/${tag}_small_kernels.txt
: > $out
cd /tmp && export TMPDIR=/tmp
run() {   # name, env...
  name=$1; shift
  rm -rf /tmp/abst
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abst -o st -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/ab.log 2>&1
  echo "== $name ($*)" >> $out
  grep -o '"value": [0-9.]*, "unit"' /tmp/ab.log | head -1 >> $out
  grep -o '"kernel_ms_per_step": {[^}]*}' /tmp/ab.log | head -1 >> $out
  grep -o '"ms_per_step": [0-9.]*' /tmp/ab.log | head -1 >> $out
  grep -o '"parity": {[^}]*}' /tmp/ab.log | cut -c1-200 >> $out
  python $R/profiles/step_timeline.py /tmp/abst/st_results.db >> $out 2>&1
  python $R/profiles/db_to_txt.py /tmp/abst/st_results.db | grep -E "k_select_seg|k_pack_pairs|k_class_hist|k_thorough_dna|k_preplace_pairs" | cut -c1-60,96-140 >> $out
}
run cur A=1
run nohop EPA_NO_BASE_HOP=1
run cur2 A=1
run nohop2 EPA_NO_BASE_HOP=1
cd $R
cat $out
