#!/bin/bash
# Same-box A/B of variant libraries (profiles/build_variant.sh): the short bench line alternately with the regular
# library and every exp/libepa_dev_<tag>.so named.   bash profiles/ab_variants.sh <out tag> <variant tag> ...
out=gpurun_out/$1_variants.txt; shift
mkdir -p gpurun_out; : > $out
for rep in 1 2; do
  for v in base "$@"; do
    unset EPA_PREPLACE_DB EPA_PREPLACE_DUO
    lib=${v%%+*}
    case $v in *+db) export EPA_PREPLACE_DB=1;; *+duo) export EPA_PREPLACE_DUO=1;; esac
    if [ $lib = base ]; then unset EPA_DEV_SO; else export EPA_DEV_SO=$PWD/exp/libepa_dev_$lib.so; fi
    python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extras $BENCH_ARGS > /tmp/v.log 2>/tmp/v.err
    echo "== $v (rep $rep)" >> $out
    grep -o '"value": [0-9.]*, "unit"' /tmp/v.log | head -1 >> $out
    grep -o '"kernel_ms_per_step": {[^}]*}' /tmp/v.log | head -1 >> $out
    grep -o '"preplace_max_abs_dlnl": [^,]*, "thorough_max_abs_dlnl": [^,]*' /tmp/v.log >> $out
    tail -2 /tmp/v.err | cut -c1-200 >> $out
  done
done
cat $out
