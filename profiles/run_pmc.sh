#!/bin/bash
# PMC passes for bench.py on the GPU box (run through gpurun from the repo root):
#   bash profiles/run_pmc.sh <tag> "<counter group 1>" "<counter group 2>" ...
# Each group is its own rocprofv3 --pmc run (no trace domains alongside); CSVs land in
# gpurun_out/<tag>_pmc_<first counters>/ and are summarised by profiles/summarize_pmc.py.
tag=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
for grp in "$@"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-40)
  out=$R/gpurun_out/${tag}_pmc_$name
  mkdir -p "$out"
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$out" -o pmc -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS} > "$out.log" 2>&1
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && [ "$f" != "$out/pmc_counter_collection.csv" ] && cp "$f" "$out/pmc_counter_collection.csv"
done
cd $R
python profiles/summarize_pmc.py gpurun_out/${tag}_pmc_
