#!/bin/bash
# Quick check of a kernel change in one gpurun call: the GPU test suite, the default bench line, a kernel trace of the
# short bench.   bash profiles/run_quick.sh <tag> [pytest args]
tag=${1:-q}
shift
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/${tag}_gputest.log 2>&1
tail -3 gpurun_out/${tag}_gputest.log
python bench.py --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_short.json 2> gpurun_out/${tag}_bench_short.err
tail -1 gpurun_out/${tag}_bench_short.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_stats
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o st -- \
  python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_stats.log 2>&1
cd $R
python profiles/db_to_txt.py gpurun_out/${tag}_stats/st_results.db > gpurun_out/${tag}_kernel_trace_stats.txt
head -22 gpurun_out/${tag}_kernel_trace_stats.txt
rm -rf gpurun_out/${tag}_stats
