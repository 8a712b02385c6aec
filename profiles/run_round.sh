#!/bin/bash
# Everything the committed profiles/<tag>_* files come from, in one gpurun call:
#   bash profiles/run_round.sh <tag>
tag=${1:-r3}
R=$PWD
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -1 gpurun_out/${tag}_bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_stats
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o st -- \
  python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_stats.log 2>&1
cd $R
python profiles/db_to_txt.py gpurun_out/${tag}_stats/st_results.db > gpurun_out/${tag}_kernel_trace_stats.txt
python profiles/step_timeline.py gpurun_out/${tag}_stats/st_results.db 12 > gpurun_out/${tag}_step_timeline.txt 2>/dev/null
head -12 gpurun_out/${tag}_kernel_trace_stats.txt
# (round 6) the default command runs the DEEP pipeline: kernels of neighbouring chunks overlap on the device and their traced
# durations include the wait for each other.  roofline.ms_per_launch / kernel_ms_per_step come from the serialised pass;
# the trace that must agree with them is the same bench in the serialised two-slot order everywhere:
cd /tmp
rm -rf $R/gpurun_out/${tag}_stats_serial
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats_serial -o st -- \
  python $R/bench.py --no-cpu-baseline --no-extras --serial-schedule > $R/gpurun_out/${tag}_stats_serial.log 2>&1
cd $R
python profiles/db_to_txt.py gpurun_out/${tag}_stats_serial/st_results.db > gpurun_out/${tag}_kernel_trace_stats_serialised.txt
head -6 gpurun_out/${tag}_kernel_trace_stats_serialised.txt
bash profiles/run_pmc.sh ${tag} \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_SMEM" \
  "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" > gpurun_out/${tag}_pmc_summary.txt 2>&1
grep -A22 "== k_thorough" gpurun_out/${tag}_pmc_summary.txt | head -30
# 20-state workload (BASELINE.json configs[2] shape): bench line, kernel trace, counters of the matrix-core kernel
AA="--workload aa --tips 2000 --width 500 --read-len 100 --chunk 50000 --pool 3"   # BASELINE configs[2] at its own size (round 4; rounds 1-3: 10000)
python bench.py $AA > gpurun_out/${tag}_aa_bench.json 2> gpurun_out/${tag}_aa_bench.err
tail -1 gpurun_out/${tag}_aa_bench.json | cut -c1-300
cd /tmp
rm -rf $R/gpurun_out/${tag}_aa_stats
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_aa_stats -o st -- \
  python $R/bench.py $AA --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_aa_stats.log 2>&1
cd $R
python profiles/db_to_txt.py gpurun_out/${tag}_aa_stats/st_results.db > gpurun_out/${tag}_aa_kernel_trace_stats.txt
head -8 gpurun_out/${tag}_aa_kernel_trace_stats.txt
BENCH_ARGS="$AA" bash profiles/run_pmc.sh ${tag}_aa \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" > gpurun_out/${tag}_aa_pmc_summary.txt 2>&1
grep -A16 "== k_thorough" gpurun_out/${tag}_aa_pmc_summary.txt | head -20
# traffic file for bench.py (stamped with the kernel-source hash)
python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
a = json.loads(open("gpurun_out/${tag}_aa_bench.json").read().strip().splitlines()[-1])
import subprocess
subprocess.check_call(["python", "profiles/make_traffic.py", "${tag}", str(d["config"]["reads_per_step_per_gpu"]),
                       str(d["roofline"]["pairs_per_launch"]), str(a["config"]["reads_per_step_per_gpu"]), str(a["roofline"]["pairs_per_launch"])])
PY
if [ -n "$EPA_ROUND_SHORT" ]; then rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_aa_stats gpurun_out/${tag}_pmc_*/ gpurun_out/${tag}_aa_pmc_*/; exit 0; fi   # EPA_ROUND_SHORT=1: without the 1e9-pair cfg5 trace and the FMA clock microbenchmark
# cfg5 at per-GPU shard size (BASELINE configs[4]: 4k tips, 1M reads, --no-heur on 8 GPUs = 125k reads per GPU):
# one epa_dev_place_all call over 125 000 reads x 7997 branches = 1e9 pairs, kernel trace committed
cd /tmp
rm -rf $R/gpurun_out/${tag}_cfg5_stats
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_cfg5_stats -o st -- \
  python $R/bench.py --workload cfg5 --chunk 125000 --steps 1 --warmup 0 --parity-sample 0 > $R/gpurun_out/${tag}_cfg5.log 2>&1
cd $R
python profiles/db_to_txt.py gpurun_out/${tag}_cfg5_stats/st_results.db > gpurun_out/${tag}_cfg5_kernel_trace_stats.txt
grep "^{" gpurun_out/${tag}_cfg5.log | cut -c1-600; head -6 gpurun_out/${tag}_cfg5_kernel_trace_stats.txt
# the clock under a bare fp64 FMA stream (profiles/fma_clock.hip)
hipcc --offload-arch=gfx950 -O3 profiles/fma_clock.hip -o /tmp/fma_clock && /tmp/fma_clock > gpurun_out/${tag}_fma_clock.txt 2>&1
cat gpurun_out/${tag}_fma_clock.txt
