#!/bin/bash
# Everything the committed profiles/<tag>_* files come from, in one gpurun call:
#   bash profiles/run_round.sh <tag>
tag=${1:-r1}
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
head -12 gpurun_out/${tag}_kernel_trace_stats.txt
bash profiles/run_pmc.sh ${tag} \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" > gpurun_out/${tag}_pmc_summary.txt 2>&1
grep -A22 "== k_thorough" gpurun_out/${tag}_pmc_summary.txt | head -30
