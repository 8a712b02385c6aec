#!/bin/bash
# two ranks on ONE GPU over the transport stand-in (plumbing evidence, not a scaling number)
python - <<'PY'
import os, sys
sys.path.insert(0, "tests")
import fake_rccl_util
print(fake_rccl_util.build())
PY
export EPA_RCCL_LIB=$PWD/tests/_build/libfake_rccl.so EPA_FAKE_RCCL_TIMEOUT_S=120 EPA_BENCH_BACKEND=gloo EPA_BENCH_ONE_GPU=1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 WORLD_SIZE=2 LOCAL_RANK=0
RANK=1 python bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline --strong-reads 2000000 > /tmp/r1.log 2>&1 &
RANK=0 python bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline --strong-reads 2000000 > gpurun_out/r6_bench_n2_standin.json 2> /tmp/r0.err
wait
tail -2 /tmp/r0.err; tail -2 /tmp/r1.log | cut -c1-300
