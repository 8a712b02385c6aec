#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (profiles/calib_traffic.hip); run through gpurun:
#   bash profiles/run_calib.sh > gpurun_out/r2_traffic_calibration.txt
R=$PWD
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 profiles/calib_traffic.hip -o /tmp/calib_traffic || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/calib_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/calib_$c -o cal -- /tmp/calib_traffic > /tmp/calib_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob
known = 2 << 30
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/calib_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0]
        acc.setdefault(k, []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        kb = sum(v) / len(v)
        print("%-10s %-10s counter %.1f KB per launch = %.3f x the %d bytes the kernel moves" % (c, k, kb, kb * 1024.0 / known, known))
PY
