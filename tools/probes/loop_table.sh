#!/bin/bash
# Kernel table (top rows) of the plain device loop under rocprofv3 for library builds:  gpurun -- 'bash tools/probes/loop_table.sh main [NAME ...]'
set -u
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/loop_table
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = main ]; then unset MQS_LIB_PATH; else export MQS_LIB_PATH=$ROOT/build/ab/libmqslam_$v.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/loop_table/$v -- python3 $ROOT/tools/run_slam_loop.py 60 --device > $ROOT/gpurun_out/loop_table/$v.json 2> $ROOT/gpurun_out/loop_table/$v.err
  f=$(find $ROOT/gpurun_out/loop_table/$v -name "*kernel_stats.csv" | head -1)
  cp $f $ROOT/gpurun_out/loop_table/${v}_kernel_stats.csv
  python3 - "$v" "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("==", sys.argv[1], "all kernels per run: %.2f ms" % (tot / 1e6))
for r in rows[:12]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("   %-34s calls %4s  avg %8.2f us  total %8.1f us" % (n.split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e3))
PY
  rm -rf $ROOT/gpurun_out/loop_table/$v
  tail -1 $ROOT/gpurun_out/loop_table/$v.json | cut -c1-60,280-420
done
