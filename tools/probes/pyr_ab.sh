#!/bin/bash
# A/B of library builds on the device loop under rocprofv3: average time of the tracker's kernels and the decision kernel per build.
#   gpurun -- 'bash tools/probes/pyr_ab.sh main pyr512 ...'   (names of build/ab/libmqslam_NAME.so; `main` = the tree's library)
set -u
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/pyr_ab
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = main ]; then unset MQS_LIB_PATH; else export MQS_LIB_PATH=$ROOT/build/ab/libmqslam_$v.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pyr_ab/$v -- python3 $ROOT/tools/run_slam_loop.py 60 --device > $ROOT/gpurun_out/pyr_ab/$v.json 2> $ROOT/gpurun_out/pyr_ab/$v.err
  f=$(find $ROOT/gpurun_out/pyr_ab/$v -name "*kernel_stats.csv" | head -1)
  python3 - "$v" "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("==", sys.argv[1], "all kernels per run: %.2f ms" % (tot / 1e6))
for r in rows:
    if any(k in r["Name"] for k in ("lk_pyramid", "pyr_level", "lk_pad", "lk_kernel", "frame_decide")):
        print("   %-40s calls %4s  avg %8.2f us" % (r["Name"].split("(")[0 if not r["Name"].startswith("(") else 2][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $ROOT/gpurun_out/pyr_ab/$v
done
