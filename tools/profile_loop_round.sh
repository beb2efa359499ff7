#!/bin/bash
# Kernel tables of the device-resident loop (plain, and with the bundle adjustment per keyframe + re-association), for profiles/rNN:
#   gpurun -- 'bash tools/profile_loop_round.sh gpurun_out/r04_loop'
set -u
OUT=${1:-gpurun_out/loop_prof}
ROOT=$(pwd)
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt_plain" -- python3 "$ROOT/tools/run_slam_loop.py" 60 --device > "$ROOT/$OUT/loop_plain_under_rocprof.json" 2> "$ROOT/$OUT/kt_plain.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt_ba" -- python3 "$ROOT/tools/run_slam_loop.py" 60 --device --ba --reassociate > "$ROOT/$OUT/loop_ba_under_rocprof.json" 2> "$ROOT/$OUT/kt_ba.err"
cd "$ROOT"
find "$OUT/kt_plain" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/loop_plain_kernel_stats.csv"
find "$OUT/kt_ba" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/loop_ba_kernel_stats.csv"
rm -rf "$OUT"/kt_plain "$OUT"/kt_ba
