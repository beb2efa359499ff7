#!/bin/bash
# Round 5, after the loop's per-frame work (null vector, result block, one-launch pyramid, register-resident DLT solves): the kernel tables of the
# device loop (plain, with the adjustment per keyframe), the example sequence with the adjustment, the decision kernel's phases, the bench line.
#   gpurun -- 'bash tools/probes/late_evidence_r05.sh'     (output: gpurun_out/r05b/; what is judged is copied to profiles/r05/)
set -u
OUT=gpurun_out/r05b; mkdir -p $OUT
ROOT=$(pwd)
timeout 400 bash tools/profile_loop_round.sh $OUT/loop_prof > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt_icl200" -- python3 "$ROOT/tools/run_icl_nuim.py" 200 --ba > "$ROOT/$OUT/icl_200_ba_under_rocprof.json" 2> /dev/null )
find $OUT/kt_icl200 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/icl_200_ba_kernel_stats.csv; rm -rf $OUT/kt_icl200
[ -f build/ab/libmqslam_stamps.so ] && MQS_LIB_PATH=build/ab/libmqslam_stamps.so python tools/probes/decide_phases.py 60 2>/dev/null | tail -1 > $OUT/decide_phases.json
python tools/run_slam_loop.py 60 --device --repeats 3 2>/dev/null | tail -1 > $OUT/loop_plain.json
python tools/run_slam_loop.py 60 --device --ba --reassociate --repeats 3 2>/dev/null | tail -1 > $OUT/loop_ba.json
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cp bench_details.json $OUT/bench_details.json
ls -la $OUT $OUT/loop_prof
