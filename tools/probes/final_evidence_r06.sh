#!/bin/bash
# Round 6: the evidence passes in one go (GPU box, from the repo root).  Output under gpurun_out/r06e/; what is judged is copied to profiles/r06/.
#   gpurun --timeout 2400 -- 'bash tools/probes/final_evidence_r06.sh'
set -u
OUT=gpurun_out/r06e; mkdir -p $OUT
ROOT=$(pwd)
# 1. the selection form of the in-loop adjustment: seeds, legs of the residual screen, per-adjustment phases, frame ingest
timeout 600 python tools/probes/selection_seed_study.py 16 > $OUT/selection_seed_study.jsonl 2>/dev/null
timeout 600 python tools/probes/screen_legs_study.py 16 > $OUT/screen_legs_study.jsonl 2>/dev/null
timeout 200 python tools/probes/icl_selection_stamps.py 200 3 0 > $OUT/selection_phase_stamps_icl_200.jsonl 2>/dev/null
timeout 600 python tools/probes/ingest_study.py 3 > $OUT/ingest_study.jsonl 2>/dev/null
# 2. kernel tables of the loops under rocprofv3 (plain / adjustment per keyframe; rendered 60 frames, the reference's 200 example frames)
timeout 400 bash tools/profile_loop_round.sh $OUT/loop_prof > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt_icl200" -- python3 "$ROOT/tools/run_icl_nuim.py" 200 --ba > "$ROOT/$OUT/icl_200_ba_under_rocprof.json" 2> /dev/null )
find $OUT/kt_icl200 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/icl_200_ba_kernel_stats.csv; rm -rf $OUT/kt_icl200
# 3. the step's kernels: one-stream kernel table, HBM traffic and vector-issue counters (separate --pmc passes)
timeout 900 bash tools/profile_round.sh $OUT/prof > /dev/null 2>&1
# 4. the matcher under the profiler (kernel table of one fp16 + one FP4 pair)
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt_match" -- python3 "$ROOT/tools/ab_match.py" 65536 2 20 > "$ROOT/$OUT/matcher_under_rocprof.json" 2> /dev/null )
find $OUT/kt_match -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/matcher_kernel_stats.csv; rm -rf $OUT/kt_match
# 5. the bench line
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cp bench_details.json $OUT/bench_details.json
ls -la $OUT $OUT/prof $OUT/loop_prof
