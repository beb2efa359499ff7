#!/bin/bash
# The profiler passes whose summaries are committed under profiles/rNN (run on the GPU box, from the repo root):
#   tools/profile_round.sh gpurun_out/r03_prof
# 1. kernel trace + stats of the bench line's command on ONE stream (clean per-kernel durations)
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes over tools/bench_ba.py and tools/bench_tri.py
# 3. vector-issue counters of the BA kernels
# Counter passes never combine --pmc with the hip / hsa / memory-copy trace domains.
set -u
OUT=${1:-gpurun_out/prof}
ROOT=$(pwd)
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt" -- python3 "$ROOT/bench.py" --one-stream --steps 30 --warmup 5 \
    --no-cpu-baseline --no-shard-proxy --no-replay --no-frontend --no-asymptote > "$ROOT/$OUT/bench_one_stream_under_rocprof.json" 2> "$ROOT/$OUT/kt.err"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_${c}_ba" -- python3 "$ROOT/tools/bench_ba.py" > /dev/null 2> "$ROOT/$OUT/pmc_${c}_ba.err"
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_${c}_tri" -- python3 "$ROOT/tools/bench_tri.py" > /dev/null 2> "$ROOT/$OUT/pmc_${c}_tri.err"
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d "$ROOT/$OUT/pmc_valu_ba" -- python3 "$ROOT/tools/bench_ba.py" > /dev/null 2> "$ROOT/$OUT/pmc_valu_ba.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d "$ROOT/$OUT/pmc_valu_tri" -- python3 "$ROOT/tools/bench_tri.py" > /dev/null 2> "$ROOT/$OUT/pmc_valu_tri.err"
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT/pmc_FETCH_SIZE_ba" "$OUT/pmc_WRITE_SIZE_ba" "$OUT/pmc_FETCH_SIZE_tri" "$OUT/pmc_WRITE_SIZE_tri" > "$OUT/pmc_hbm_traffic_summary.json"
python3 tools/pmc_summary.py "$OUT/pmc_valu_ba" "$OUT/pmc_valu_tri" > "$OUT/pmc_valu_summary.json"
find "$OUT/kt" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/bench_one_stream_kernel_stats.csv"
# keep the merge small: the raw traces stay on the box
rm -rf "$OUT"/kt "$OUT"/pmc_FETCH_SIZE_* "$OUT"/pmc_WRITE_SIZE_* "$OUT"/pmc_valu_ba "$OUT"/pmc_valu_tri
ls -la "$OUT"
