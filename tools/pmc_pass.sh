#!/bin/bash
# Counter passes of one benchmark script, one rocprofv3 run per counter set (kernel-trace only beside --pmc: the pool refuses
# --pmc together with the hip / hsa trace domains), summarised per kernel by tools/pmc_summary.py.
#   gpurun -- 'bash tools/pmc_pass.sh OUT.json tools/bench_ba.py "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQC_ICACHE_HITS SQC_ICACHE_MISSES" ...'
# The summary lands in gpurun_out/OUT.json.  Extra arguments for the script: MQS_PMC_ARGS="125000 4".
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the copy on the GPU box)}"
R=$GRAFT_REPO_ROOT
OUT=$1; BENCH=$2; shift 2
W=$R/gpurun_out/pmcp_$$
mkdir -p "$W"
cd /tmp && export TMPDIR=/tmp
k=0
for set in "$@"; do
  k=$((k + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$W/set$k" -- python3 "$R/$BENCH" ${MQS_PMC_ARGS:-} > /dev/null 2> "$R/gpurun_out/pmcp_set$k.err" || true
done
cd "$R"
python3 tools/pmc_summary.py "$W"/* > "gpurun_out/$OUT"
rm -rf "$W"
