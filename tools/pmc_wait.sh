#!/bin/bash
# SQ wait / issue accounting of the BA kernels (separate --pmc passes, kernel-trace only: the pool refuses --pmc with
# the hip/hsa trace domains).  Run on the GPU box:  gpurun -- 'bash tools/pmc_wait.sh [bench script] [out name]'
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the copy on the GPU box)}"
R=$GRAFT_REPO_ROOT
BENCH=${1:-tools/bench_ba.py}
OUT=${2:-pmc_wait_summary.json}
W=$R/gpurun_out/pmcw
mkdir -p "$W"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_WAIT[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_INSTS_[A-Z_]*" | sort -u | tr '\n' ' ' > "$R/gpurun_out/sq_counters.txt" || true
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  n=$(echo "$set" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$W/$n" -- python3 "$R/$BENCH" > /dev/null 2> "$R/gpurun_out/pmcw_$n.err" || true
done
cd "$R"
python3 tools/pmc_summary.py "$W"/* > "gpurun_out/$OUT"
rm -rf "$W"
