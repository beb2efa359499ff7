#!/bin/bash
# What do SQ_WAIT_ANY / SQ_ACTIVE_INST_VALU read on instruction streams that contain NO s_waitcnt at all?  The class-by-class
# issue probe (tools/probes/valu_issue.hip: straight-line VALU streams, one and two waves per SIMD) under the SQ counters, and
# GRBM_GUI_ACTIVE for the clock the chip holds under each stream.  gpurun -- 'bash tools/probes/valu_issue_pmc.sh'
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun}"
R=$GRAFT_REPO_ROOT
W=$R/gpurun_out/vip_$$
mkdir -p "$W"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$W/set1" -- "$R/build/valu_issue" 100000 > "$R/gpurun_out/valu_issue_under_pmc.json" 2> "$R/gpurun_out/vip.err" || true
cd "$R"
python3 tools/pmc_summary.py "$W"/* > gpurun_out/valu_issue_pmc_summary.json
rm -rf "$W"
