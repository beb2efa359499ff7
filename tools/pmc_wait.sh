cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_WAIT[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_INSTS_[A-Z_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/sq_counters.txt
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcw/$n -- python3 $R/tools/bench_ba.py > /dev/null 2> $R/gpurun_out/pmcw_$n.err
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmcw/* > gpurun_out/pmc_wait_summary.json
rm -rf gpurun_out/pmcw
