#!/bin/bash
# Round 5: the evidence passes in one go (GPU box, from the repo root).  Output under gpurun_out/r05/; what is judged is copied to profiles/r05/.
set -u
OUT=gpurun_out/r05; mkdir -p $OUT
ROOT=$(pwd)
python tools/probes/slam_ba_twin.py 60 > $OUT/adjuster_twin_60.log 2>/dev/null
python tools/probes/slam_ba_stamps.py 60 2>/dev/null | tail -2 > $OUT/adjuster_phase_stamps_60.json
python tools/probes/icl_seed_study.py 200 8 2>/dev/null | tail -1 > $OUT/icl_seed_study_200.json
python tools/probes/icl_seed_study.py 80 8 2>/dev/null | tail -1 > $OUT/icl_seed_study_80.json
python tools/probes/loop_ba_seed_study.py 2>/dev/null > $OUT/loop_ba_seed_study.json
bash tools/profile_loop_round.sh $OUT/loop_prof > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt_icl200" -- python3 "$ROOT/tools/run_icl_nuim.py" 200 --ba > "$ROOT/$OUT/icl_200_ba_under_rocprof.json" 2> /dev/null )
find $OUT/kt_icl200 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/icl_200_ba_kernel_stats.csv; rm -rf $OUT/kt_icl200
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$ROOT/$OUT/pmc_shape" -- "$ROOT/tools/probes/mfma_shape_mix" 2048 10 > "$ROOT/$OUT/mfma_shape_mix_under_pmc.json" 2> /dev/null )
python3 tools/pmc_summary.py $OUT/pmc_shape > $OUT/mfma_shape_mix_clock.json 2>/dev/null; rm -rf $OUT/pmc_shape
./tools/probes/mfma_shape_mix 2048 20 > $OUT/mfma_shape_mix.json
bash tools/profile_round.sh $OUT/prof > /dev/null 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cp bench_details.json $OUT/bench_details.json
ls -la $OUT $OUT/prof $OUT/loop_prof
