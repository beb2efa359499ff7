#!/bin/bash
# The evidence passes of the example-sequence / loop studies in one go (needs tests/golden/icl_nuim_traj3n/sequence_200.npz:
# `python tests/golden/make_icl_nuim.py 200`, moved aside).  Output under gpurun_out/final/.
set -u
OUT=gpurun_out/final; mkdir -p $OUT
bash tools/probes/icl_nuim_seeds.sh > $OUT/icl_seeds.txt 2>&1
python tools/probes/icl_many_seeds.py 32 2>/dev/null | tail -1 > $OUT/icl_many_seeds.json
bash tools/probes/icl_screen_200.sh > $OUT/icl_screen_200.txt 2>&1
bash tools/probes/icl_window.sh > $OUT/icl_window.txt 2>&1
python tools/probes/loop_ba_seed_study.py 2>/dev/null > $OUT/loop_ba_seed_study.json
python tests/probe_icl_match_step.py 2>/dev/null | tail -4 > $OUT/match_step.txt
python tools/probes/icl_ba_timing.py 2>/dev/null | tail -2 > $OUT/icl_ba_timing.txt
bash tools/profile_loop_round.sh $OUT/loop_prof > /dev/null 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
