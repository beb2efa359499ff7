#!/bin/bash
# Round-6 A/B of the matcher (csrc/match.hip): the row start values loaded a stage ahead, the stage fill through a buffer descriptor, its pieces
# spread over the stage's steps, the early reject and two query tiles per fragment read on the fp16 path, the FP4 reject per eight values --
# each switched off in turn against this tree, and all off (round 5's form);
# built side by side, timed interleaved in ONE job.   bash tools/probes/ab_matcher_r06.sh build|run
set -e
cd "$(dirname "$0")/../.."
PKG=multiple-quadrotor-slam_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
declare -A V
V[tree]=""
V[round5]="-DMQS_MATCH_TN_AHEAD=0 -DMQS_MATCH_BUFFER_DMA=0 -DMQS_MATCH_SPREAD_DMA=0 -DMQS_MATCH_SCHED=0 -DMQS_MATCH_PRUNE_F16=0 -DMQS_MATCH_F4_REJECT8=0 -DMQS_MATCH_F16_GROUP=1"
V[only_tn_ahead]="-DMQS_MATCH_TN_AHEAD=1 -DMQS_MATCH_BUFFER_DMA=0 -DMQS_MATCH_SPREAD_DMA=0"
V[no_spread]="-DMQS_MATCH_SPREAD_DMA=0"
V[no_buffer]="-DMQS_MATCH_BUFFER_DMA=0"
V[x_nobarrier]="-DMQS_MATCH_EXPERIMENT_NOBARRIER"     # timing experiment only (races): what the per-stage workgroup barrier costs
V[pf2]="-DMQS_MATCH_PF=2"
V[pf8]="-DMQS_MATCH_PF=8"
V[no_sched]="-DMQS_MATCH_SCHED=0"
V[no_prune16]="-DMQS_MATCH_PRUNE_F16=0"
V[f4_reject16]="-DMQS_MATCH_F4_REJECT16=1"
V[f16_group1]="-DMQS_MATCH_F16_GROUP=1"
V[f16_g2_pf1]="-DMQS_MATCH_F16_PF=1"
V[f16_g2_pf3]="-DMQS_MATCH_F16_PF=3"
V[f16_g2_pf4]="-DMQS_MATCH_F16_PF=4"
V[f4_reject4]="-DMQS_MATCH_F4_REJECT8=0"
V[f4_group4]="-DMQS_MATCH_F4_GROUP=4"
V[f4_group1]="-DMQS_MATCH_F4_GROUP=1"
V[stage64]="-DMQS_MATCH_STAGE_ROWS=64"
if [ "$1" = build ]; then
    mkdir -p build/ab6
    for v in ${ONLY:-"${!V[@]}"}; do
        ( /opt/rocm/bin/hipcc $FLAGS ${V[$v]} -c -o build/ab6/match_$v.o $PKG/csrc/match.hip 2>build/ab6/match_$v.log &&
          objs=$(ls build/obj/*.o | grep -v '/match.o') &&
          /opt/rocm/bin/hipcc $FLAGS -shared -o build/ab6/libmqslam_m6_$v.so build/ab6/match_$v.o $objs -L/opt/rocm/lib -pthread && echo built $v ) || echo "FAILED $v: $(tail -3 build/ab6/match_$v.log)"
    done
else
    for round in 1 2 3; do
        for v in ${ORDER:-tree round5 f16_group1 no_prune16 no_sched no_spread no_buffer f4_reject4}; do
            [ -f build/ab6/libmqslam_m6_$v.so ] && MQS_LIB_PATH=$PWD/build/ab6/libmqslam_m6_$v.so python tools/ab_match.py 65536 2 20 2>/dev/null | tail -1
        done
    done
fi
