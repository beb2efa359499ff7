#!/bin/bash
# Round-6 A/B of the dominant kernel (ba_linearize_wave_kernel<4, true>): the window reduction's lane bits 3 and 2 as two DPP moves with
# complementary bank masks (this tree) against the select + exchange form of rounds 2-5 (-DMQS_WAVE_REDUCE_SELECTS), built side by side
# and timed interleaved in ONE job (boxes differ by 10 %).   bash tools/probes/ab_lineariser_r06.sh build|run
set -e
cd "$(dirname "$0")/../.."
PKG=multiple-quadrotor-slam_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
declare -A V
V[dpp_bank_masks]=""
V[selects]="-DMQS_WAVE_REDUCE_SELECTS"
if [ "$1" = build ]; then
    mkdir -p build/ab6
    for v in "${!V[@]}"; do
        ( /opt/rocm/bin/hipcc $FLAGS ${V[$v]} -c -o build/ab6/ba_$v.o $PKG/csrc/ba.hip 2>build/ab6/ba_$v.log &&
          objs=$(ls build/obj/*.o | grep -v '/ba.o') &&
          /opt/rocm/bin/hipcc $FLAGS -shared -o build/ab6/libmqslam_r06_$v.so build/ab6/ba_$v.o $objs -L/opt/rocm/lib -pthread && echo built $v ) || echo "FAILED $v: $(tail -3 build/ab6/ba_$v.log)"
    done
else
    for round in 1 2 3 4; do
        for v in dpp_bank_masks selects; do
            [ -f build/ab6/libmqslam_r06_$v.so ] && MQS_LIB_PATH=$PWD/build/ab6/libmqslam_r06_$v.so python tools/ab_lin.py 1000000 4 2 200 2>/dev/null | tail -1
        done
    done
fi
