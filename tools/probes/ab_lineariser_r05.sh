#!/bin/bash
# Round-5 A/B of the dominant kernel (ba_linearize_wave_kernel<4, true>): variants of the landmarks-per-lane / stash / occupancy geometry
# on the CURRENT tree, built side by side and timed interleaved in ONE job (boxes differ by 10 %).   bash tools/probes/ab_lineariser_r05.sh build|run
set -e
cd "$(dirname "$0")/../.."
PKG=multiple-quadrotor-slam_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
declare -A V
V[base]=""
V[l3]="-DMQS_WL_MAXL=3 -DMQS_WL_LDSL=3 -DMQS_WL_OCC=1"
V[l3s2]="-DMQS_WL_MAXL=3 -DMQS_WL_LDSL=2 -DMQS_WL_OCC=1"
V[l2]="-DMQS_WL_MAXL=2 -DMQS_WL_LDSL=1 -DMQS_WL_OCC=1"
V[occ2]="-DMQS_WL_MAXL=2 -DMQS_WL_LDSL=1 -DMQS_WL_OCC=2 -DMQS_WL_SCALAR_CAMS=0"
if [ "$1" = build ]; then
    mkdir -p build/ab
    for v in "${!V[@]}"; do
        ( /opt/rocm/bin/hipcc $FLAGS ${V[$v]} -c -o build/ab/ba_$v.o $PKG/csrc/ba.hip 2>build/ab/ba_$v.log &&
          objs=$(ls build/obj/*.o | grep -v '/ba.o') &&
          /opt/rocm/bin/hipcc $FLAGS -shared -o build/ab/libmqslam_r05_$v.so build/ab/ba_$v.o $objs -L/opt/rocm/lib && echo built $v ) || echo "FAILED $v: $(tail -3 build/ab/ba_$v.log)"
    done
else
    for round in 1 2 3; do
        for v in base l3 l3s2 l2 occ2; do
            [ -f build/ab/libmqslam_r05_$v.so ] && MQS_LIB_PATH=$PWD/build/ab/libmqslam_r05_$v.so python tools/ab_lin.py 1000000 4 2 200 2>/dev/null | tail -1
        done
    done
fi
