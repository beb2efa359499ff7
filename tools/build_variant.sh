#!/bin/bash
# A/B build of the library with extra defines on ONE translation unit:
#   tools/build_variant.sh NAME ba.hip -DMQS_WL_SCALAR_CAMS=0   ->  build/ab/libmqslam_NAME.so   (run with MQS_LIB_PATH=...)
set -eu
NAME=$1; TU=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/multiple-quadrotor-slam_amd
mkdir -p "$ROOT/build/ab"
OBJ=$ROOT/build/ab/${TU%.hip}_$NAME.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c -o "$OBJ" "$PKG/csrc/$TU"
OTHERS=$(ls "$ROOT"/build/obj/*.o | grep -v "/${TU%.hip}.o")
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -Wl,-rpath,/opt/rocm/lib -o "$ROOT/build/ab/libmqslam_$NAME.so" "$OBJ" $OTHERS -L/opt/rocm/lib
echo "$ROOT/build/ab/libmqslam_$NAME.so"
