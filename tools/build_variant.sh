#!/bin/bash
# usage: bash tools/build_variant.sh NAME FILE.hip "-DFLAG=1 ..."   -> ab_libs/NAME.so (FILE rebuilt with the flags, other objects reused)
set -e
NAME=$1; FILE=$2; FLAGS=$3
HERE=$(cd "$(dirname "$0")/.." && pwd)
C=$HERE/rayuela.jl_amd/csrc
mkdir -p $HERE/ab_libs
OBJ=$HERE/ab_libs/$NAME.${FILE%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable $FLAGS -c $C/$FILE -o $OBJ
OTHERS=$(ls $C/*.o | grep -v "/${FILE%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ $OTHERS -ldl -o $HERE/ab_libs/$NAME.so
echo built ab_libs/$NAME.so
