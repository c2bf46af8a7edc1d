#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: an experimental build of the library as exp/<name>.so
# (loaded with CRAY_LIB=exp/<name>.so; exp/ is git-ignored but travels to the GPU box)
set -e
cd "$(dirname "$0")/../craytracer_amd/csrc"
name=$1; shift
mkdir -p ../../exp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function "$@" -o ../../exp/$name.so cray_hip.hip cray_host.cpp cray_cry.cpp cray_io.cpp cray_image.cpp
echo built exp/$name.so
