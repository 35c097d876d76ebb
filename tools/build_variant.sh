#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG=...]...   -> build_tmp/libnbk_<name>.so (same flags as numbotics_amd/csrc/build.py + the defines)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fno-gpu-rdc -Wall -Wno-unused-function -Wno-pass-failed "$@" numbotics_amd/csrc/nbk.hip -o build_tmp/libnbk_$name.so
echo build_tmp/libnbk_$name.so
