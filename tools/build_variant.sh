#!/bin/bash
# An A/B build of the product library with extra compiler flags, beside the shipped one (select it with OCT_PHMM_LIB=<path> in bench.py / tools/ / tests):
#   bash tools/build_variant.sh <name> [flags...]     ->  octopus_amd/variants/liboct_phmm_<name>.so   (git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/.."
mkdir -p octopus_amd/variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" octopus_amd/csrc/oct_phmm.hip -o octopus_amd/variants/liboct_phmm_$name.so
echo "built octopus_amd/variants/liboct_phmm_$name.so with: $*"
