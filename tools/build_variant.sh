#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]   -- builds ray_tracing_amd/variants/<name>.so with extra compile flags
# (A/B experiments; run with RT_VARIANT_LIB=ray_tracing_amd/variants/<name>.so python3 tools/profile_driver.py ...)
set -e
cd "$(dirname "$0")/../ray_tracing_amd/csrc"
mkdir -p ../variants
NAME=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" -shared \
  -o ../variants/$NAME.so render.hip lbvh.hip lbvh_host.cpp refit.hip refit_host.cpp capi.cpp scene_builder.cpp flatten.cpp wide_bvh.cpp config.cpp
