#!/bin/bash
# Diagnostic: builds pointnerf2studio_amd/_abl/libpnr_<name>.so with extra compiler flags, e.g.
#   tools/build_variant.sh xc4 -DPNR_XCD_CHUNK=4
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p pointnerf2studio_amd/_abl
C=pointnerf2studio_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -pragma-unroll-threshold=4000000 -shared "$@" \
  -Iinclude -I$C $C/pnr_scan.hip $C/pnr_scene.hip $C/pnr_query.hip $C/pnr_shade.hip $C/pnr_shade_fp32.hip $C/pnr_shade_bf16.hip $C/pnr_render.hip $C/pnr_train.hip \
  -o pointnerf2studio_amd/_abl/libpnr_$name.so
ls -la pointnerf2studio_amd/_abl/libpnr_$name.so
