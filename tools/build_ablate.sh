#!/bin/bash
# Diagnostic builds of the library with -DPNR_ABLATE=<bits> (timing only: results are wrong by construction).
# bits: 1 gather/PE, 2 MFMAs, 4 epilogue, 8 barrier+DMA, 16 split, 32 DMA issue only, 64 barrier only
set -e
cd "$(dirname "$0")/.."
mkdir -p pointnerf2studio_amd/_abl
C=pointnerf2studio_amd/csrc
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -shared \
    -DPNR_ABLATE=$a -Iinclude -I$C $C/pnr_scan.hip $C/pnr_scene.hip $C/pnr_query.hip $C/pnr_shade.hip $C/pnr_shade_fp32.hip $C/pnr_shade_bf16.hip $C/pnr_render.hip \
    -o pointnerf2studio_amd/_abl/libpnr_abl$a.so &
done
wait
ls -la pointnerf2studio_amd/_abl/
