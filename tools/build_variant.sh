#!/bin/bash
# Diagnostic: builds pointnerf2studio_amd/_abl/libpnr_<name>.so with extra compiler flags on top of the product's
# (pointnerf2studio_amd/build.py: FLAGS + FILE_FLAGS), e.g.
#   tools/build_variant.sh xc4 -DPNR_XCD_CHUNK=4
set -e
cd "$(dirname "$0")/.."
name=$1; shift
C=pointnerf2studio_amd/csrc
O=pointnerf2studio_amd/_abl/obj_$name
mkdir -p $O
pids=()
for f in pnr_scan pnr_scene pnr_query pnr_shade pnr_shade_fp32 pnr_shade_bf16 pnr_render pnr_train pnr_train_chain pnr_optim; do
  extra=""
  if { [ $f = pnr_shade_fp32 ] || [ $f = pnr_train_chain ]; } && [ -z "$PNR_NO_FILE_FLAGS" ]; then extra="-mllvm -amdgpu-mfma-vgpr-form"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -pragma-unroll-threshold=4000000 \
    $extra "$@" -Iinclude -I$C -c $C/$f.hip -o $O/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pointnerf2studio_amd/_abl/libpnr_$name.so $O/*.o
rm -rf $O
ls -la pointnerf2studio_amd/_abl/libpnr_$name.so
