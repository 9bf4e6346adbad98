#!/bin/bash
# GPU box: per-kernel times of the training step (65536 rays) for library variants, one after the other on one device.
#   bash tools/ab_train_kernels.sh base nox0 ...   -> prints the top kernels per variant
for v in "$@"; do
  if [ $v = base ]; then export PNR_LIB=""; else export PNR_LIB="$PWD/pointnerf2studio_amd/_abl/libpnr_$v.so"; fi
  echo "== $v"
  bash tools/train_kernel_times.sh ${RAYS:-65536} fp32 | head -${TOP:-6}
done
