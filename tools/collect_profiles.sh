#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh [what]'): the bench line, the rocprofv3 kernel trace of
# the same command, the HBM-traffic PMC passes and the MFMA-busy PMC pass, into gpurun_out/prof_rNN/.
# tools/summarize_profiles.py then condenses them into profiles/<round>/ (run on the host).
#   what = all (default) | bench | trace | pmc | cfg2 (the lego-like configuration's bench line, fp32)
#          | cfg4 (trace + PMC passes of the K = 12 configuration, fp32)
#          | train (per-kernel times and a PMC pass of the training step, tools/train_step_bench.py, 4096 and 65 536 rays)
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WHAT=${1:-all}
O=${PROF_DIR:-gpurun_out/prof_r04}
mkdir -p $O
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  echo "[bench] default flags (fp32 headline, all side legs, CPU baseline)"
  python bench.py > $O/bench_default.json 2> $O/bench_default.err
fi
# the profiled runs skip the side legs (other arithmetic mode, early termination, training step, CPU baseline): the
# last launches of every kernel are then the timed ones
if [ "$WHAT" = all ] || [ "$WHAT" = trace ]; then
  for mode in fp32 bf16x3; do
    echo "[trace] kernel trace $mode"
    rm -rf $O/trace_$mode
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -- python bench.py --steps 4 --warmup 1 \
        --cpu-rays-side 0 --no-other-mode --precision $mode > $O/bench_under_rocprof_$mode.json 2> $O/trace_$mode.err
  done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  for mode in fp32 bf16x3; do
    for counter in FETCH_SIZE WRITE_SIZE; do
      echo "[pmc] $counter $mode"
      rm -rf $O/pmc_${counter}_$mode
      rocprofv3 --pmc $counter --output-format csv -d $O/pmc_${counter}_$mode -- python bench.py --steps 2 --warmup 1 \
          --cpu-rays-side 0 --no-other-mode --precision $mode > $O/bench_pmc_${counter}_$mode.json 2> $O/pmc_${counter}_$mode.err
    done
    echo "[pmc] MFMA busy / clock $mode"
    rm -rf $O/pmc_mfma_$mode
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_mfma_$mode -- python bench.py \
        --steps 3 --warmup 1 --cpu-rays-side 0 --no-other-mode --precision $mode > $O/bench_pmc_mfma_$mode.json 2> $O/pmc_mfma_$mode.err
  done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = cfg2 ]; then
  echo "[cfg2] bench line (lego-like: max_o 830000, P 9)"
  python bench.py --config cfg2_lego_6m --cpu-rays-side 0 --no-other-mode --precision fp32 --steps 4 --warmup 1 \
      > $O/bench_cfg2.json 2> $O/bench_cfg2.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = cfg4 ]; then
  C="--config cfg4_scannet_20m --cpu-rays-side 0 --no-other-mode --precision fp32"
  echo "[cfg4] bench line"
  python bench.py $C --steps 4 --warmup 1 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
  echo "[cfg4] kernel trace"
  rm -rf $O/trace_cfg4
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg4 -- python bench.py $C --steps 3 --warmup 1 \
      > $O/bench_under_rocprof_cfg4.json 2> $O/trace_cfg4.err
  for counter in FETCH_SIZE WRITE_SIZE; do
    echo "[cfg4] pmc $counter"
    rm -rf $O/pmc_${counter}_cfg4
    rocprofv3 --pmc $counter --output-format csv -d $O/pmc_${counter}_cfg4 -- python bench.py $C --steps 2 --warmup 1 \
        > $O/bench_pmc_${counter}_cfg4.json 2> $O/pmc_${counter}_cfg4.err
  done
  echo "[cfg4] pmc MFMA busy"
  rm -rf $O/pmc_mfma_cfg4
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_mfma_cfg4 -- python bench.py $C \
      --steps 2 --warmup 1 > $O/bench_pmc_mfma_cfg4.json 2> $O/pmc_mfma_cfg4.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = train ]; then
  for rays in 4096 65536; do
    echo "[train] kernel times $rays rays"
    bash tools/train_kernel_times.sh $rays fp32 > $O/train_kernel_times_$rays.txt 2>&1
    cp gpurun_out/tkt.json $O/train_step_under_rocprof_$rays.json
    python3 tools/train_step_bench.py --skip-autograd --steps 10 --rays $rays > $O/train_step_$rays.json 2> $O/train_step_$rays.err
  done
  echo "[train] pmc 65536 rays"
  bash tools/pmc_train.sh 65536 > $O/pmc_train_65536.txt 2>&1
fi
echo done; ls $O
