#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh'): the bench line, the rocprofv3 kernel trace of the
# same command and the two HBM-traffic PMC passes, into gpurun_out/prof_final/.  Summaries are then copied to
# profiles/<round>/ by hand (tools/summarize_profiles.py makes the JSON / trimmed CSV).
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_final
rm -rf $O
mkdir -p $O
echo "[1/5] bench (default flags)"; python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "[2/5] bench fp32"; python bench.py --precision fp32 --cpu-rays-side 0 > $O/bench_fp32.json 2> $O/bench_fp32.err
# the profiled runs skip the side legs (other arithmetic mode, early termination): the last launches of every kernel
# are then the timed ones
echo "[3/7] kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 4 --warmup 1 --cpu-rays-side 0 \
    --no-other-mode > $O/bench_under_rocprof.json 2> $O/trace.err
for mode in bf16x3 fp32; do
  echo "[PMC] FETCH_SIZE $mode"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$mode -- python bench.py --steps 2 --warmup 1 \
      --cpu-rays-side 0 --no-other-mode --precision $mode > $O/bench_pmc_fetch_$mode.json 2> $O/pmc_fetch_$mode.err
  echo "[PMC] WRITE_SIZE $mode"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$mode -- python bench.py --steps 2 --warmup 1 \
      --cpu-rays-side 0 --no-other-mode --precision $mode > $O/bench_pmc_write_$mode.json 2> $O/pmc_write_$mode.err
done
echo done; ls -R $O | head -40
