#!/bin/bash
# Profiles of the headline bench on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats, then separate PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hit) and the
#   FETCH_SIZE calibration launches.  Results land under gpurun_out/prof_<tag>/.
set -o pipefail
tag=${1:-r01}
wl=${2:-s100m}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-rmat-leg > $out/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-rmat-leg > $out/bench_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-rmat-leg > $out/bench_write.log 2>&1 || exit 3
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2 -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-rmat-leg > $out/bench_l2.log 2>&1 || exit 4
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- python3 tools/pmc_calibrate.py > $out/cal_fetch.log 2>&1 || exit 5
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- python3 tools/pmc_calibrate.py > $out/cal_write.log 2>&1 || exit 6
find $out -name "*.csv" | head -40
du -sh $out
