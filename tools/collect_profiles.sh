#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats and two separate PMC passes
# of the default bench command, raw output under gpurun_out/prof_*; tools/summarize_profiles.py then
# condenses them into profiles/.  Usage: bash tools/collect_profiles.sh [steps]
set -e
STEPS=${1:-3}
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python bench.py --steps $STEPS --warmup 1 > $OUT/prof_bench.json 2> $OUT/prof_bench.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o run -- python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_write.log 2>&1
if [ "${WITH_KLD:-0}" = "1" ]; then
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kld -o run -- python3 tools/kld_bench.py 2500 > $OUT/prof_kld.log 2>&1
python tools/throughput_bench.py 64 512 16384 131072 > $OUT/prof_throughput.log 2>&1
fi
# large-blanket path (GLC Dense on sphere.g2o at full size): kernel stats of the dense pipeline
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_big -o run -- python3 tools/big_bench.py > $OUT/prof_big.log 2>&1 || true
# block-sparse path at the headline size (optimize after marginalisation, global KLD) and the interior-point kernel
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sparse -o run -- python3 tools/sparse_bench.py 100000 400 all > $OUT/prof_sparse.log 2>&1 || true
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ip -o run -- python3 tools/ip_bench.py > $OUT/prof_ip.log 2>&1 || true
# GLC Tree on the bench graph and BASELINE config 4 (parking.g2o, NFR Tree): kernel stats behind the figures quoted in DESIGN.md
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_glc -o run -- python3 tools/glc_check.py > $OUT/prof_glc.log 2>&1 || true
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_parking -o run -- python3 bench.py --config parking --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_parking.log 2>&1 || true
# clusters of the generic NFR kernel: sphere.g2o and parking.g2o under CliqueyDense at full size
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cluster -o run -- python3 tools/cluster_bench.py sphere parking > $OUT/prof_cluster.log 2>&1 || true
find $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write -name '*.csv' | head -20
