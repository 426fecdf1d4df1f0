#!/bin/bash
# AddressSanitizer + UBSan run of the HOST code (scheduler, graph update, I/O, drivers) on the CPU: the HIP
# translation units are replaced by stubs (tools/asan_stubs.cpp), the arithmetic by the injected oracle
# backend of the CPU tests. GPU sanitizers are not available on the test pool.   Usage: bash tools/sanitize_host.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/spg_asan
mkdir -p $OUT
g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/sparsifyposegraph_amd/csrc \
    -shared -pthread -o $OUT/libspg_host_asan.so $ROOT/sparsifyposegraph_amd/csrc/spg_host.cpp $ROOT/tools/asan_stubs.cpp
cd $ROOT
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
UBSAN_OPTIONS=print_stacktrace=1 SPG_LIB_PATH=$OUT/libspg_host_asan.so \
python -m pytest tests/test_host_scheduler.py tests/test_stream_scheduler.py tests/test_decimation_and_io.py tests/test_substitute_edge.py tests/test_distributed.py tests/test_cliquey.py -x -q -m "not gpu"
