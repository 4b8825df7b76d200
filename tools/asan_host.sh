#!/bin/bash
# CPU-only memory-safety check of the host interpreter (GPU AddressSanitizer is not available on the pool):
# gfh_main.cpp + headers built with -fsanitize=address,undefined, driven over the committed .sgcl programs with the
# CPU oracle as the TaylorPoly backend.  Usage: bash tools/asan_host.sh   (from the repo root; needs oracle/liborc.so)
set -e
ROOT=$(pwd)
mkdir -p /tmp/gft_asan
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off \
    -I"$ROOT/genfer_amd/csrc/host" -I"$ROOT/genfer_amd/csrc" tools/asan_host_main.cpp genfer_amd/csrc/host/gfh_main.cpp -ldl -o /tmp/gft_asan/asan_host
FILES=$(ls tests/golden/sgcl/test_expect/*/*.sgcl | grep -v /slow/; ls tests/golden/sgcl/neurips2023/exact/*/*.sgcl \
        tests/golden/sgcl/neurips2023/approx/population*/*.sgcl tests/golden/sgcl/example.sgcl)
ASAN_OPTIONS=detect_leaks=1 /tmp/gft_asan/asan_host "$ROOT/oracle/liborc.so" orc_ $FILES 2>&1 | grep -v "EXPERIMENTAL" | tail -5
