#!/bin/bash
# Builds the library of the last commit as ab/libA.so next to the working-tree build (ab/libB.so) so that one
# gpurun call can time both on the same box:  BABELFDTD_HIP_LIB=$PWD/ab/libA.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
tmp=$(mktemp -d)
git archive ${AB_BASE:-HEAD} babelbrain_amd/csrc include | tar -x -C "$tmp"
make -s -C "$tmp/babelbrain_amd/csrc" > /dev/null
cp "$tmp/babelbrain_amd/libbabelfdtd_hip.so" ab/libA.so
rm -rf "$tmp"
make -s -C babelbrain_amd/csrc > /dev/null
cp babelbrain_amd/libbabelfdtd_hip.so ab/libB.so
ls -la ab
