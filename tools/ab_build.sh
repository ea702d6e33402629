#!/bin/bash
# Builds the library of a git ref (or of the working tree with "WORK") into build_ab/<name>.so for tools/ab_bench.py.
# usage: bash tools/ab_build.sh <ref|WORK> <name> [extra hipcc flags]
set -e
REF=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build_ab
TMP=$(mktemp -d)
if [ "$REF" = "WORK" ]; then
  mkdir -p $TMP/bez_isaacgym_amd $TMP/include
  cp -r $ROOT/bez_isaacgym_amd/csrc $TMP/bez_isaacgym_amd/; cp $ROOT/include/bez_sim.h $TMP/include/
else
  git -C $ROOT archive $REF bez_isaacgym_amd/csrc include | tar -x -C $TMP
fi
pids=""
for f in $TMP/bez_isaacgym_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c -o ${f%.hip}.o $f 2>/dev/null &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_ab/$NAME.so $TMP/bez_isaacgym_amd/csrc/*.o
rm -rf $TMP
echo built build_ab/$NAME.so
