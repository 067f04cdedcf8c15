#!/bin/bash
# Build libdicow_hip.so for gfx950 in-tree (cross-compiles without a GPU).  Usage: build.sh [--exp] [extra hipcc flags]
#   --exp : build the EXPERIMENTS library instead (../libdicow_hip_exp.so, objects in build_exp/, -DDICOW_EXPERIMENTS): the stable ABI
#           plus the experimental entry points of include/dicow_hip.h (tests/test_gpu_lnfold.py and the A/B tools load it explicitly)
set -e
cd "$(dirname "$0")"
OUT=../libdicow_hip.so
BD=build
if [ "$1" = "--exp" ]; then shift; OUT=../libdicow_hip_exp.so; BD=build_exp; set -- -DDICOW_EXPERIMENTS "$@"; fi
SRCS=$(ls *.hip)
mkdir -p $BD
OBJS=""
pids=""
for s in $SRCS; do
  o=$BD/${s%.hip}.o
  OBJS="$OBJS $o"
  stale=0
  for dep in "$s" common.h ../../include/dicow_hip.h $(ls *.inc experiments/*.inc 2>/dev/null); do
    if [ ! -f "$o" ] || [ "$dep" -nt "$o" ]; then stale=1; fi
  done
  if [ $stale -eq 1 ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$s" -o "$o" "$@" &
    pids="$pids $!"
  fi
done
fail=0
for p in $pids; do wait $p || fail=1; done
if [ $fail -ne 0 ]; then echo "build.sh: a compilation failed" >&2; exit 1; fi
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
echo "built $(realpath $OUT)"
