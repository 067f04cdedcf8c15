#!/bin/bash
# Build libdicow_hip.so for gfx950 in-tree (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../libdicow_hip.so
SRCS=$(ls *.hip)
mkdir -p build
OBJS=""
pids=""
for s in $SRCS; do
  o=build/${s%.hip}.o
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
