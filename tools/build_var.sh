#!/bin/bash
# Build A/B variants of the library that differ in the -D flags of gemm.hip, fddt_ln.hip, attention.hip and elementwise.hip:
#   tools/build_var.sh name "gemm flags" "fddt_ln flags" "attention flags" "elementwise flags" [name ...]     ("" = the shipped object)
# -> tools/libv_<name>.so (git-ignored; travels to the GPU box).  Used with DICOW_HIP_LIB=...   The variant OBJECTS go to gpurun_out/var_build/
# (never pushed to the GPU box, never in csrc/build/).
set -e
cd "$(dirname "$0")/../ts-asr-whisper_amd/csrc"
bash build.sh > /dev/null
VB=../../gpurun_out/var_build; mkdir -p $VB
pids=""
names=()
while [ $# -gt 4 ]; do
  n=$1; g=$2; f=$3; t=$4; e=$5; shift 5
  names+=("$n")
  rm -f $VB/gemm_v_$n.o $VB/fddt_ln_v_$n.o $VB/attention_v_$n.o $VB/elementwise_v_$n.o
  if [ -n "$g" ]; then ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c gemm.hip -o $VB/gemm_v_$n.o $g ) & pids="$pids $!"; fi
  if [ -n "$f" ]; then ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c fddt_ln.hip -o $VB/fddt_ln_v_$n.o $f ) & pids="$pids $!"; fi
  if [ -n "$e" ]; then ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c elementwise.hip -o $VB/elementwise_v_$n.o $e ) & pids="$pids $!"; fi
  if [ -n "$t" ]; then ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c attention.hip -o $VB/attention_v_$n.o $t ) & pids="$pids $!"; fi
done
for p in $pids; do wait $p; done
for n in "${names[@]}"; do
  objs=""
  for s in $(ls *.hip); do
    b=${s%.hip}
    if [ -f $VB/${b}_v_$n.o ]; then objs="$objs $VB/${b}_v_$n.o"; else objs="$objs build/$b.o"; fi
  done
  hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../tools/libv_$n.so
  echo "built tools/libv_$n.so"
done
