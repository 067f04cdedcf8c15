#!/bin/bash
# Build A/B variants of the library that differ only in attention.hip's -D flags (with the measured-and-rejected kernels of csrc/experiments/ compiled in): tools/build_attn_variants.sh name "flags" [name "flags" ...]
# -> tools/libv_<name>.so (git-ignored; travels to the GPU box).  Used with DICOW_HIP_LIB=... and tools/ab_*.py.
set -e
cd "$(dirname "$0")/../ts-asr-whisper_amd/csrc"
bash build.sh > /dev/null
pids=""
names=()
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  names+=("$n")
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c attention.hip -DDICOW_EXPERIMENTS -o build/attention_v_$n.o $f ) &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
for n in "${names[@]}"; do
  objs=$(ls build/*.o | grep -v "_v_" | grep -v "build/attention.o")
  hipcc --offload-arch=gfx950 -shared -fPIC $objs build/attention_v_$n.o -o ../../tools/libva_$n.so
  echo "built tools/libva_$n.so"
done
