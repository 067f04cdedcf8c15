#!/bin/bash
# Experiment build: the library with the deferred-epilogue fc1 kernel (csrc/experiments/gemm_ntd.hip) -> tools/libv_ntd.so
# (run-time switch DICOW_NT_DEFER=1|2|3; tools/ab_ntd.py tools/libv_ntd.so compares it with the ring kernel bit for bit and in time)
set -e
cd "$(dirname "$0")/../ts-asr-whisper_amd/csrc"
bash build.sh > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c gemm.hip -o build/gemm_v_ntd.o -DNT_DEFER=0 -DNT_DEFER_BUILD=1 &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c experiments/gemm_ntd.hip -o build/gemm_ntd_x.o $NTD_FLAGS &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c experiments/gemm_ntl.hip -o build/gemm_ntl_x.o $NTD_FLAGS &
wait
objs="build/gemm_ntd_x.o build/gemm_ntl_x.o"
for s in $(ls *.hip); do b=${s%.hip}; if [ "$b" = gemm ]; then objs="$objs build/gemm_v_ntd.o"; else objs="$objs build/$b.o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../tools/libv_${NTD_NAME:-ntd}.so
rm -f build/gemm_ntd_x.o build/gemm_ntl_x.o
echo "built tools/libv_${NTD_NAME:-ntd}.so"
