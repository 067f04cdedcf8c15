#!/bin/bash
# round 4, call 13: what do the epilogue's stores cost the next tile's counted waits?  (tile timeline with the stores dropped)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04k; mkdir -p $O
DICOW_HIP_LIB=tools/libv_prof.so timeout 300 python tools/profile_ntr.py > $O/timeline_stores.txt 2>&1
DICOW_HIP_LIB=tools/libv_profns.so timeout 300 python tools/profile_ntr.py > $O/timeline_nostores.txt 2>&1
DICOW_HIP_LIB=tools/libv_prof.so timeout 300 python tools/profile_ntr.py > $O/timeline_stores2.txt 2>&1
DICOW_HIP_LIB=tools/libv_profns.so timeout 300 python tools/profile_ntr.py > $O/timeline_nostores2.txt 2>&1
for f in stores nostores stores2 nostores2; do echo "== $f"; grep -v amdgpu.ids $O/timeline_$f.txt | grep -v "shader clocks"; done
