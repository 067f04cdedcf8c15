out=gpurun_out/r06x7; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python tools/probe_opt_overlap.py 8 3 2>&1 | grep -v amdgpu.ids | tee $out/probe_opt_overlap.txt
