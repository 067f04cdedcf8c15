#!/bin/bash
# round 4, call 10: the 128 x 256 mid-size NT kernel (tests, per-shape table, whisper-base step A/B) + the log-mel LDS trim
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_nt or logmel" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
echo "== shipped" > $O/base_shapes.txt; timeout 300 python tools/bench_base_shapes.py >> $O/base_shapes.txt 2>&1
echo "== NT128W=0" >> $O/base_shapes.txt; DICOW_HIP_LIB=tools/libv_nt128w0.so timeout 300 python tools/bench_base_shapes.py >> $O/base_shapes.txt 2>&1
for r in 1 2; do
  python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > $O/base_new_$r.json 2>/dev/null
  DICOW_HIP_LIB=tools/libv_nt128w0.so python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > $O/base_old_$r.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04h/base_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median'))
PY
timeout 200 python tools/bench_logmel.py tools/libv_lmdirect.so > $O/bench_logmel.txt 2>&1; cat $O/bench_logmel.txt
cat $O/base_shapes.txt
