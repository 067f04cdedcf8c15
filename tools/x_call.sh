# scratch call: the DP rehearsal (one-rank RCCL reducer forced, emulated 8-rank fabric load) with the two half-batch streams (default) and on one stream
out=gpurun_out/r06x13; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_rccl.py tests/test_gpu_dp.py -x -q 2>&1 | tail -n 6 | tee $out/tests.txt
python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -n 1 > $out/bench_one_stream_ref.json
python -c "
import json; d=json.loads(open('$out/bench_one_stream_ref.json').read()); print(d['ms_per_step'], d['one_stream_reference'])" | tee -a $out/tests.txt
echo "== two half-batch streams (default)" | tee $out/dp_emulated.txt
DP_EMUL_REPS=2 timeout 1200 python tools/dp_emulate.py 100 2>&1 | grep -v amdgpu.ids | tee -a $out/dp_emulated.txt
echo "== one stream (DICOW_SPLIT_FWD=0)" | tee -a $out/dp_emulated.txt
DICOW_SPLIT_FWD=0 DP_EMUL_REPS=1 timeout 1200 python tools/dp_emulate.py 100 2>&1 | grep -v amdgpu.ids | tee -a $out/dp_emulated.txt
