cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_realdims.py tests/test_gpu_dp.py tests/test_gpu_graph.py -q -k "gemm_tn or rd_base or rd_tiny or two_ranks or graph" 2>&1 | tail -3
for i in 1 2; do python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base graph', d['value'], d['ms_per_step'], d['kernels'].get('gemm_tn_kernel'))"; done
bash tools/prof_step.sh --model whisper-base --batch 8 > /dev/null 2>&1; head -8 gpurun_out/kernel_stats.csv | cut -c1-120; mkdir -p gpurun_out/r03; cp gpurun_out/kernel_stats.csv gpurun_out/r03/base_b8_kernel_stats.csv
python bench.py --model whisper-base --batch 8 --no-cpu-baseline > gpurun_out/r03/bench_base_b8.json 2>/dev/null; python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > gpurun_out/r03/bench_base_b8_graph.json 2>/dev/null
bash tools/dp_single_rank.sh > /dev/null 2>&1; cp gpurun_out/dp_single_rank.txt gpurun_out/r03/dp_single_rank.txt; cat gpurun_out/dp_single_rank.txt
