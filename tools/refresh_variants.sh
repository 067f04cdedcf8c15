#!/bin/bash
# bench lines + rocprofv3 kernel summaries of the other workloads (SE-DiCoW, CTC recipe, preheat phase, whisper-base B=8 eager / graph)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in se ctc preheat; do
  python bench.py --$v --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r02w_bench_$v.json 2> gpurun_out/r02w_bench_$v.err
  bash tools/prof_step.sh --$v > /dev/null 2>&1; cp gpurun_out/kernel_stats.csv gpurun_out/r02w_kernel_stats_$v.csv
done
python bench.py --model whisper-base --batch 8 --no-cpu-baseline > gpurun_out/r02w_bench_base_b8.json 2> gpurun_out/r02w_base.err
python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > gpurun_out/r02w_bench_base_b8_graph.json 2> gpurun_out/r02w_baseg.err
bash tools/prof_step.sh --model whisper-base --batch 8 > /dev/null 2>&1; cp gpurun_out/kernel_stats.csv gpurun_out/r02w_base_b8_kernel_stats.csv
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02w_bench_default_samebox.json 2>/dev/null
for f in gpurun_out/r02w_bench_*.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step_median'], d.get('power'))"; done
