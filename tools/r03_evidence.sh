#!/bin/bash
# Round-3 evidence on ONE box: bench lines of every workload, rocprofv3 kernel summaries, DP single-rank timings, the
# front-end bench.  Outputs under gpurun_out/r03/ (copy what is to be judged into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/prof_step.sh > $O/prof_step.txt 2>&1; cp gpurun_out/kernel_stats.csv $O/kernel_stats.csv
bash tools/prof_encfwd.sh > $O/prof_encfwd.txt 2>&1; cp gpurun_out/encfwd_kernel_stats.csv $O/encfwd_kernel_stats.csv
python bench.py --from-audio --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_from_audio.json 2>/dev/null
for v in se ctc preheat; do
  python bench.py --$v --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  bash tools/prof_step.sh --$v > /dev/null 2>&1; cp gpurun_out/kernel_stats.csv $O/kernel_stats_$v.csv
done
python bench.py --model whisper-base --batch 8 --no-cpu-baseline > $O/bench_base_b8.json 2> $O/base.err
python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > $O/bench_base_b8_graph.json 2>> $O/base.err
bash tools/prof_step.sh --model whisper-base --batch 8 > /dev/null 2>&1; cp gpurun_out/kernel_stats.csv $O/base_b8_kernel_stats.csv
bash tools/dp_single_rank.sh > $O/dp_single_rank.log 2>&1; cp gpurun_out/dp_single_rank.txt $O/dp_single_rank.txt
python tools/bench_kernels.py > $O/bench_kernels.txt 2>&1
python tools/bench_rows.py > $O/bench_rows.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default_again.json 2>/dev/null
for f in $O/bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], (d.get('kernels') or {}).get('gemm_tn_kernel',{}).get('tflops'), (d.get('encoder_forward') or {}).get('ms'), d.get('power'))"; done
cat $O/dp_single_rank.txt
