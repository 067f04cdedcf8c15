#!/bin/bash
# Round-5 evidence, everything on one box in one gpurun call: full GPU test run, counter passes, bench lines of every workload,
# kernel summaries, the one-rank RCCL table, the preflight line.  Outputs under gpurun_out/r05/ (what is judged is copied into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt)
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/prof_pmc.sh > /dev/null 2>&1; cp gpurun_out/pmc_summary.json $O/pmc_hbm_traffic.json
bash tools/prof_step.sh > $O/prof_step.txt 2>&1; cp gpurun_out/kernel_stats.csv $O/kernel_stats.csv
bash tools/prof_encfwd.sh > $O/prof_encfwd.txt 2>&1; cp gpurun_out/encfwd_kernel_stats.csv $O/encfwd_kernel_stats.csv
python bench.py --preflight > $O/preflight_1gpu.json 2> $O/preflight.err
python bench.py --from-audio --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_from_audio.json 2>/dev/null
for v in se ctc preheat; do
  python bench.py --$v --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
done
python bench.py --model whisper-base --batch 8 --no-cpu-baseline > $O/bench_base_b8.json 2> $O/base.err
python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > $O/bench_base_b8_graph.json 2>> $O/base.err
bash tools/dp_single_rank.sh > $O/dp_single_rank.log 2>&1; cp gpurun_out/dp_single_rank.txt $O/dp_single_rank.txt
bash tools/prof_pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma_summary.json $O/pmc_mfma_lds.json 2>/dev/null
bash tools/prof_pmc_l2.sh > /dev/null 2>&1; cp gpurun_out/pmc_l2_summary.json $O/pmc_l2_hit_rate.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default_again.json 2>/dev/null
for f in $O/bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], (d.get('kernels') or {}).get('gemm_tn_kernel',{}).get('tflops'), (d.get('encoder_forward') or {}).get('ms'), (d.get('encoder_forward_train') or {}).get('ms'), d.get('power'))"; done
cat $O/dp_single_rank.txt | tail -12; head -c 600 $O/preflight_1gpu.json
