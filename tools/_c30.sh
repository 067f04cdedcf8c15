cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c30_tests.txt 2>&1; tail -3 gpurun_out/c30_tests.txt)
O=gpurun_out/r03; mkdir -p $O
bash tools/prof_pmc.sh > /dev/null 2>&1; cp gpurun_out/pmc_summary.json $O/pmc_hbm_traffic.json; cp gpurun_out/pmc_summary.json profiles/r03_pmc_hbm_traffic.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/prof_step.sh > /dev/null 2>&1; cp gpurun_out/kernel_stats.csv $O/kernel_stats.csv
bash tools/prof_encfwd.sh > /dev/null 2>&1; cp gpurun_out/encfwd_kernel_stats.csv $O/encfwd_kernel_stats.csv
python bench.py --from-audio --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_from_audio.json 2>/dev/null
for v in ctc preheat se; do python bench.py --$v --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_$v.json 2>/dev/null; done
python bench.py --model whisper-base --batch 8 --no-cpu-baseline > $O/bench_base_b8.json 2>/dev/null
python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > $O/bench_base_b8_graph.json 2>/dev/null
for f in $O/bench_*.json; do python -c "
import json; d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], d['roofline']['traffic'], (d.get('kernels') or {}).get('gemm_tn_kernel',{}).get('tflops'), (d.get('encoder_forward') or {}).get('ms'), (d.get('power') or {}).get('sclk_mhz_mean'))"; done
