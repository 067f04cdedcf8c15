cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c27_tests.txt 2>&1; tail -3 gpurun_out/c27_tests.txt)
mkdir -p gpurun_out/r03
python bench.py --se --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r03/bench_se.json 2>/dev/null
bash tools/prof_step.sh --se > /dev/null 2>&1; cp gpurun_out/kernel_stats.csv gpurun_out/r03/kernel_stats_se.csv
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03/bench_default_samebox.json 2>/dev/null
for f in bench_se bench_default_samebox; do python -c "
import json; d=json.loads([l for l in open('gpurun_out/r03/$f.json').read().splitlines() if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], (d.get('power') or {}).get('sclk_mhz_mean'))"; done
