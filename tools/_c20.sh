cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c20_tests.txt 2>&1; tail -3 gpurun_out/c20_tests.txt)
bash tools/r03_evidence.sh > gpurun_out/c20_evidence.txt 2>&1; tail -14 gpurun_out/c20_evidence.txt | cut -c1-200
bash tools/prof_pmc.sh > gpurun_out/c20_pmc.txt 2>&1; cp gpurun_out/pmc_summary.json gpurun_out/r03/pmc_hbm_traffic.json
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('traffic check:', d['roofline']['traffic'], d['roofline']['traffic_source'][:160])"
