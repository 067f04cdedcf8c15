#!/bin/bash
# Round-3 FINAL evidence, everything on one box in one gpurun call: full GPU test run, counter passes, bench lines of every
# workload, kernel summaries.  Outputs under gpurun_out/r03/ (copy what is to be judged into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
(timeout 1700 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt)
bash tools/prof_pmc.sh > /dev/null 2>&1; cp gpurun_out/pmc_summary.json $O/pmc_hbm_traffic.json; cp gpurun_out/pmc_summary.json profiles/r03_pmc_hbm_traffic.json
bash tools/r03_evidence.sh > $O/evidence.log 2>&1; tail -16 $O/evidence.log
bash tools/prof_pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma_summary.json $O/pmc_mfma_lds.json 2>/dev/null
bash tools/prof_pmc_l2.sh > /dev/null 2>&1; cp gpurun_out/pmc_l2_summary.json $O/pmc_l2_hit_rate.json 2>/dev/null
bash tools/prof_trace_base.sh > /dev/null 2>&1; cp gpurun_out/base_trace_summary.txt $O/base_trace_summary.txt
ls -la $O | tail -40
