cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c25_tests.txt 2>&1; tail -3 gpurun_out/c25_tests.txt)
bash tools/prof_pmc.sh > gpurun_out/c25_pmc.txt 2>&1; mkdir -p gpurun_out/r03; cp gpurun_out/pmc_summary.json gpurun_out/r03/pmc_hbm_traffic.json; cp gpurun_out/pmc_summary.json profiles/r03_pmc_hbm_traffic.json
bash tools/r03_evidence.sh > gpurun_out/c25_evidence.txt 2>&1; tail -12 gpurun_out/c25_evidence.txt | cut -c1-200
bash tools/prof_pmc_mfma.sh > gpurun_out/c25_mfma.txt 2>&1; cp gpurun_out/pmc_mfma_summary.json gpurun_out/r03/pmc_mfma_lds.json; tail -13 gpurun_out/c25_mfma.txt | cut -c1-220
bash tools/prof_pmc_l2.sh > gpurun_out/c25_l2.txt 2>&1; cp gpurun_out/pmc_l2_summary.json gpurun_out/r03/pmc_l2_hit_rate.json; tail -14 gpurun_out/c25_l2.txt
