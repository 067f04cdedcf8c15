cd $GRAFT_REPO_ROOT
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -i -E "DRAM|MALL|EA0_RD|EA_RD|HBM|TCC_EA" | head -40) > gpurun_out/c10_counters.txt 2>&1
bash tools/dp_single_rank.sh > gpurun_out/c10_dp.txt 2>&1
bash tools/prof_pmc.sh > gpurun_out/c10_pmc.txt 2>&1
