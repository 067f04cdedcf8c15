#!/bin/bash
# round 4, call 12: deferred LIGHT epilogues (bf16 stores / fp32 residual) -- bit identity against the ring kernel and timing
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04j; mkdir -p $O
timeout 900 python tools/ab_ntd.py tools/libv_ntd.so 12 > $O/ab_ntd_light.txt 2>&1; cat $O/ab_ntd_light.txt
