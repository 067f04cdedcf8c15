#!/bin/bash
# round 4, call 11: deferred-epilogue fc1 kernel -- bit identity against the ring kernel and timing
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04i; mkdir -p $O
timeout 600 python tools/ab_ntd.py > $O/ab_ntd.txt 2>&1; cat $O/ab_ntd.txt
