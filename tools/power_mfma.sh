#!/bin/bash
# Sustained MFMA-only rate, board power and shader clock (constant and random operands): the attainable matrix-pipe ceiling
# under the board's power cap.   bash tools/power_mfma.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for mode in "" random; do
  tools/probe_clock 5 $mode > /tmp/pm_$mode.log &
  pid=$!
  sleep 2.5
  for i in 1 2 3; do rocm-smi --showpower --showclocks | grep -E "sclk|Power" | tr -s ' ' | cut -c1-90; sleep 0.5; done
  wait $pid
  tail -2 /tmp/pm_$mode.log
done
