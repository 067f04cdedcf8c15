#!/bin/bash
# The DP side effects that can be timed on ONE GPU: the headline step (a) plain, (b) as rank 0 of a one-rank
# RCCL group with the bucketed side-stream reducer forced (DICOW_FORCE_REDUCE=1: one all-reduce of the flat gradient store per
# backward segment on the comm stream), (c) with the persistent GEMM grids limited to 240 CUs (what GradReducer sets for N > 1:
# 16 CUs left to the RCCL channels), (d) both.  -> gpurun_out/dp_single_rank.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
out=gpurun_out/dp_single_rank.txt; : > $out
cat > /tmp/_dp_line.py <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); a = d["allreduce"]
print("%-40s %8.3f ms/step (median %8.3f) %7.2f utt/s  backend=%s bytes/step=%d buckets=%d exposed_ms=%s gemm_cus=%s loss=%.6f" % (
    sys.argv[2], d["ms_per_step"], d["ms_per_step_median"], d["value"], a["backend"], a["bytes_per_step"], a["buckets_per_step"],
    a["exposed_ms_per_step"], a["gemm_cus"], d["loss"]))
PY
line() { python /tmp/_dp_line.py "$1" "$2" >> $out 2>&1; }
for rep in 1 2; do
python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-power > gpurun_out/_dp_a.json 2>/dev/null; line gpurun_out/_dp_a.json "plain (no process group)"
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 DICOW_FORCE_REDUCE=1 python bench.py --gpus 1 --steps 12 --warmup 4 --no-cpu-baseline --no-power > gpurun_out/_dp_b.json 2>/dev/null; line gpurun_out/_dp_b.json "one-rank RCCL group, reducer forced"
python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-power --gemm-cus 240 > gpurun_out/_dp_c.json 2>/dev/null; line gpurun_out/_dp_c.json "plain, GEMM grids on 240 CUs"
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29518 DICOW_FORCE_REDUCE=1 python bench.py --gpus 1 --steps 12 --warmup 4 --no-cpu-baseline --no-power --gemm-cus 240 > gpurun_out/_dp_d.json 2>/dev/null; line gpurun_out/_dp_d.json "RCCL reducer forced + 240 CUs"
done
rm -f gpurun_out/_dp_?.json
cat $out
