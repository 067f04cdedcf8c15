#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04e; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt)
python tools/bench_logmel.py tools/libv_lmdirect.so > $O/bench_logmel.txt 2>&1; tail -4 $O/bench_logmel.txt
python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04e/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["encoder_forward"], d["encoder_forward_train"], d["front_end"])
PY
