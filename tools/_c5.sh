cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_optimizer.py -q -k "gemm_tn or optimizer or schedule or adamw or train_step" > gpurun_out/c5_tests.txt 2>&1; tail -12 gpurun_out/c5_tests.txt
timeout 1500 python -m pytest tests/test_gpu_realdims.py tests/test_gpu_model.py tests/test_gpu_dp.py tests/test_gpu_graph.py -q -x > gpurun_out/c5_tests2.txt 2>&1; tail -8 gpurun_out/c5_tests2.txt
DICOW_BENCH_BREAKDOWN=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c5_bench.json 2> gpurun_out/c5_bench.err; grep gemm_tn gpurun_out/c5_bench.err | head; python -c "
import json; d=json.loads(open('gpurun_out/c5_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels'], d['encoder_forward']['ms'])"
bash tools/prof_step.sh > gpurun_out/c5_prof.txt 2>&1; head -24 gpurun_out/c5_prof.txt | cut -c1-150
