cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_optimizer.py tests/test_gpu_dp.py tests/test_gpu_graph.py tests/test_gpu_rccl.py tests/test_gpu_ddp_dropin.py tests/test_gpu_fullsize.py -q -k "first_writer or two_ranks or graph or rccl or ddp or train_step or update_rule" 2>&1 | tail -4
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('turbo', d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels']['gemm_tn_kernel']['tflops'], d['roofline']['traffic'])"; done
