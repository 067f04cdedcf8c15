cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_graph.py tests/test_gpu_optimizer.py tests/test_gpu_bench_contract.py -q > gpurun_out/c4_tests.txt 2>&1; tail -15 gpurun_out/c4_tests.txt
timeout 900 python -m pytest tests/test_gpu_realdims.py -q -k "batch16 or rd_turbo" -s > gpurun_out/c4_realdims.txt 2>&1; tail -15 gpurun_out/c4_realdims.txt
python tools/bench_kernels.py 2>&1 | grep -E "logmel|augment" > gpurun_out/c4_frontend.txt; cat gpurun_out/c4_frontend.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err; tail -c 1500 gpurun_out/c4_bench.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --from-audio > gpurun_out/c4_bench_audio.json 2> gpurun_out/c4_bench_audio.err; tail -c 600 gpurun_out/c4_bench_audio.json
