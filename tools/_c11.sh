cd $GRAFT_REPO_ROOT
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 DICOW_FORCE_REDUCE=1 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-power > gpurun_out/c11_forced.json 2> gpurun_out/c11_forced.err; tail -c 300 gpurun_out/c11_forced.json; tail -15 gpurun_out/c11_forced.err
timeout 2000 python -m pytest tests -m gpu -q -x > gpurun_out/c11_tests.txt 2>&1; tail -6 gpurun_out/c11_tests.txt
python bench.py --model whisper-base --batch 8 --no-cpu-baseline > gpurun_out/c11_base.json 2> gpurun_out/c11_base.err; python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > gpurun_out/c11_base_graph.json 2>> gpurun_out/c11_base.err
python -c "
import json
for f in ('gpurun_out/c11_base.json','gpurun_out/c11_base_graph.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels'].get('gemm_tn_kernel',{}).get('tflops'))"
