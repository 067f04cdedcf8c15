cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 0 22; do echo "bench NT_VARIANT=$v: $(DICOW_HIP_LIB=tools/libv_abl.so DICOW_NT_VARIANT=$v DICOW_BENCH_BREAKDOWN=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-power 2>gpurun_out/c19_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['encoder_forward']['ms'])")"; done; done
for v in 0 22; do echo "== $v"; grep "gemm_nt M24000" gpurun_out/c19_$v.err | cut -c1-150; done
