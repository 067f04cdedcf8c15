cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in 0 21 22; do echo "NT_VARIANT=$v: $(DICOW_HIP_LIB=tools/libv_abl.so DICOW_NT_VARIANT=$v python tools/enc_fwd.py 20 2>/dev/null | tail -1)"; done
done
for v in 0 21; do echo "bench NT_VARIANT=$v: $(DICOW_HIP_LIB=tools/libv_abl.so DICOW_NT_VARIANT=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"; done
