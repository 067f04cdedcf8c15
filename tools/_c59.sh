cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=3 timeout 900 python tools/ab_step.py m128=$T/libv_m128.so m192=$T/libv_m192.so m256=$T/libv_m256.so m400=$T/libv_m400.so 2>&1 | grep -v amdgpu.ids | tail -5
for v in m128 m192 m256 m400; do
DICOW_HIP_LIB=$T/libv_$v.so python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done
