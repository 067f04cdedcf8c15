out=gpurun_out/r06x11; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_forward.py -x -q 2>&1 | tail -25 | tee $out/split_tests.txt
timeout 900 python -m pytest tests/test_gpu_realdims.py -x -q -k "configs4" 2>&1 | tail -5 | tee -a $out/split_tests.txt
pr() { python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], d['ms_per_step_median'], 'roof', d['roofline']['frac'], 'train-fwd', (d.get('encoder_forward_train') or {}).get('ms'), (d.get('power') or {}).get('sclk_mhz_mean'))"; }
for rep in 1 2; do for v in se_on se_bwd_off se_off; do e="DICOW_SPLIT_BWD=1"; [ $v = se_bwd_off ] && e="DICOW_SPLIT_BWD=0"; [ $v = se_off ] && e="DICOW_SPLIT_FWD=0"
  env $e timeout 600 python bench.py --se --steps 8 --warmup 3 --no-extra --no-cpu-baseline 2>$out/bench_$v.err | tail -1 > $out/bench_${v}_$rep.json; pr $out/bench_${v}_$rep.json $v | tee -a $out/ab.txt
done; done
for v in preheat ctc; do for e in 1 0; do DICOW_SPLIT_FWD=$e timeout 600 python bench.py --$v --steps 8 --warmup 3 --no-extra --no-cpu-baseline 2>$out/bench_$v.err | tail -1 > $out/bench_${v}_$e.json; pr $out/bench_${v}_$e.json ${v}_split$e | tee -a $out/ab.txt; done; done
