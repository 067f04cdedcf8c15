#!/bin/bash
# Round-4 experiments, one gpurun call each:   gpurun -- 'bash tools/r04_experiments.sh <name>'
# (the evidence of the SHIPPED tree is tools/r04_final.sh; every table under profiles/r04_*.txt names the experiment it came from).
# Variant libraries (tools/libv_*.so, git-ignored, built here on the CPU box before the call):
#   lnfold        -                                   the LayerNorm fold is a run-time switch (DICOW_LN_FOLD=1)
#   attn_variants tools/build_attn_variants.sh        libv_acts.so (-DATTN_DKV_CTS=1), libv_acts1b.so (+ -DATTN_DKV_1BAR=1)
#   tn_w4         tools/build_variants.sh tn8 "-DTN_W4=0" against a -DTN_W4=1 shipped build of that day
#   nt128w        tools/build_var.sh nt128w0 "-DNT128W=0" "" "" ""    against a -DNT128W=1 build
#   ntd_gelu / ntd_light   tools/build_ntd.sh         libv_ntd.so (csrc/experiments/gemm_ntd.hip, DICOW_NT_DEFER)
#   store_cost    tools/build_variants.sh prof "-DNTR_PROFILE" profns "-DNTR_PROFILE -DNTR_NO_STORES"
#   logmel        tools/libv_lmdirect.so              a -DLOGMEL_DIRECT build (the direct DFT of rounds 1-3)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04x_$1; mkdir -p $O
case "$1" in
lnfold)         # profiles/r04_lnfold.txt
  (timeout 900 python -m pytest tests/test_gpu_lnfold.py -q -x > $O/lnfold_tests.txt 2>&1; tail -4 $O/lnfold_tests.txt)
  python tools/ab_lnfold.py > $O/ab_lnfold.txt 2>&1; tail -8 $O/ab_lnfold.txt
  DICOW_LN_FOLD=1 bash tools/prof_encfwd.sh > $O/prof_fold.txt 2>&1; cp gpurun_out/encfwd_kernel_stats.csv $O/encfwd_fold.csv
  DICOW_LN_FOLD=0 bash tools/prof_encfwd.sh > $O/prof_nofold.txt 2>&1; cp gpurun_out/encfwd_kernel_stats.csv $O/encfwd_nofold.csv ;;
hipblaslt)      # profiles/r04_gemm_vs_hipblaslt.{json,txt}
  python tools/gemm_vs_hipblaslt.py $O/gemm_vs_hipblaslt.json > $O/gemm_vs_hipblaslt.txt 2>&1; tail -30 $O/gemm_vs_hipblaslt.txt ;;
attn_variants)  # profiles/r04_attn_bwd_variants.txt
  ATTN_LOG2=1 REPS=3 python tools/ab_attn.py shipped=ts-asr-whisper_amd/libdicow_hip.so cts=tools/libv_acts.so cts1b=tools/libv_acts1b.so > $O/ab_attn.txt 2>&1; cat $O/ab_attn.txt
  (DICOW_HIP_LIB=tools/libv_acts1b.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -k "attn or attention" > $O/attn_tests_cts1b.txt 2>&1; tail -4 $O/attn_tests_cts1b.txt) ;;
tn_w4)          # profiles/r04_tn_w4.txt
  (timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -q -k "tn" > $O/tn_tests.txt 2>&1; tail -8 $O/tn_tests.txt)
  REPS=3 python tools/ab_step.py w4=ts-asr-whisper_amd/libdicow_hip.so w8=tools/libv_tn8.so > $O/ab_step_tn.txt 2>&1; tail -6 $O/ab_step_tn.txt ;;
nt128w)         # profiles/r04_nt128w.txt
  timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_nt" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
  echo "== shipped" > $O/base_shapes.txt; timeout 300 python tools/bench_base_shapes.py >> $O/base_shapes.txt 2>&1
  echo "== NT128W=0" >> $O/base_shapes.txt; DICOW_HIP_LIB=tools/libv_nt128w0.so timeout 300 python tools/bench_base_shapes.py >> $O/base_shapes.txt 2>&1
  for r in 1 2; do
    python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > $O/base_new_$r.json 2>/dev/null
    DICOW_HIP_LIB=tools/libv_nt128w0.so python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > $O/base_old_$r.json 2>/dev/null
  done
  for f in $O/base_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"; done
  cat $O/base_shapes.txt ;;
ntd_gelu)       # profiles/r04_ntd_deferred_epilogue.txt (first half), r04_ntd_ab_raw.txt
  timeout 600 python tools/ab_ntd.py tools/libv_ntd.so 3 > $O/ab_ntd.txt 2>&1; cat $O/ab_ntd.txt ;;
ntd_light)      # profiles/r04_ntd_deferred_epilogue.txt (second half), r04_ntd_light_ab_raw.txt
  timeout 900 python tools/ab_ntd.py tools/libv_ntd.so 12 > $O/ab_ntd_light.txt 2>&1; cat $O/ab_ntd_light.txt ;;
store_cost)     # profiles/r04_ntr_tile_timeline_{stores,nostores}.txt
  for rep in 1 2; do
    DICOW_HIP_LIB=tools/libv_prof.so timeout 300 python tools/profile_ntr.py > $O/timeline_stores$rep.txt 2>&1
    DICOW_HIP_LIB=tools/libv_profns.so timeout 300 python tools/profile_ntr.py > $O/timeline_nostores$rep.txt 2>&1
  done
  for f in $O/timeline_*.txt; do echo "== $f"; grep -v amdgpu.ids $f | grep -v "shader clocks"; done ;;
logmel)         # profiles/r04_bench_logmel.txt
  timeout 600 python -m pytest tests/ -x -q -m gpu -k "logmel or from_audio or augment or features" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
  timeout 200 python tools/bench_logmel.py tools/libv_lmdirect.so > $O/bench_logmel.txt 2>&1; cat $O/bench_logmel.txt ;;
*) echo "unknown experiment '$1'"; exit 2 ;;
esac
