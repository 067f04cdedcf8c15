#!/bin/bash
# One gpurun call of round 6: tools/r06_call.sh <tag> <step> [<step> ...]   (outputs under gpurun_out/r06<tag>/)
#   steps: attn_tests | attn_bench | gpu_tests | bench | bench_extra | prof | pmc | encfwd | hf | ...  (see the case below)
tag=$1; shift
out=gpurun_out/r06$tag; mkdir -p $out
export TMPDIR=/tmp
for step in "$@"; do
  echo "=== $step" | tee -a $out/log.txt
  case $step in
    attn_tests) timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn" 2>&1 | tail -15 | tee $out/attn_tests.txt
                timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "attention_at_bench_shape" -s 2>&1 | tail -15 | tee -a $out/attn_tests.txt ;;
    attn_bench) ATTN_LOG2=1 timeout 600 python tools/bench_attn.py 2>&1 | tee $out/attn_bench.txt ;;
    attn_prof)  ATTN_LOG2=1 ATTN_BWD_REPS=1 ATTN_PROFILE_FUSED=1 DICOW_HIP_LIB=$PWD/tools/libv_fprof.so timeout 600 python tools/bench_attn.py 2>&1 | tee $out/attn_prof.txt ;;
    attn_rot)   for r in 0 1 2 99; do echo "rstride $r" | tee -a $out/attn_rot.txt; DICOW_ATTN_FUSED_RSTRIDE=$r ATTN_LOG2=1 ATTN_BWD_REPS=2 timeout 600 python tools/bench_attn.py 2>&1 | grep attn_bwd | tee -a $out/attn_rot.txt; done ;;
    attn_rocprof) (cd /tmp && ATTN_LOG2=1 ATTN_BWD_REPS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_attn_$tag -o attn -- python $GRAFT_REPO_ROOT/tools/bench_attn.py > $GRAFT_REPO_ROOT/$out/attn_rocprof_run.txt 2>&1)
                find /tmp/prof_attn_$tag -name "*kernel_stats.csv" -exec cp {} $out/attn_kernel_stats.csv \; ; head -12 $out/attn_kernel_stats.csv ;;
    attn_abl)   for rep in 1 2; do for v in base abl1 abl3 abl67 abl63; do l=$PWD/ts-asr-whisper_amd/libdicow_hip.so; [ $v != base ] && l=$PWD/tools/libv_f$v.so
                  echo -n "$v: " | tee -a $out/attn_abl.txt; DICOW_HIP_LIB=$l ATTN_LOG2=1 ATTN_BWD_REPS=1 timeout 300 python tools/bench_attn.py 2>&1 | grep "fused" | tee -a $out/attn_abl.txt; done; done ;;
    attn_pmc)   for r in 1 2 99; do for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
                  (cd /tmp && DICOW_ATTN_FUSED_RSTRIDE=$r ATTN_LOG2=1 ATTN_BWD_REPS=1 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python $GRAFT_REPO_ROOT/tools/bench_attn.py > /tmp/pmc_run.txt 2>&1)
                  f=$(ls /tmp/pmc_$tag/*/*counter_collection.csv 2>/dev/null | head -1)
                  python - "$f" "$r" <<'PY' | tee -a $out/attn_pmc.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for row in csv.DictReader(open(sys.argv[1])):
    name = (row.get("Kernel_Name") or "").split("(")[0].replace("void ", "")
    if not name.startswith("attn_"): continue
    k = (name, row["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"] or 0)
for (name, c), (n, v) in sorted(agg.items()): print(f"rstride {sys.argv[2]}  {name:40s} {c:14s} per launch {v / n / 1e6:12.2f} M   ({n} launches)")
PY
                  rm -rf /tmp/pmc_$tag; done; done ;;
    hf)         timeout 900 python -m pytest tests/test_gpu_hf_trainer.py tests/test_gpu_bench_contract.py -x -q -s 2>&1 | tail -30 | tee $out/hf_tests.txt ;;
    attn_sq)    ATTN_LOG2=1 ATTN_BWD_REPS=1 bash tools/prof_attn_pmc.sh 2>&1 | tail -120 | tee $out/attn_sq.txt; cp gpurun_out/attn_pmc_summary.json $out/ 2>/dev/null ;;
    dp_emul)    timeout 1500 python tools/dp_emulate.py 100 150 200 2>&1 | tee $out/dp_emulated.txt ;;
    streams)    timeout 900 python tools/probe_streams.py 2>&1 | grep -v amdgpu.ids | tee $out/probe_streams.txt ;;
    dp_prof)    for v in base emul; do ex=""; [ $v = emul ] && ex="--emulate-fabric-gbps 100"
                  (cd /tmp && RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 DICOW_FORCE_REDUCE=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dpp_$v -o dp -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --gemm-cus 240 --no-extra --no-cpu-baseline --no-power --steps 8 --warmup 3 --profile-steps 1 $ex > $GRAFT_REPO_ROOT/$out/dp_prof_$v.json 2>/tmp/dpp_$v.err)
                  find /tmp/dpp_$v -name "*kernel_stats.csv" -exec cp {} $out/dp_kernel_stats_$v.csv \;
                  [ $v = emul ] && find /tmp/dpp_$v -name "*kernel_trace.csv" -exec python tools/trace_overlap.py {} \; | tee $out/dp_trace_overlap.txt ; done
                python - $out <<'PY'
import csv, sys
o = sys.argv[1]
def load(v):
    d = {}
    for r in csv.DictReader(open(f"{o}/dp_kernel_stats_{v}.csv")):
        d[r["Name"].split("(")[0][:70]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    return d
a, b = load("base"), load("emul")
rows = sorted(((b.get(k, (0, 0))[1] - v[1], k, v, b.get(k, (0, 0))) for k, v in a.items()), reverse=True)
print("kernel                                                                  calls   base ms   emul ms   delta")
for dlt, k, va, vb in rows[:14]:
    print(f"{k:70s} {va[0]:6d} {va[1]:9.2f} {vb[1]:9.2f} {dlt:+8.2f}")
PY
                ;;
    dual)       for pr in 0 1; do STREAM_PRIO=$pr timeout 600 python tools/dual_stream_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $out/dual_stream_probe.txt; done ;;
    rows)       timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -k "fddt or ln or row or layernorm" 2>&1 | tail -5 | tee $out/rows_tests.txt
                for l in $PWD/tools/libv_rows_old.so $PWD/ts-asr-whisper_amd/libdicow_hip.so $PWD/tools/libv_rows_d2.so $PWD/tools/libv_rows_old.so $PWD/ts-asr-whisper_amd/libdicow_hip.so $PWD/tools/libv_rows_d2.so; do [ -f $l ] && (echo "== $l"; DICOW_HIP_LIB=$l timeout 300 python tools/bench_rows.py 2>&1 | grep -v amdgpu.ids) | tee -a $out/bench_rows.txt; done ;;
    final)      # the round's evidence on ONE box (what is judged is copied into profiles/ afterwards)
                (timeout 2700 python -m pytest tests -m gpu -q > $out/gpu_tests.txt 2>&1; tail -3 $out/gpu_tests.txt)
                python bench.py > $out/bench_default.json 2> $out/bench_default.err
                bash tools/prof_pmc.sh > /dev/null 2>&1; cp gpurun_out/pmc_summary.json $out/pmc_hbm_traffic.json
                bash tools/prof_step.sh > $out/prof_step.txt 2>&1; cp gpurun_out/kernel_stats.csv $out/kernel_stats.csv
                bash tools/prof_encfwd.sh > $out/prof_encfwd.txt 2>&1; cp gpurun_out/encfwd_kernel_stats.csv $out/encfwd_kernel_stats.csv
                python bench.py --preflight > $out/preflight_1gpu.json 2> $out/preflight.err
                python bench.py --from-audio --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_from_audio.json 2>/dev/null
                for v in se ctc preheat; do python bench.py --$v --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err; done
                python bench.py --model whisper-base --batch 8 --no-cpu-baseline > $out/bench_base_b8.json 2> $out/base.err
                python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > $out/bench_base_b8_graph.json 2>> $out/base.err
                bash tools/prof_pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma_summary.json $out/pmc_mfma_lds.json 2>/dev/null
                bash tools/prof_pmc_l2.sh > /dev/null 2>&1; cp gpurun_out/pmc_l2_summary.json $out/pmc_l2_hit_rate.json 2>/dev/null
                python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --split-streams > $out/bench_split_streams.json 2>/dev/null
                timeout 900 python tools/ab_split.py 6 3 2>&1 | grep -v amdgpu.ids > $out/ab_split.txt
                python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_default_again.json 2>/dev/null
                for f in $out/bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], (d.get('kernels') or {}).get('gemm_tn_kernel',{}).get('tflops'), (d.get('encoder_forward') or {}).get('ms'), (d.get('encoder_forward_train') or {}).get('ms'), d.get('power'))"; done ;;
    final2)     # the last session's evidence (encoder forward on two streams by default): one-stream runs beside the default ones; per-kernel profiles
                # and counter passes with DICOW_SPLIT_FWD=0 so that a kernel's duration is its own and not that of two overlapping launches
                (timeout 2700 python -m pytest tests -m gpu -q > $out/gpu_tests.txt 2>&1; tail -3 $out/gpu_tests.txt)
                python bench.py > $out/bench_default.json 2> $out/bench_default.err
                python bench.py --no-split-fwd --no-extra --no-cpu-baseline > $out/bench_one_stream.json 2>/dev/null
                DICOW_SPLIT_FWD=0 bash tools/prof_pmc.sh > /dev/null 2>&1; cp gpurun_out/pmc_summary.json $out/pmc_hbm_traffic.json
                DICOW_SPLIT_FWD=0 bash tools/prof_step.sh > $out/prof_step.txt 2>&1; cp gpurun_out/kernel_stats.csv $out/kernel_stats.csv
                bash tools/prof_step.sh > $out/prof_step_two_streams.txt 2>&1; cp gpurun_out/kernel_stats.csv $out/kernel_stats_two_streams.csv
                DICOW_SPLIT_FWD=0 bash tools/prof_encfwd.sh > $out/prof_encfwd.txt 2>&1; cp gpurun_out/encfwd_kernel_stats.csv $out/encfwd_kernel_stats.csv
                timeout 900 python tools/ab_split_fwd.py 2 2>&1 | grep -v amdgpu.ids > $out/ab_split_fwd.txt
                python bench.py --preflight > $out/preflight_1gpu.json 2> $out/preflight.err
                python bench.py --from-audio --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_from_audio.json 2>/dev/null
                for v in se ctc preheat; do python bench.py --$v --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err; done
                python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline > $out/bench_base_b8_graph.json 2> $out/base.err
                DICOW_SPLIT_FWD=0 bash tools/prof_pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma_summary.json $out/pmc_mfma_lds.json 2>/dev/null
                python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_default_again.json 2>/dev/null
                for f in $out/bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], (d.get('kernels') or {}).get('gemm_tn_kernel',{}).get('tflops'), (d.get('encoder_forward') or {}).get('ms'), (d.get('encoder_forward') or {}).get('mfma_frac'), (d.get('encoder_forward_train') or {}).get('ms'), (d.get('encoder_forward_train') or {}).get('mfma_frac'), d.get('power'))"; done ;;
    final3)     # evidence on the final tree (encoder forward + backward chain + frozen decoder as two half batches on two streams by default):
                # suite, default line, one-stream line, interleaved A/B of the three switches, kernel stats of the default and of the one-stream step
                (timeout 2700 python -m pytest tests -m gpu -q > $out/gpu_tests.txt 2>&1; tail -n 3 $out/gpu_tests.txt)
                python bench.py > $out/bench_default.json 2> $out/bench_default.err
                for rep in 1 2; do for v in all bwd_off dec_off one_stream; do e="DICOW_SPLIT_FWD=1"; [ $v = bwd_off ] && e="DICOW_SPLIT_BWD=0"; [ $v = dec_off ] && e="DICOW_SPLIT_BWD=0 DICOW_SPLIT_DEC=0"; [ $v = one_stream ] && e="DICOW_SPLIT_FWD=0"
                  env $e python bench.py --steps 15 --warmup 4 --no-extra --no-cpu-baseline 2>/dev/null | tail -n 1 > $out/bench_ab_${v}_$rep.json; done; done
                DICOW_SPLIT_FWD=0 bash tools/prof_step.sh --no-extra > $out/prof_step.txt 2>&1; cp gpurun_out/kernel_stats.csv $out/kernel_stats.csv
                bash tools/prof_step.sh --no-extra > $out/prof_step_two_streams.txt 2>&1; cp gpurun_out/kernel_stats.csv $out/kernel_stats_two_streams.csv
                for v in se ctc preheat; do python bench.py --$v --steps 8 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_$v.json 2> $out/bench_$v.err; done
                python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_default_again.json 2>/dev/null
                for f in $out/bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], (d.get('encoder_forward') or {}).get('ms'), (d.get('encoder_forward') or {}).get('mfma_frac'), (d.get('encoder_forward_train') or {}).get('ms'), (d.get('encoder_forward_train') or {}).get('mfma_frac'), (d.get('power') or {}).get('sclk_mhz_mean'))" | tee -a $out/summary.txt; done ;;
    # ---- last sessions: scheduling experiments around the two half-batch streams (profiles/r06_split_bwd.txt, r06_base_kernel_table.txt, r06_dp_emulated.txt)
    ab_env)     # tools/r06_call.sh <tag> ab_env    with AB_ENVS="name1:VAR=val,VAR2=val name2:..." AB_ARGS="bench.py args" AB_REPS=2   -> interleaved bench lines
                for rep in $(seq 1 ${AB_REPS:-2}); do for spec in $AB_ENVS; do n=${spec%%:*}; e=$(echo "${spec#*:}" | tr ',' ' ')
                  env $e timeout 300 python bench.py ${AB_ARGS:---steps 15 --warmup 4 --no-extra --no-cpu-baseline --no-one-stream-ref --profile-steps 1} 2>$out/err_$n.txt | tail -n 1 > $out/b_${n}_$rep.json
                  python -c "
import json; d=json.loads(open('$out/b_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n', d.get('value'), d.get('ms_per_step'), d.get('ms_per_step_median'), 'loss', d.get('loss'), (d.get('power') or {}).get('sclk_mhz_mean'), (d.get('encoder_forward_train') or {}).get('ms'))" | tee -a $out/ab.txt; done; done ;;
                # used as:  basecap  AB_ARGS="--model whisper-base --batch 8 --graph --no-extra --no-cpu-baseline --no-power --steps 30 --warmup 5"
                #                    AB_ENVS="none:DICOW_SPLIT_IN_CAPTURE=0 fwd:DICOW_SPLIT_IN_CAPTURE=1,DICOW_SPLIT_FWD_MIN_ROWS=6000,DICOW_SPLIT_DEC=0,DICOW_SPLIT_BWD=0 all:DICOW_SPLIT_IN_CAPTURE=1,DICOW_SPLIT_FWD_MIN_ROWS=6000"
                #           wgrad3   AB_ENVS="alt:DICOW_SPLIT_BWD_WGRAD=alt third:DICOW_SPLIT_BWD_WGRAD=third main:DICOW_SPLIT_BWD_WGRAD=main"
                #           noreduce AB_ENVS="base1:DICOW_SPLIT_FWD=1 base0:DICOW_SPLIT_FWD=0 nored1:DICOW_HIP_LIB=$PWD/tools/libv_noreduce.so nored0:DICOW_HIP_LIB=$PWD/tools/libv_noreduce.so,DICOW_SPLIT_FWD=0"
                #                    (tools/build_var.sh noreduce "" "" "" "-DDICOW_SKIP_REDUCE_MULTI" first)
    cus)        for rep in 1 2; do for c in 0 128 160 192 224; do timeout 300 python bench.py --gemm-cus $c --steps 12 --warmup 4 --no-extra --no-cpu-baseline --no-one-stream-ref 2>/dev/null | tail -n 1 > $out/b_${c}_$rep.json
                  python -c "
import json; d=json.loads(open('$out/b_${c}_$rep.json').read().strip().splitlines()[-1]); print('cus$c', d['ms_per_step'], (d.get('encoder_forward_train') or {}).get('ms'))" | tee -a $out/ab.txt; done; done ;;
    dp_split)   timeout 1200 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_rccl.py tests/test_gpu_dp.py -x -q 2>&1 | tail -n 6 | tee $out/tests.txt
                echo "== two half-batch streams (default)" | tee $out/dp_emulated.txt; DP_EMUL_REPS=2 timeout 1200 python tools/dp_emulate.py 100 2>&1 | grep -v amdgpu.ids | tee -a $out/dp_emulated.txt
                echo "== one stream (DICOW_SPLIT_FWD=0)" | tee -a $out/dp_emulated.txt; DICOW_SPLIT_FWD=0 DP_EMUL_REPS=1 timeout 1200 python tools/dp_emulate.py 100 2>&1 | grep -v amdgpu.ids | tee -a $out/dp_emulated.txt ;;
    conc)       (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/conc_$tag -o g -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --no-power --no-one-stream-ref --steps 10 --warmup 3 --profile-steps 1 > $GRAFT_REPO_ROOT/$out/conc_bench.json 2>/dev/null)
                find /tmp/conc_$tag -name "*kernel_trace.csv" -exec python tools/trace_concurrency.py {} \; | tee $out/trace_concurrency.txt ;;
    split_tests) timeout 900 python -m pytest tests/test_gpu_split_forward.py -x -q 2>&1 | tail -n 5 | tee $out/split_tests.txt ;;
    epi)        DICOW_HIP_LIB=$PWD/tools/libv_ntabl.so timeout 900 python tools/ab_epilogues.py 2>&1 | grep -v amdgpu.ids | tee -a $out/ab_epilogues.txt ;;
    base_prof)  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_base_$tag -o base -- python $GRAFT_REPO_ROOT/bench.py --model whisper-base --batch 8 --graph --no-extra --no-cpu-baseline --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$out/base_prof_bench.json 2>$GRAFT_REPO_ROOT/$out/base_prof_err.txt)
                find /tmp/prof_base_$tag -name "*kernel_stats.csv" -exec cp {} $out/base_kernel_stats.csv \; ; head -30 $out/base_kernel_stats.csv | cut -c1-170 ;;
    dual_step)  timeout 900 python tools/dual_stream_step_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/dual_stream_step.txt ;;
    split)      timeout 900 python tools/ab_split.py 6 3 2>&1 | grep -v amdgpu.ids | tee $out/ab_split.txt ;;
    split_var)  for e in "DICOW_SPLIT_GROUP=1" "DICOW_SPLIT_GROUP=2" "DICOW_SPLIT_GROUP=4" "DICOW_SPLIT_LEAD_FIRST=0 DICOW_SPLIT_GROUP=2" "DICOW_SPLIT_NOWAIT=1"; do echo "== $e" | tee -a $out/ab_split_var.txt
                  env $e timeout 900 python tools/ab_split.py 6 2 2>&1 | grep -v amdgpu.ids | tee -a $out/ab_split_var.txt; done ;;
    wgs)        for e in "DICOW_WGRAD_STREAM_PRIORITY=-1" "DICOW_WGRAD_STREAM_PRIORITY=0"; do echo "== $e" | tee -a $out/ab_wgrad_stream.txt
                  env $e timeout 900 python tools/ab_wgrad_stream.py 6 3 2>&1 | grep -v amdgpu.ids | tee -a $out/ab_wgrad_stream.txt; done ;;
    base_fused) for rep in 1 2; do for l in ts-asr-whisper_amd/libdicow_hip.so tools/libv_fmin512.so; do echo -n "$l: " | tee -a $out/base_fused.txt
                  DICOW_HIP_LIB=$PWD/$l python bench.py --model whisper-base --batch 8 --graph --no-extra --no-cpu-baseline --no-power --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" | tee -a $out/base_fused.txt; done; done ;;
    gaps)       (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_$tag -o g -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --no-power --steps 12 --warmup 3 --profile-steps 1 > $GRAFT_REPO_ROOT/$out/gaps_bench.json 2>/dev/null)
                find /tmp/gaps_$tag -name "*kernel_trace.csv" -exec python tools/trace_gaps.py {} \; | tee $out/trace_gaps.txt ;;
    gpu_tests)  timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $out/gpu_tests.txt ;;
    bench)      timeout 900 python bench.py 2>&1 | tail -3 | tee $out/bench_default.json ;;
    bench2)     timeout 900 python bench.py 2>&1 | tail -1 | tee $out/bench_default_again.json ;;
    ab_step)    for i in 1 2; do DICOW_ATTN_BWD_FUSED=0 timeout 600 python bench.py --no-extra --no-cpu-baseline 2>&1 | tail -1 | tee -a $out/ab_step_unfused.json
                                 DICOW_ATTN_BWD_FUSED=1 timeout 600 python bench.py --no-extra --no-cpu-baseline 2>&1 | tail -1 | tee -a $out/ab_step_fused.json; done ;;
    prof)       (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o step -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$out/prof_bench.json 2>$GRAFT_REPO_ROOT/$out/prof_err.txt)
                find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \; ;;
    *) echo "unknown step $step" ;;
  esac
done
