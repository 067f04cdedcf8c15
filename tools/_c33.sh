cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" ph1 ph2 ph3 ph4; do
  if [ -n "$v" ]; then export DICOW_HIP_LIB=$PWD/tools/libv_$v.so; else unset DICOW_HIP_LIB; fi
  echo "${v:-shipped}: $(python tools/bench_res_shapes.py 2>/dev/null | tail -1)"
done
done
for v in "" ph1 ph2 ph3; do
  if [ -n "$v" ]; then export DICOW_HIP_LIB=$PWD/tools/libv_$v.so; else unset DICOW_HIP_LIB; fi
  echo "encfwd ${v:-shipped}: $(python tools/enc_fwd.py 2>/dev/null | tail -1)"
done
