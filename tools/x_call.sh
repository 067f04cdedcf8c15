out=gpurun_out/r06x6; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_realdims.py -x -q -s -k configs2 2>&1 > $out/realdims_b16.txt; grep -n "AssertionError" -A3 $out/realdims_b16.txt | head -30; grep "configs\[2\]" $out/realdims_b16.txt; tail -3 $out/realdims_b16.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $out/gpu_tests.txt
