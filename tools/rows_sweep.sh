for cfg in "4 3" "4 2" "4 4" "2 3" "2 6" "2 4" "2 8"; do set -- $cfg; echo "R=$1 cap=$2: $(DICOW_HIP_LIB=$PWD/tools/libvr_abl.so DICOW_ROW_R=$1 DICOW_ROW_CAP=$2 python tools/bench_rows.py 2>/dev/null | grep 'fwd FDDT+LN')"; done
DICOW_HIP_LIB=$PWD/tools/libvr_abl.so python tools/bench_rows.py 2>/dev/null
