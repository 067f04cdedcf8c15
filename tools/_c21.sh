cd $GRAFT_REPO_ROOT
for v in ps0 ps2 ps5; do echo "== $v"; DICOW_HIP_LIB=tools/libv_$v.so python tools/profile_ntr.py 2>&1 | grep -A1 -E "N1280 K1280 plain|N1280 K5120 plain|N3840 K1280 plain" | grep -v "^--"; done
for r in 1 2; do for v in "" tools/libv_s2.so tools/libv_s5.so tools/libv_s10.so; do echo "enc_fwd lib=[$v]: $(DICOW_HIP_LIB=$v python tools/enc_fwd.py 20 2>/dev/null | tail -1 | cut -c1-40)"; done; done
