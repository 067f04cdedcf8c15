#!/bin/bash
# round 5, call 2: hand-off probe (fixed poll), gemm_nt2 timeline + ablations
mkdir -p gpurun_out/r05b
O=gpurun_out/r05b
export TMPDIR=/tmp
timeout 90 tools/probe_handoff > $O/probe_handoff.txt 2>&1; echo "probe rc $?" >> $O/probe_handoff.txt
cat $O/probe_handoff.txt
DICOW_HIP_LIB=$PWD/tools/libv_prof.so timeout 300 python tools/profile_ntr.py > $O/timeline_ntr.txt 2>&1
DICOW_HIP_LIB=$PWD/tools/libv_nt2p.so timeout 300 python tools/profile_ntr.py > $O/timeline_nt2.txt 2>&1
cat $O/timeline_nt2.txt
ROUNDS=3 timeout 900 python tools/ab_nt2.py shipped=ts-asr-whisper_amd/libdicow_hip.so nt2=tools/libv_nt2.so one=tools/libv_nt2one.so nomfma=tools/libv_nt2a1.so noread=tools/libv_nt2a2.so nodma=tools/libv_nt2a4.so > $O/ab_nt2_abl.txt 2>&1
cat $O/ab_nt2_abl.txt
