#!/bin/bash
# kernel names / resources of the hipBLASLt kernels behind torch.matmul at the layer's shapes -> gpurun_out/lt_kernels.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/ltk
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ltk -- python $R/tools/lt_kernels.py > $R/gpurun_out/ltk.log 2>&1
f=$(ls $R/gpurun_out/ltk/*/*kernel_trace.csv | head -1)
python - "$f" > $R/gpurun_out/lt_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
seen = {}
for r in rows:
    n = r["Kernel_Name"]
    if "Cijk" in n or "gemm" in n.lower():
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = (n, r.get("Grid_Size_X"), r.get("Workgroup_Size_X"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"))
        seen.setdefault(k, []).append(d)
for k, v in seen.items():
    print(f"{min(v):9.1f} us x{len(v)}  grid {k[1]} wg {k[2]} lds {k[3]} vgpr {k[4]} agpr {k[5]} sgpr {k[6]} scratch {k[7]}\n    {k[0]}")
PY
rm -rf $R/gpurun_out/ltk
cat $R/gpurun_out/lt_kernels.txt
