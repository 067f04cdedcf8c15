"""Instruction mix of one kernel in a device-only assembly listing:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only gemm.hip -o /tmp/gemm.s;  python tools/count_isa.py /tmp/gemm.s _Z15gemm_ntr_kernelILi3ELi4ELi4EEv15dicow_gemm_args"""
import sys, re, collections
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index("\n" + name + ":")
j = s.index("s_endpgm", i)
body = s[i:j]
ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
c = collections.Counter(ins)
print(name, "instructions:", len(ins))
for k, v in c.most_common(40):
    print(f"  {k:32s} {v}")
