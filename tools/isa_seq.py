"""One-letter-per-instruction view of a kernel's basic blocks: python tools/isa_seq.py file.s mangled_kernel_name
M mfma  e exp  c cvt_pk  x max  f fma  a add_f32  v other VALU  D ds_read  d other ds  B buffer  w waitcnt  # barrier  n nop  J branch  s scalar  S scratch"""
import re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"(.*?)\.end_amdhsa_kernel", s, re.S)
for k in ["next_free_vgpr", "group_segment_fixed_size", "private_segment_fixed_size"]:
    print(k, re.search(k + r"\s+(\S+)", m.group(1)).group(1))
i = s.index("\n" + name + ":"); j = s.index("s_endpgm", i)
out = []
for l in s[i:j].split("\n"):
    if re.match(r"^\.LBB", l):
        out.append("\n|" + l.split(":")[0] + ("*" if "Loop" in l else "") + "| ")
        continue
    if not l.startswith("\t") or not l.strip() or l.strip().startswith((".", ";")):
        continue
    op = l.split()[0]
    for pre, ch in (("v_mfma", "M"), ("v_exp", "e"), ("v_cvt_pk", "c"), ("v_max", "x"), ("v_fma", "f"), ("v_add_f32", "a"), ("scratch_", "S"), ("v_", "v"),
                    ("ds_read", "D"), ("ds_", "d"), ("buffer", "B"), ("s_waitcnt", "w"), ("s_barrier", "#"), ("s_nop", "n"),
                    ("s_cbranch", "J"), ("s_branch", "J")):
        if op.startswith(pre):
            out.append(ch); break
    else:
        out.append("s")
print("".join(out))
