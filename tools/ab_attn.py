"""A/B builds of the attention kernels at the encoder shape: python tools/ab_attn.py label=lib.so ...  (tools/build_attn_variants.sh)"""
import os, subprocess, sys
specs = [a.split("=", 1) for a in sys.argv[1:]]
for rep in range(int(os.environ.get("REPS", "2"))):
    for label, lib in specs:
        env = dict(os.environ, DICOW_HIP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "tools/bench_attn.py"], env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("attn_")]
        print(f"{label:12s}", " | ".join(l.split(":")[1].strip().split("  ")[0] for l in lines) if lines else r.stderr[-300:], flush=True)
