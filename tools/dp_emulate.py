"""One-GPU rehearsal of the 8-rank gradient exchange: the headline step
  (a) plain (no process group),
  (b) as rank 0 of a one-rank RCCL group with the bucketed side-stream reducer forced (34 all-reduces of the flat store per step),
  (c..) the same with the fabric emulator behind every bucket (bench.py --emulate-fabric-gbps G: 16 workgroups rewrite the bucket in
        place twice, paced to G GB/s of algorithm bandwidth -- the CU / HBM / power load of a ring all-reduce over xGMI),
and what each implies for 8 GPUs: every rank of a real run does this work and nothing else waits on the network as long as the
exposed wait stays ~0, so  throughput(8) <= 8 x batch / step_time(emulated)  and the scaling factor vs one plain GPU is
8 x step(plain) / step(emulated).  python tools/dp_emulate.py [rates ...]   -> stdout (copy into profiles/)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rates = [float(x) for x in sys.argv[1:]] or [100.0, 150.0, 200.0]
STEPS = os.environ.get("DP_EMUL_STEPS", "12")


def run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DICOW_FORCE_REDUCE", "DICOW_EMULATE_FABRIC_GBPS"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", STEPS, "--warmup", "4", "--no-cpu-baseline", "--no-extra"] + extra,
                       capture_output=True, text=True, env=e, cwd=ROOT)
    line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
    if line is None:
        raise SystemExit(f"bench failed ({extra}): {r.stderr[-800:]}")
    return json.loads(line)


rows = []
for rep in range(int(os.environ.get("DP_EMUL_REPS", "2"))):
    rows.append(("plain (no process group)", run([])))
    rows.append(("one-rank RCCL group, reducer forced", run(["--gpus", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                                                                       "MASTER_PORT": "29541", "DICOW_FORCE_REDUCE": "1"})))
    rows.append(("reducer forced, GEMM grids on 240 CUs", run(["--gpus", "1", "--gemm-cus", "240"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                                                                                              "MASTER_PORT": "29542", "DICOW_FORCE_REDUCE": "1"})))
    for g in rates:                                # (240 CUs for the persistent GEMMs: what GradReducer sets for N > 1 -- the 16 channel workgroups own their CUs)
        rows.append((f"+ fabric emulator at {g:.0f} GB/s", run(["--emulate-fabric-gbps", str(g), "--gemm-cus", "240"])))
plain = min(d["ms_per_step"] for n, d in rows if n.startswith("plain"))
base_busy = min(((d["allreduce"].get("buckets_per_rank") or [None])[0] or {}).get("busy_ms_per_step", 0.0) for n, d in rows if n.startswith("reducer forced, GEMM"))
print(f"{'':42s} {'ms/step':>8s} {'utt/s':>7s} {'exposed':>8s} {'busy':>8s} {'lag sum':>8s} {'board W':>8s} {'sclk':>6s}   implied at 8 GPUs")
for n, d in rows:
    ar = d["allreduce"]
    b = (ar.get("buckets_per_rank") or [None])[0] or {}
    pw = d.get("power") or {}
    x8 = 8 * plain / d["ms_per_step"]
    print(f"{n:42s} {d['ms_per_step']:8.2f} {d['value']:7.2f} {ar['exposed_ms_per_step'][0]:8.3f} {b.get('busy_ms_per_step', 0.0):8.2f} "
          f"{(b.get('start_lag_ms') or {}).get('sum_per_step', 0.0):8.2f} {pw.get('board_w_mean', 0):8.0f} {pw.get('sclk_mhz_mean', 0):6.0f}   "
          f"{8 * d['value']:7.1f} utt/s = {x8:4.2f} x one plain GPU" + ("" if n.startswith("+") else "   (no fabric load)")
          + (f"   [emulator ran at {2549.2 / max(1e-9, b.get('busy_ms_per_step', 0.0) - base_busy):.0f} GB/s]" if n.startswith("+") else ""))
print(f"\n(ms: exposed = compute stream waiting for the side stream before the optimizer; busy = the side stream's collectives + emulator per step; "
      f"lag sum = total time ready buckets queued behind earlier ones per step.  2.55 GB of fp32 gradients per step in 34 buckets: at G GB/s the "
      f"exchange alone takes 2549 / G ms -- {', '.join(f'{2549 / g:.1f} ms at {g:.0f}' for g in rates)} -- against a {plain:.0f} ms step.)")
