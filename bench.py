#!/usr/bin/env python3
"""Benchmark of the DiCoW training step on MI355X (contract: see the task description / DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W

N > 1: run it either under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment; WORLD_SIZE must
equal N) or plainly -- without a rendezvous in the environment `python bench.py --gpus N` spawns the N ranks itself (one process
per GPU, LOCAL_RANK -> device, RCCL rendezvous on 127.0.0.1; what scripts/submit_slurm.sh:34 does with torchrun in the reference).

One "step" = forward + backward + gradient all-reduce + clip + AdamW on one synthetic batch of 30 s clips
(whisper-large-v3-turbo dims, per-GPU batch 16, decoder frozen, L=128: BASELINE.json configs[2]/[3]); inputs are
resident in HBM before the timed region.  Prints ONE JSON line (rank 0) with the metric, the roofline of the
dominant kernel (live HIP-event timing of every launch of it inside the timed region) and the CPU baseline
(the oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="whisper-large-v3-turbo")
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--labels", type=int, default=128)
    ap.add_argument("--se", action="store_true", help="SE-DiCoW (enrollment cross-attention, scb_layers=8), config 5")
    ap.add_argument("--ctc", action="store_true", help="recipe CTC auxiliary branch (ctc_weight 0.3, subsample, extra attention)")
    ap.add_argument("--preheat", action="store_true",
                    help="time the recipe's first phase instead (use_fddt_only_n_steps: only FDDT parameters train)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-fabric-gbps", type=float, default=0.0,
                    help="one-GPU rehearsal of the 8-rank exchange: run as rank 0 of a one-rank RCCL group with the bucketed reducer forced and, "
                         "behind every bucket's all-reduce, a 16-workgroup side-stream kernel that rewrites the bucket in place twice at this "
                         "algorithm bandwidth (GB/s) -- the CU, HBM and power load of a real ring all-reduce (dicow_fabric_emulate)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the other BASELINE.json workloads (configs[1] whisper-base B=8 from a hipGraph, configs[4]'s per-rank SE-DiCoW "
                         "B=16) that the default one-GPU headline run times after its own timed region (`other_workloads` of the line)")
    ap.add_argument("--no-power", action="store_true", help="do not sample rocm-smi (board power / shader clock) during the timed region")
    ap.add_argument("--cpu-sample", default="turbo-b1")
    ap.add_argument("--profile-steps", type=int, default=5,
                    help="instrumented steps run AFTER the timed region (per-launch HIP events of the dominant kernel: roofline)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (launch-bound configs)")
    ap.add_argument("--split-streams", action="store_true",
                    help="run each batch as two half batches on two HIP streams (trainer.SplitSync; same loss and gradients as the two halves "
                         "accumulated; measured -0.3...-1.8 ms per step, profiles/r06_split_streams.txt: not the default)")
    ap.add_argument("--no-one-stream-ref", action="store_true", help="skip the short one-stream reference run behind the timed region")
    ap.add_argument("--no-split-fwd", "--one-stream", dest="no_split_fwd", action="store_true",
                    help="the whole step on ONE stream (default: the row-parallel work of a large even batch -- encoder forward, the frozen decoder's "
                         "layers, the encoder backward's dgrad / attention / row-kernel chain -- runs as two half batches on two HIP streams into the "
                         "halves of the same full-batch buffers; weight gradients stay one full-batch launch per layer: engine.SPLIT_FWD / SPLIT_DEC / SPLIT_BWD)")
    ap.add_argument("--from-audio", action="store_true",
                    help="the step starts from 16 kHz waveforms resident in HBM: dicow_logmel -> BatchAugmenter (STNO segment "
                         "augmentation + joint SpecAug, collators.py:189-214) -> training step (reported beside the headline, never AS it)")
    ap.add_argument("--gemm-cus", type=int, default=0,
                    help="limit the persistent GEMM grids to this many CUs (what trainer.GradReducer does for N > 1: CUs left to the RCCL channels)")
    ap.add_argument("--preflight", action="store_true",
                    help="multi-GPU dress rehearsal: every rank reports its device / PCI bus id / link types / RCCL version / NCCL_* "
                         "environment, a 256 MB all-reduce is timed, the gradient bucket schedule is listed; one JSON line, exit 0")
    ap.add_argument("--max-exposed-frac", type=float, default=None,
                    help="N > 1: fail (non-zero exit, \"error\" in the JSON line) when a rank waits longer than this share of a step for the "
                         "gradient exchange (default 0.20 over RCCL; off in the shared-GPU gloo test mode unless given)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher self-test: rendezvous + one all-reduce per rank, no training step (runs without a GPU over gloo)")
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(a):
    """`python bench.py --gpus N` with no rendezvous in the environment: start one process per GPU and relay rank 0's JSON line.
    Every child is polled: the first non-zero exit (or the overall timeout) kills the others -- a rank that died before the
    rendezvous would otherwise leave rank 0 waiting in RCCL forever -- and the stderr tail of the failed ranks is shown."""
    import subprocess
    import tempfile
    port = _free_port()
    procs, errs = [], []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DICOW_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=errs[-1], text=True))
    deadline = time.time() + float(os.environ.get("DICOW_BENCH_LAUNCH_TIMEOUT", "3600"))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in rcs):
            failed = f"rank exit codes {rcs}"
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.time() > deadline:
            failed = f"timeout; rank exit codes so far {rcs}"
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        sys.stderr.write(f"bench.py: {failed}\n")
        for r, f in enumerate(errs):
            f.seek(0)
            tail = f.read()[-1500:]
            if tail.strip() and procs[r].returncode not in (0, -9):
                sys.stderr.write(f"---- rank {r} stderr (tail)\n{tail}\n")
        out0.seek(0)
        for ln in out0.read().splitlines():           # a run that failed its own checks still printed its line (with "error")
            if ln.startswith("{"):
                sys.stdout.write(ln + "\n")
        sys.stdout.flush()
        sys.exit(1)
    for f in errs:                                   # warnings of a clean run: relay rank 0's only
        f.seek(0)
    sys.stderr.write(errs[0].read())
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()


class KernelTimer:
    """HIP-event bracket around every launch of selected C-ABI entry points (events on the launch stream)."""

    def __init__(self, ops_mod, names):
        self.ops, self.names, self.rec, self.orig, self.on = ops_mod, names, {n: [] for n in names}, {}, False

    def install(self):
        for n in self.names:
            fn = getattr(self.ops, n)
            self.orig[n] = fn

            def wrapped(*a, __fn=fn, __n=n, **kw):
                if not self.on:
                    return __fn(*a, **kw)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = __fn(*a, **kw)
                e.record()
                self.rec[__n].append((s, e, self.work(__n, a, kw), self.key(__n, a, kw)))
                return r
            setattr(self.ops, n, wrapped)
        import ts_asr_whisper_amd.engine as eng      # engine calls through the module attribute `ops.<name>`
        assert eng.ops is self.ops
        # the pooled weight-gradient launch of an encoder layer (ops.TnGroup.run -> dicow_gemm_tn_group) counts as gemm_tn
        if "gemm_tn" in self.names:
            run0 = self.ops.TnGroup.run
            timer = self

            def run(grp):
                if not timer.on or not grp.items:
                    return run0(grp)
                work = sum(2.0 * it[3] * it[4] * it[5] for it in grp.items)
                key = (grp.items[0][3], sum(it[4] * it[5] for it in grp.items), len(grp.items), 1, 0, "pooled", "float32")
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = run0(grp)
                e.record()
                timer.rec["gemm_tn"].append((s, e, work, key))
                return r
            self.ops.TnGroup.run = run

    @staticmethod
    def work(name, a, kw):
        if name == "gemm_nt":
            return 2.0 * a[3] * a[4] * a[5] * kw.get("batch", 1)
        if name == "gemm_tn":
            return 2.0 * a[3] * a[4] * a[5] * kw.get("batch", 1)
        return 0.0

    @staticmethod
    def key(name, a, kw):
        epi = "+".join(k for k in ("bias", "residual", "aux") if kw.get(k) is not None)
        return (a[3], a[4], a[5], kw.get("batch", 1), kw.get("flags", 0), epi, str(a[2].dtype).replace("torch.", ""))

    def breakdown(self, name, steps):
        """per (shape, epilogue) totals, for `DICOW_BENCH_BREAKDOWN=1 python bench.py` (stderr)"""
        agg = {}
        for s, e, w, k in self.rec[name]:
            t = agg.setdefault(k, [0, 0.0, 0.0])
            t[0] += 1; t[1] += s.elapsed_time(e); t[2] += w
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        for k, (n, ms, fl) in rows:
            print(f"  {name} M{k[0]} N{k[1]} K{k[2]} batch{k[3]} flags{k[4]} [{k[5]}] {k[6]}: {n // steps}/step, "
                  f"{ms / steps:.3f} ms/step, {ms / n:.4f} ms each, {fl / ms / 1e9:.0f} TF", file=sys.stderr)

    def summary(self, name):
        rec = self.rec[name]
        if not rec:
            return None
        ms = [r[0].elapsed_time(r[1]) for r in rec]
        fl = [r[2] for r in rec]
        return {"launches": len(rec), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms), "flops": sum(fl),
                "tflops": sum(fl) / (sum(ms) * 1e-3) / 1e12 if sum(ms) > 0 else 0.0}


OTHER_WORKLOADS = {"whisper_base_b8_graph": ["--model", "whisper-base", "--batch", "8", "--graph", "--steps", "10", "--warmup", "4"],      # BASELINE.json configs[1]
                   "se_dicow_b16": ["--se", "--steps", "6", "--warmup", "2"]}                                                          # configs[4], one rank's share


def other_workloads(legs=None):
    """BASELINE.json configs[1] and configs[4] (per-rank form) under the same clock as the headline: each is THIS script run again as
    its own process (<= 10 timed steps, no CPU baseline, no extras), after the headline's timed region and instrumented steps are
    over.  Returns {name: {ms_per_step, utt_s, step_mfma_frac, loss, wall_s}} -- a failed leg reports why instead of a number."""
    import subprocess
    legs = legs or OTHER_WORKLOADS
    out = {}
    for name, extra in legs.items():
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-extra", "--no-cpu-baseline", "--no-power",
                                "--profile-steps", "1"] + extra, capture_output=True, text=True, timeout=240,
                               env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DICOW_BENCH_CHILD")})
            line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
            if line is None:
                out[name] = {"ms_per_step": None, "note": f"no line (rc {r.returncode}): {r.stderr[-300:]}"}
                continue
            d = json.loads(line)
            out[name] = {"ms_per_step": d["ms_per_step"], "utt_s": d["value"], "step_mfma_frac": d.get("step_mfma_frac"),
                         "batch": d["config"]["global_batch"], "steps": d["steps"], "loss": d.get("loss"), "workload": d["config"]["workload"],
                         "encoder_forward_ms": (d.get("encoder_forward") or {}).get("ms"), "wall_s": round(time.time() - t0, 1)}
        except Exception as ex:
            out[name] = {"ms_per_step": None, "note": f"failed: {ex!r}", "wall_s": round(time.time() - t0, 1)}
    return out


def cpu_baseline(cfg_name, labels):
    """Oracle (kind "port") timed on the host cores: fwd+bwd of whisper-large-v3-turbo dims, B=1 (bounded sample)."""
    import amd_pkg
    pkg = amd_pkg.load()
    from oracle import dicow_oracle as O
    cfg = pkg.DiCoWConfig.preset(cfg_name, use_pre_pos_fddt=True, non_target_fddt_value=0.5)
    ocfg = O.OracleConfig(**{k: getattr(cfg, k) for k in O.OracleConfig.__dataclass_fields__ if hasattr(cfg, k)})
    cores = os.cpu_count() or 1
    p = O.init_state(ocfg, seed=0)
    for n, t in p.items():
        if "decoder" not in n and n != "proj_out.weight" and t.is_floating_point():
            t.requires_grad_(True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, cfg.num_mel_bins, 3000, generator=g).clamp_(-1.5, 1.5)
    st = torch.softmax(torch.randn(1, 4, 1500, generator=g), 1)
    lab = torch.randint(0, 50257, (1, labels), generator=g)
    # SURVEY 8(d) asks for torch.set_num_threads(nproc).  Measured in round 5 on the pool's host (256 logical cores, profiles/r05_bench_default.json):
    # one oracle step takes 27 s at 64 threads and 589 s at 256 -- the restatement is thousands of small fp32 ops and oversubscribes -- so
    # the baseline is timed at min(nproc, 64) threads, the fastest setting, and says so; `cores` is the thread count actually used
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    times = []
    for it in range(2):
        t0 = time.time()
        out = O.model_forward(p, ocfg, x, st, lab, lab)
        out["loss"].backward()
        times.append(time.time() - t0)
        for t in p.values():
            t.grad = None
        if sum(times) > 40:
            break
    best = min(times)
    return {"value": 1.0 / best, "unit": "utt/s", "cores": threads, "kind": "port",
            "sample": f"oracle (plain PyTorch fp32 restatement) fwd+bwd, {cfg_name} dims, B=1, L={labels}, {len(times)} steps, "
                      f"best {best:.2f} s at {threads} threads (host has {cores} logical cores; at all {cores} the same step measured "
                      f"589 s in round 5: oversubscribed)"}


def cpu_config1():
    """BASELINE.json configs[0] on the host cores: whisper-tiny + FDDT, one synthetic 30 s clip, forward only (oracle, fp32)."""
    import amd_pkg
    pkg = amd_pkg.load()
    from oracle import dicow_oracle as O
    cfg = pkg.DiCoWConfig.preset("whisper-tiny", use_pre_pos_fddt=True, non_target_fddt_value=0.5)
    ocfg = O.OracleConfig(**{k: getattr(cfg, k) for k in O.OracleConfig.__dataclass_fields__ if hasattr(cfg, k)})
    p = O.init_state(ocfg, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, cfg.num_mel_bins, 3000, generator=g).clamp_(-1.5, 1.5)
    st = torch.softmax(torch.randn(1, 4, 1500, generator=g), 1)
    lab = torch.randint(0, 50257, (1, 64), generator=g)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    times = []
    with torch.no_grad():
        for _ in range(3):
            t0 = time.time()
            O.model_forward(p, ocfg, x, st, lab, lab)
            times.append(time.time() - t0)
    return {"value": round(1.0 / min(times), 3), "unit": "utt/s (forward only)", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"configs[0]: whisper-tiny + FDDT, B=1, L=64, fp32 oracle forward, best of {len(times)}: {min(times):.3f} s"}


class PowerSampler:
    """Board power and shader clock of this rank's GPU, sampled by rocm-smi from a side thread while the timed region runs
    (the MFMA-heavy kernels of the step run at the board's power cap: the clock they get is part of the measurement)."""
    def __init__(self, device):
        self.device, self.rows, self._stop, self._th = device, [], False, None

    def _run(self):
        import re, subprocess
        while not self._stop:
            try:
                r = subprocess.run(["rocm-smi", "-d", str(self.device), "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
                card = next(iter(json.loads(r.stdout).values()))
                w = next((float(v) for k, v in card.items() if "Power" in k and "(W)" in k), None)
                m = next((re.search(r"(\d+)\s*Mhz", str(v), re.I) for k, v in card.items() if k.lower().startswith("sclk clock speed")), None)
                if w is not None and m:
                    self.rows.append((w, float(m.group(1))))
            except Exception:
                pass
            time.sleep(0.25)

    def start(self):
        import threading
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()

    def stop(self):
        self._stop = True
        if self._th is not None:
            self._th.join(timeout=10)
        if not self.rows:
            return None
        n = len(self.rows)
        return {"board_w_mean": round(sum(r[0] for r in self.rows) / n, 1), "board_w_max": max(r[0] for r in self.rows),
                "sclk_mhz_mean": round(sum(r[1] for r in self.rows) / n, 1), "sclk_mhz_min": min(r[1] for r in self.rows), "samples": n,
                "note": "rocm-smi samples during the timed region (board power cap 1400 W, peak shader clock 2400 MHz)"}


PMC_FILES = ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01g_pmc_hbm_traffic.json")      # newest first
TRAFFIC_SOURCE = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/prof_pmc.sh; fabric-side bytes "
                  "per persistent NT GEMM launch, FETCH_SIZE x2 per the gfx950 correction)")


def pmc_traffic(live):
    """HBM-side bytes per launch of the dominant kernel class from the committed PMC summary (counters cannot be read live).
    `live`: {kernel instantiation: launches} of THIS run (ops.gemm_dispatch_log()).  The summary is only quoted when it was
    taken on the kernels that ran here: every gemm_ntr instantiation of the live run must be in the file and vice versa,
    and the per-instantiation launch mix must agree (the file's numbers are weighted with the LIVE launch counts)."""
    global TRAFFIC_SOURCE
    try:
        name = next(n for n in PMC_FILES if os.path.exists(os.path.join(ROOT, "profiles", n)))
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
        filed = {k for k, v in d.items() if k.startswith("gemm_ntr") and v.get("hbm_read_bytes_per_launch_corrected") is not None}
        ran = {k for k in live if k.startswith("gemm_ntr")}
        if not ran or filed != ran:
            TRAFFIC_SOURCE = (f"REFUSED profiles/{name}: its persistent-GEMM kernels {sorted(filed)} are not the ones of this run "
                              f"{sorted(ran)} -- re-take the counter passes (tools/prof_pmc.sh)")
            return None
        TRAFFIC_SOURCE = TRAFFIC_SOURCE % name
        tot = n = 0
        for k in ran:
            v = d[k]
            tot += (v["hbm_read_bytes_per_launch_corrected"] + (v.get("hbm_write_bytes_per_launch") or 0)) * live[k]
            n += live[k]
        return round(tot / n) if n else None
    except Exception as ex:
        TRAFFIC_SOURCE = f"no usable PMC summary: {ex!r}"
        return None


def _topology_text():
    """`rocm-smi --showtopo` (link type / hops / weight between every pair of GPUs) -- best effort, rank 0 only."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        r = subprocess.run([exe, "--showtopo"], capture_output=True, text=True, timeout=60)
        return (r.stdout or r.stderr)[-6000:]
    except Exception as ex:
        return f"rocm-smi --showtopo failed: {ex!r}"


def _link_types(topo_text):
    """The 'Link Type between two GPUs' table of rocm-smi --showtopo as {"GPU0": ["0", "XGMI", ...], ...} (None if absent)."""
    out, on = {}, False
    for ln in topo_text.splitlines():
        if "Link Type between two GPUs" in ln:
            on = True
            continue
        if on:
            t = ln.split()
            if not t or t[0].startswith("="):
                if out:
                    break
                continue
            if t[0].startswith("GPU") and len(t) > 1 and not t[1].startswith("GPU"):
                out[t[0]] = t[1:]
    return out or None


def preflight(a, world, rank, local, ts, share):
    """One-shot rehearsal of everything a multi-GPU run depends on (a SCALE run on an 8-GPU node is a single attempt):
    who sits where, what fabric links them, which RCCL and which NCCL_* settings are live, what a large all-reduce moves per
    second, and what the gradient exchange will send.  Every rank contributes; rank 0 prints ONE JSON line; exit code 0."""
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None
    info = {"rank": rank, "local_rank": local, "device_ordinal": dev, "pid": os.getpid(), "host": os.uname().nodename}
    if dev is not None:
        pr = torch.cuda.get_device_properties(dev)
        info.update(name=pr.name, cus=pr.multi_processor_count, hbm_gb=round(pr.total_memory / 2 ** 30, 1),
                    pci_bus_id=":".join(f"{getattr(pr, k):02x}" for k in ("pci_domain_id", "pci_bus_id", "pci_device_id") if hasattr(pr, k)) or None,
                    uuid=str(getattr(pr, "uuid", "")) or None, gcn_arch=getattr(pr, "gcnArchName", None))
    info["env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_VISIBLE", "ROCR_VISIBLE", "DICOW_RCCL", "DICOW_FORCE"))}
    bw = None
    backend = dist.get_backend() if dist.is_initialized() else None
    if dist.is_initialized():
        n = (256 << 20) // 4 if (torch.cuda.is_available() and not os.environ.get("DICOW_PREFLIGHT_SMALL")) else (1 << 20) // 4
        buf = torch.ones(n, dtype=torch.float32, device="cuda" if (torch.cuda.is_available() and backend == "nccl") else "cpu")
        for _ in range(3):
            dist.all_reduce(buf)
            buf.fill_(1.0)
        if buf.is_cuda:
            torch.cuda.synchronize()
        dist.barrier()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(buf)
        if buf.is_cuda:
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        chk = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=buf.device)
        dist.all_reduce(chk)
        ok = bool(float(chk[0]) == world * (world + 1) / 2 and float(chk[-1]) == world * (world + 1) / 2)
        alg = buf.numel() * 4 / dt / 1e9
        bw = {"bytes": buf.numel() * 4, "ms": round(dt * 1e3, 3), "algbw_gbps": round(alg, 1),
              "busbw_gbps": round(alg * 2 * (world - 1) / world, 1) if world > 1 else 0.0, "result_ok": ok}
    info["allreduce_256mb"] = bw
    every = [None] * world
    if dist.is_initialized():
        dist.all_gather_object(every, info)
    else:
        every = [info]
    if rank == 0:
        topo = _topology_text()
        segs = [(nm, int(b_ - a_) * 4) for nm, a_, b_ in ts.store.segments] if ts is not None else []
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        emit(json.dumps({"preflight": True, "n_gpus": world, "backend": backend, "rccl_version": rccl, "torch": torch.__version__,
                          "shared_gpu_test_mode": bool(share), "ranks": every, "link_types": _link_types(topo), "topology": topo,
                          "gradient_exchange": {"buckets": len(segs), "bytes_per_step": sum(b for _, b in segs),
                                                "schedule_in_backward_order": [{"bucket": nm, "bytes": b} for nm, b in segs],
                                                "op": "all-reduce, AVG inside the collective on RCCL (sum + divide on gloo), fp32, in place, side stream",
                                                "gemm_cus_left_to_compute": a.gemm_cus or None,
                                                "replica_sync_at_startup": ts.replica_sync if ts is not None else None,
                                                "note": None if ts is not None else "no GPU: the model was not built, the schedule is not listed"},
                          "config": {"global_batch": a.batch * world, "parallelism": f"dp{world}"}}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def exposed_guard(rank_exposed_ms, step_ms, max_frac):
    """None, or why the run must fail: some rank's compute stream waits more than `max_frac` of a step for the gradient exchange."""
    if max_frac is None or not rank_exposed_ms:
        return None
    worst = max(rank_exposed_ms)
    if worst > max_frac * step_ms:
        return (f"a rank waits {worst:.2f} ms per step for the gradient exchange: more than {max_frac:.0%} of the {step_ms:.2f} ms step "
                f"(per rank: {list(rank_exposed_ms)}) -- the all-reduce is not hidden behind the backward pass")
    return None


def _replicas_agree(ts, world):
    """After an optimizer step every rank must hold bit-identical parameters (same averaged gradient, same update): a 64-bit
    checksum of the flat parameter store, all-gathered.  Returns (ok, checksums)."""
    from ts_asr_whisper_amd.trainer import _bit_checksum
    mine = _bit_checksum(ts.store.params) & 0x7FFFFFFFFFFFFFFF
    dev = ts.store.params.device if dist.get_backend() == "nccl" else "cpu"
    got = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(got, torch.tensor([mine], dtype=torch.int64, device=dev))
    sums = [int(g.item()) for g in got]
    return len(set(sums)) == 1, sums


def dry_launch(a, world, rank, local):
    """Launcher self-test (no training step): every rank joins the process group, checks its size, all-reduces its rank id."""
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world
    if use_gpu:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        t = torch.tensor([float(rank + 1)], device="cuda")
    else:
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)])
    assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        emit(json.dumps({"dry_launch": True, "n_gpus": dist.get_world_size(), "backend": dist.get_backend(),
                          "allreduce_sum": float(t), "expected_sum": a.gpus * (a.gpus + 1) / 2,
                          "config": {"global_batch": a.batch * a.gpus, "parallelism": f"dp{a.gpus}"}}))
    dist.destroy_process_group()


_REAL_STDOUT = None


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner ("RCCL version : ...", five lines) to file
    descriptor 1 when a communicator is created / destroyed, rocm-smi helpers may chat too: point fd 1 at stderr for the
    whole run and keep the real stdout for the JSON line."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    out = _REAL_STDOUT if _REAL_STDOUT is not None else sys.stdout
    out.write(line + "\n")
    out.flush()


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(a)                       # self-launch: one process per GPU
    _claim_stdout()
    if a.emulate_fabric_gbps > 0.0 and a.gpus == 1:               # rank 0 of a one-rank group, reducer forced, the emulator behind every bucket
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ["DICOW_FORCE_REDUCE"] = "1"
        os.environ["DICOW_EMULATE_FABRIC_GBPS"] = str(a.emulate_fabric_gbps)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and (a.gpus > 1 or world > 1):
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world} in the environment")
    if os.environ.get("DICOW_BENCH_FAIL_RANK") == str(rank):     # launcher self-test: this rank dies before the rendezvous
        sys.stderr.write(f"bench.py: rank {rank} failing on request (DICOW_BENCH_FAIL_RANK)\n")
        sys.exit(3)
    if a.dry_launch:
        return dry_launch(a, world, rank, local)
    if a.preflight and not torch.cuda.is_available():             # no GPU (tests): the rendezvous / all-reduce / report over gloo, no model
        if world > 1 or "RANK" in os.environ:
            dist.init_process_group("gloo")
        return preflight(a, world, rank, local, None, False)
    # DICOW_BENCH_SHARE_GPU=1 (tests on a one-GPU box): every rank uses device 0 and the exchange runs over gloo
    share = os.environ.get("DICOW_BENCH_SHARE_GPU") == "1"
    torch.cuda.set_device(0 if share else local)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the gradient all-reduce overlaps the backward pass and needs few channels; every RCCL channel is a workgroup that
        # keeps a CU from the persistent GEMM (trainer.GradReducer reserves DICOW_RCCL_CUS = 16 CUs for them)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
        if share:
            dist.init_process_group("gloo")
        else:
            import amd_pkg as _ap
            _ap.load()
            from ts_asr_whisper_amd.trainer import nccl_pg_options
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), pg_options=nccl_pg_options())
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
    import amd_pkg
    pkg = amd_pkg.load()
    from ts_asr_whisper_amd import ops
    from ts_asr_whisper_amd import engine as _engine
    from ts_asr_whisper_amd.trainer import TrainStep
    if a.no_split_fwd:
        _engine.SPLIT_FWD = False
    from ts_asr_whisper_amd.data import synthetic_batch

    over = dict(use_fddt=True, fddt_is_diagonal=True, use_pre_pos_fddt=True, fddt_init="suppressive", non_target_fddt_value=0.5)
    if a.se:
        over.update(use_enrollments=True, scb_layers=8)
    if a.ctc:
        over.update(ctc_weight=0.3, pre_ctc_sub_sample=True, additional_self_attention_layer=True, remove_timestamps_from_ctc=True)
    cfg = pkg.DiCoWConfig.preset(a.model, **over)
    torch.manual_seed(0)
    model = pkg.DiCoWForConditionalGeneration(cfg).cuda()
    model.tie_weights()
    prefixes = ("model.encoder.fddts", "model.encoder.initial_fddt") + (("model.encoder.ca_enrolls",) if a.se else ())
    ts = TrainStep(model, lr=2e-6, fddt_lr_multiplier=100.0, max_grad_norm=1.0, warmup_steps=2000, max_steps=40000,
                   preheat_prefixes=prefixes, use_fddt_only_n_steps=10 ** 9 if a.preheat else 0,
                   split_streams=a.split_streams, **({"graph": True} if a.graph else {}))
    if a.preflight:
        return preflight(a, world, rank, local, ts, share)
    batches = [synthetic_batch(cfg, a.batch, a.labels, seed=1000 + rank * 17 + i, mixed_length=a.se, enrollments=a.se)
               for i in range(2)]
    if a.gemm_cus:
        ops.set_gemm_cus(a.gemm_cus)
    # ---- the batch-preparation front end (reference local_datasets.py:196-214 feature extraction + collators.py:189-214
    # augmentation: CPU work of the data-loader workers there, kernels here).  Synthetic 30 s waveforms resident in HBM.
    from ts_asr_whisper_amd.features import log_mel, N_SAMPLES
    from ts_asr_whisper_amd.augment import BatchAugmenter
    gw = torch.Generator().manual_seed(77 + rank)
    waves = [(torch.randn(a.batch, N_SAMPLES, generator=gw) * 0.1).cuda() for _ in range(2)]
    augmenter = BatchAugmenter(stno_segment_augment_prob=1.0, spec_aug_prob=1.0)      # both augmentations on EVERY step (worst case)

    def front_end(i):
        b = dict(batches[i % 2])
        b["input_features"] = log_mel(waves[i % 2], cfg.num_mel_bins)
        return augmenter(b)

    def run_step(i, **kw):
        return ts.step(front_end(i) if a.from_audio else batches[i % 2], **kw)
    timer = KernelTimer(ops, ["gemm_nt", "gemm_tn"])
    timer.install()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    errors = []
    for i in range(a.warmup):
        loss = run_step(i)
        if i == 0 and world > 1:
            # the replicas saw different minibatches; after the exchange + update they must still be bit-identical
            ok, sums = _replicas_agree(ts, world)
            if not ok:
                errors.append(f"replicas differ after the first optimizer step (parameter checksums per rank {[hex(v) for v in sums]}): the gradient exchange is broken")
    sync()
    # ---- the timed region: exactly K steps, no per-launch instrumentation (one event per step boundary for the median)
    ts.reducer.time_exposed = True
    ts.reducer.time_buckets = world > 1 or ts.reducer.force
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    sampler = PowerSampler(local) if rank == 0 and not a.no_power else None
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    marks[0].record()
    host_s = 0.0
    for i in range(a.steps):
        h0 = time.perf_counter()
        loss = run_step(i)
        host_s += time.perf_counter() - h0                       # host time to ENQUEUE the step (no synchronisation inside): the lead it has over the GPU
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    power = sampler.stop() if sampler else None
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    med_ms = step_ms[len(step_ms) // 2] if a.steps % 2 else 0.5 * (step_ms[a.steps // 2 - 1] + step_ms[a.steps // 2])
    exposed_ms = ts.reducer.exposed_ms()
    ts.reducer.time_exposed = False
    bucket_rep = ts.reducer.bucket_report(a.steps)
    ts.reducer.time_buckets = False
    rank_buckets = [bucket_rep]
    if world > 1:
        rank_buckets = [None] * world
        dist.all_gather_object(rank_buckets, bucket_rep)
    rank_ms = [round(dt / a.steps * 1e3, 3)]
    rank_exposed = [round(exposed_ms, 3)]
    if world > 1:
        t = torch.tensor([dt, -dt, exposed_ms], device="cuda", dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        dt = max(float(e[0]) for e in every)
        rank_ms = [round(float(e[0]) / a.steps * 1e3, 3) for e in every]
        rank_exposed = [round(float(e[2]), 3) for e in every]
    # ---- instrumented steps OUTSIDE the timed region: per-launch HIP events of the GEMM classes (roofline)
    nprof = max(1, min(a.profile_steps, a.steps))
    timer.on = True
    split_was, ts.split_streams = ts.split_streams, False   # per-launch events of two overlapping streams would time the overlap, not the kernels
    fwd_split_was, _engine.SPLIT_FWD = _engine.SPLIT_FWD, False      # (the same for the encoder forward's two half-batch streams)
    for i in range(nprof):
        run_step(i, **({"eager": True} if a.graph else {}))
    sync()
    timer.on = False
    # ---- the same step on ONE stream, un-instrumented, right behind the timed region (same box, same clock state): what the two half-batch streams are worth
    one_stream = None
    if (fwd_split_was and not a.graph and not a.no_one_stream_ref and a.batch % 2 == 0
            and a.batch * cfg.max_source_positions >= _engine.SPLIT_FWD_MIN_ROWS):        # (only where the two streams were in use)
        n1 = max(2, min(6, a.steps))
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(n1 + 1)]
        run_step(0)
        sync()
        ev1[0].record()
        for i in range(n1):
            run_step(i)
            ev1[i + 1].record()
        sync()
        one_stream = {"ms_per_step": round(ev1[0].elapsed_time(ev1[n1]) / n1, 3), "steps": n1,
                      "what": "the same step with engine.SPLIT_FWD off (forward, frozen decoder and backward chain on one stream), timed after the timed region"}
    ts.split_streams = split_was
    _engine.SPLIT_FWD = fwd_split_was
    # ---- the front end by itself (HIP events on the launch stream, after the timed region): log-mel, augmentation
    fe = None
    if rank == 0:
        try:
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            for i in range(2):
                front_end(i)
            t_mel = t_aug = 0.0
            NFE = 6
            for i in range(NFE):
                b = dict(batches[i % 2])
                evs[0].record()
                b["input_features"] = log_mel(waves[i % 2], cfg.num_mel_bins)
                evs[1].record()
                augmenter(b)
                evs[2].record()
                torch.cuda.synchronize()
                t_mel += evs[0].elapsed_time(evs[1]); t_aug += evs[1].elapsed_time(evs[2])
            frames = N_SAMPLES // 160
            mel_flop = a.batch * frames * (2.0 * 400 * 201 * 2 + 2.0 * 201 * cfg.num_mel_bins)
            mel_bytes = a.batch * (N_SAMPLES * 4 + 2 * cfg.num_mel_bins * frames * 4 * 1.5)   # wave in; mel written, then read + rewritten by the normalisation pass
            fe = {"logmel_ms": round(t_mel / NFE, 4), "logmel_gflops": round(mel_flop / (t_mel / NFE) / 1e6, 1),
                  "logmel_algorithmic_mb": round(mel_bytes / 1e6, 1), "logmel_gbps": round(mel_bytes / (t_mel / NFE) / 1e6, 1),
                  "augment_ms": round(t_aug / NFE, 4),
                  "what": "dicow_logmel (the 400-point DFT as an exact-fp32 matrix product on v_mfma_f32_32x32x2_f32 + slaney mel + log / clamp) on B x 30 s of 16 kHz audio resident in HBM; "
                          "BatchAugmenter with the STNO segment augmentation and the joint SpecAug forced on (host planner incl.)",
                  "in_timed_region": bool(a.from_audio)}
        except Exception as ex:
            fe = {"logmel_ms": None, "note": f"failed: {ex!r}"}
    if world > 1:
        ok, sums = _replicas_agree(ts, world)
        if not ok:
            errors.append(f"replicas differ at the end of the run (parameter checksums per rank {[hex(v) for v in sums]})")
        lim = a.max_exposed_frac if a.max_exposed_frac is not None else (0.20 if dist.get_backend() == "nccl" else None)
        msg = exposed_guard(rank_exposed, dt / a.steps * 1e3, lim)
        if msg:
            errors.append(msg)
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        if errors:
            sys.exit(4)
        return
    ms = dt / a.steps * 1e3
    utts = a.batch * world * a.steps / dt
    nt, tn = timer.summary("gemm_nt"), timer.summary("gemm_tn")
    prof_ms = sum(r[0].elapsed_time(r[1]) for r in timer.rec["gemm_nt"])
    if os.environ.get("DICOW_BENCH_BREAKDOWN"):
        timer.breakdown("gemm_nt", nprof)
        timer.breakdown("gemm_tn", nprof)
    peak = 2500.0
    SUSTAINED_MFMA_TF = 1845.0
    TRAFFIC = None                                   # filled in after the encoder-forward leg (the committed counter passes cover the whole command)
    # algorithmic TFLOP per utterance of one step: 3 x encoder + 2 x (decoder + head) with the decoder frozen;
    # turbo: 3 x 2.2738 + 2 x 0.0841 = 6.99.  SE-DiCoW: the survey's 10.7 (3.51 encoder) for the headline model only.
    T_, D_, F_, Le, Ld = cfg.max_source_positions, cfg.d_model, cfg.encoder_ffn_dim, cfg.encoder_layers, cfg.decoder_layers
    Lx, V_ = a.labels, cfg.vocab_size
    enc_tf = (Le * (8 * T_ * D_ * D_ + 4 * T_ * T_ * D_ + 4 * T_ * D_ * F_) + 6 * (2 * T_) * cfg.num_mel_bins * D_ + 6 * T_ * D_ * D_) / 1e12
    dec_tf = (Ld * (12 * Lx * D_ * D_ + 4 * Lx * Lx * D_ + 4 * T_ * D_ * D_ + 4 * Lx * T_ * D_ + 4 * Lx * D_ * cfg.decoder_ffn_dim)
              + 2 * Lx * D_ * V_) / 1e12
    if a.se:
        tf_utt = (10.7 - (3.51 if a.preheat else 0.0)) if a.model.endswith("large-v3-turbo") else None
    else:
        tf_utt = (2 if a.preheat else 3) * enc_tf + 2 * dec_tf
    out = {
        "metric": f"train utterances/sec (30 s clips) {a.model} DiCoW" if not a.se else
                  "train utterances/sec (30 s clips) SE-DiCoW large-v3-turbo",
        "value": round(utts, 3), "unit": "utt/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms, 3), "ms_per_step_median": round(med_ms, 3),
        "value_at_median_step": round(a.batch * world * 1e3 / med_ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic (random-init weights, N(0,1) mel clamped to [-1.5,1.5], 3-speaker STNO process, random labels)",
        "config": {"workload": f"{a.model} DiCoW fine-tune step, per-GPU batch {a.batch}, L={a.labels}, decoder frozen, "
                               f"bf16 AMP{', SE-DiCoW scb_layers=8 mixed-length' if a.se else ''}{', CTC 0.3' if a.ctc else ''}{', preheat phase (FDDT-only training)' if a.preheat else ''}"
                               f"{', step replayed from a hipGraph' if a.graph else ''}{', from 16 kHz audio (log-mel + augmentation inside the step)' if a.from_audio else ''}",
                   "global_batch": a.batch * world, "parallelism": f"dp{world}", "split_streams": bool(a.split_streams), "two_half_batch_streams": bool(_engine.SPLIT_FWD and a.batch % 2 == 0 and a.batch * cfg.max_source_positions >= _engine.SPLIT_FWD_MIN_ROWS and not a.graph) and {"encoder_forward": "behind the speaker-communication layers" if a.se else True, "frozen_decoder_layers": bool(_engine.SPLIT_DEC), "encoder_backward_chain": bool(_engine.SPLIT_BWD) and ("above the speaker-communication layers" if a.se else True), "weight_gradients": "one full-batch pooled launch per layer"}, "trainable_params": sum(n for q, _, n, _ in ts.store.entries if q.requires_grad)},
        "loss": float(loss),
        "per_rank_ms_per_step": rank_ms,
        "one_stream_reference": one_stream,
        "host_enqueue_ms_per_step": round(host_s / a.steps * 1e3, 3),
        "allreduce": {"exposed_ms_per_step": rank_exposed,
                      "note": "time the compute stream waits for the side-stream RCCL buckets before the optimizer (0 at one rank)",
                      "backend": dist.get_backend() if dist.is_initialized() else None,
                      "forced_on_one_rank": bool(ts.reducer.force and world == 1),
                      "buckets_per_step": len(ts.reducer.seg) if (world > 1 or ts.reducer.force) else 0,
                      "gemm_cus": a.gemm_cus or None,
                      "bytes_per_step": 4 * ts.store.n_trainable if (world > 1 or ts.reducer.force) else 0,
                      # per rank: how long a bucket that was ready on the compute stream queued before its all-reduce started on the
                      # side stream, and how long the collectives themselves kept that stream busy (GradReducer.bucket_report)
                      "buckets_per_rank": rank_buckets if any(r is not None for r in rank_buckets) else None,
                      "emulated_fabric_gbps": ts.reducer.emulate_gbps or None},
        "roofline": {"bound": "mfma", "kernel": "gemm_ntr_kernel (persistent LDS-ring 256x256 / 192x320) / gemm_nt128t_kernel (bf16 MFMA 32x32x16; every forward Linear/conv GEMM and dgrad)",
                     "achieved": round(nt["tflops"], 1), "peak": peak, "unit": "TFLOP/s", "frac": round(nt["tflops"] / peak, 4),
                     "traffic": TRAFFIC, "avg_launch_ms": round(nt["avg_ms"], 4), "launches_per_step": nt["launches"] // nprof,
                     "share_of_step": round(prof_ms / nprof / ms, 3),
                     "measured_over": f"{nprof} instrumented steps after the timed region (HIP events around every launch, on the launch stream)",
                     "traffic_source": TRAFFIC_SOURCE,
                     # what the matrix pipe sustains on this board with NO memory traffic at all (tools/power_mfma.sh)
                     "sustained_mfma_only": {"tflops": SUSTAINED_MFMA_TF, "frac_of_it": round(nt["tflops"] / SUSTAINED_MFMA_TF, 4),
                                             "what": "register-only v_mfma_f32_32x32x16_bf16 loop on all 256 CUs, random bf16 operands, 5 s: the board "
                                                     "settles at 1.90 GHz (constant operands: 2.39 GHz, 2453 TF); profiles/r02_power_mfma.txt"}},
        "power": power,
        "front_end": fe,
        "kernels": {"gemm_tn_kernel": {"tflops": round(tn["tflops"], 1), "frac": round(tn["tflops"] / peak, 4),
                                       "share_of_step": round(tn["total_ms"] / nprof / ms, 3),
                                       "what": "every weight gradient incl. its split fix-up / reduce launches: gemm_tn256g_kernel (an encoder "
                                               "layer's four dW pooled in one persistent launch) + tn_group_fixup_kernel, gemm_tn256_kernel / "
                                               "gemm_tn_kernel + tn_reduce_kernel for the rest"}} if tn else {},
        # preheat phase: forward + dgrad only (2 x encoder + 2 x decoder), no encoder weight gradients
        "step_tflops": None if tf_utt is None else round(tf_utt * utts, 1),
    }
    out["step_mfma_frac"] = None if tf_utt is None else round(out["step_tflops"] / peak, 4)
    # north-star headline: MFMA utilisation of the FDDT-conditioned ENCODER FORWARD alone (no activations kept), same batch
    try:
        T_, D_, F_, Le, Mm = cfg.max_source_positions, cfg.d_model, cfg.encoder_ffn_dim, cfg.encoder_layers, cfg.num_mel_bins
        enc_flops = (Le * (8 * T_ * D_ * D_ + 4 * T_ * T_ * D_ + 4 * T_ * D_ * F_) + 6 * (2 * T_) * Mm * D_ + 6 * T_ * D_ * D_) * a.batch
        b0 = batches[0]
        kw_e = dict(stno_mask=b0["stno_mask"])
        if a.se:
            kw_e["enrollments"] = b0["enrollments"]
            enc_flops = None                                   # (the SE encoder's count differs; report the time only)
        def time_encoder(grad):
            # grad = False: torch.no_grad(), nothing kept for a backward pass (the inference-specialised forward: the next layer's
            # FDDT in the fc2 epilogue, no gelu' / pre-activation stores).  grad = True: the forward the TRAINING STEP runs
            # (need_grad: FDDT + LayerNorm row kernel, gelu' saved by fc1, every activation of every layer kept until the output
            # is dropped) -- the same launches as the forward half of a timed step, timed alone.
            with torch.set_grad_enabled(grad):
                for _ in range(2):
                    o_ = model.model.encoder(b0["input_features"], **kw_e)
                    del o_
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    o_ = model.model.encoder(b0["input_features"], **kw_e)
                    del o_
                e1.record()
                torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 5

        def enc_entry(ms_, note):
            return {"ms": round(ms_, 2), "batch": a.batch, "tflops": None if enc_flops is None else round(enc_flops / ms_ / 1e9, 1),
                    "mfma_frac": None if enc_flops is None else round(enc_flops / ms_ / 1e9 / peak, 4), "note": note}

        enc_ms = time_encoder(False)
        out["encoder_forward"] = enc_entry(enc_ms, "encoder forward only, torch.no_grad() (inference-specialised launches), algorithmic "
                                                   "FLOPs of SURVEY 8d / dense bf16 peak 2.5 PF")
        try:
            assert any(p_.requires_grad for p_ in model.model.encoder.parameters())
            out["encoder_forward_train"] = enc_entry(time_encoder(True), "encoder forward only, gradients enabled: the forward the training "
                                                     "step runs (activations, gelu' and LayerNorm statistics saved for the backward pass)")
        except Exception as ex:
            out["encoder_forward_train"] = {"ms": None, "note": f"failed: {ex!r}"}
    except Exception as ex:
        out["encoder_forward"] = {"ms": None, "note": f"failed: {ex!r}"}
    out["roofline"]["traffic"] = pmc_traffic(ops.gemm_dispatch_log())
    out["roofline"]["traffic_source"] = TRAFFIC_SOURCE
    headline = a.model.endswith("large-v3-turbo") and a.batch == 16 and not (a.se or a.ctc or a.preheat or a.graph or a.from_audio)
    if world == 1 and headline and not a.no_extra and "DICOW_BENCH_CHILD" not in os.environ:
        del model, ts, batches, waves                             # (the legs are processes of their own: give the memory back first)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out["other_workloads"] = other_workloads()
    if not a.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(a.model, a.labels)
        except Exception as ex:      # the bench line must still be printed
            out["cpu_baseline"] = {"value": None, "unit": "utt/s", "cores": 0, "kind": "port", "sample": f"failed: {ex!r}"}
        try:
            out["cpu_baseline"]["config1"] = cpu_config1()
        except Exception as ex:
            out["cpu_baseline"]["config1"] = {"value": None, "sample": f"failed: {ex!r}"}
    if dist.is_initialized():
        dist.destroy_process_group()
    if errors:
        out["error"] = "; ".join(errors)
    emit(json.dumps(out))
    if errors:
        sys.stderr.write("bench.py: " + out["error"] + "\n")
        sys.exit(4)


if __name__ == "__main__":
    main()
