"""The bench.py output contract the round driver parses (one JSON line on stdout; keys, types, derived values), on a small
workload so that it runs in seconds.  Run with `pytest -m gpu`."""
import json
import os
import subprocess
import sys

import pytest
import torch

from tests.util import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_fields():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "whisper-tiny", "--batch", "2", "--labels", "16",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["unit"] == "utt/s"
    assert "workload" in d["config"] and d["config"]["global_batch"] == 2 and "model" not in d["config"]
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]          # whole-job utterances / s
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["avg_launch_ms"] > 0 and "traffic" in rf
    assert d["ms_per_step_median"] > 0 and d["per_rank_ms_per_step"] == [d["ms_per_step"]]
    assert d["loss"] == d["loss"]                                                        # finite
    sm = rf["sustained_mfma_only"]                                                      # the matrix pipe's ceiling under the power cap
    assert 1000.0 < sm["tflops"] < rf["peak"] and abs(sm["frac_of_it"] - rf["achieved"] / sm["tflops"]) < 1e-3
    assert d["one_stream_reference"] is None and d["config"]["two_half_batch_streams"] is False   # (a 2 x 1500-row batch runs on one stream)
    assert "power" in d                                                                 # rocm-smi samples of the timed region, or None
    if d["power"] is not None:
        assert 50.0 < d["power"]["board_w_mean"] < 2000.0 and 100.0 < d["power"]["sclk_mhz_mean"] <= 2500.0 and d["power"]["samples"] >= 1


def test_other_workloads_legs_report_step_time_and_throughput():
    """`other_workloads` of the default headline line (BASELINE.json configs[1] and configs[4]'s per-rank share, each timed as a process
    of its own after the headline): the leg runner on a seconds-sized stand-in, and the real legs' command lines."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import bench
    legs = bench.OTHER_WORKLOADS
    assert legs["whisper_base_b8_graph"][:5] == ["--model", "whisper-base", "--batch", "8", "--graph"] and "--se" in legs["se_dicow_b16"]
    assert all(int(v[v.index("--steps") + 1]) <= 10 for v in legs.values())
    got = bench.other_workloads({"tiny": ["--model", "whisper-tiny", "--batch", "2", "--labels", "16", "--steps", "2", "--warmup", "1"],
                                 "broken": ["--model", "no-such-model"]})
    t = got["tiny"]
    assert t["ms_per_step"] > 0 and abs(t["utt_s"] - 2 * 1e3 / t["ms_per_step"]) < 1e-2 * t["utt_s"] and t["batch"] == 2 and t["steps"] == 2
    assert t["loss"] == t["loss"] and "whisper-tiny" in t["workload"] and t["wall_s"] > 0
    assert got["broken"]["ms_per_step"] is None and "note" in got["broken"]                # a failed leg says why; the line still prints


def test_bench_gpus_2_launches_two_ranks_by_itself():
    """`bench.py --gpus 2` with no rendezvous in the environment: the launcher starts both ranks; on this one-GPU box they share
    device 0 and exchange over gloo (DICOW_BENCH_SHARE_GPU=1), on a multi-GPU node the same command runs RCCL."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, DICOW_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "whisper-tiny", "--batch", "2",
                        "--labels", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert len(d["per_rank_ms_per_step"]) == 2 and len(d["allreduce"]["exposed_ms_per_step"]) == 2
    assert d["allreduce"]["bytes_per_step"] > 0 and d["scaling"] == "weak"
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]          # whole-job utterances / s over both ranks
    assert "cpu_baseline" not in d                                                      # rank 0 at N = 1 only


def test_bench_preflight_lists_the_bucket_schedule_and_a_failed_check_fails_the_run():
    """`bench.py --preflight` on the GPU box (one rank, no process group): device report + the gradient bucket schedule of the
    model; and a two-rank run (shared GPU, gloo) with --max-exposed-frac -1 (a bound no run can meet) must exit non-zero with ONE JSON line carrying "error"."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preflight", "--model", "whisper-tiny", "--batch", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    gx = d["gradient_exchange"]
    assert d["preflight"] is True and d["n_gpus"] == 1 and d["ranks"][0]["device_ordinal"] == 0 and d["ranks"][0]["cus"] >= 64
    assert gx["buckets"] == len(gx["schedule_in_backward_order"]) >= 6 and gx["bytes_per_step"] == sum(b["bytes"] for b in gx["schedule_in_backward_order"])
    assert gx["schedule_in_backward_order"][0]["bucket"] == "final_ln" and gx["schedule_in_backward_order"][-1]["bucket"] == "stem"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "whisper-tiny", "--batch", "2",
                        "--labels", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--max-exposed-frac", "-1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, DICOW_BENCH_SHARE_GPU="1"))
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-1500:])
    d = json.loads(lines[0])
    assert "error" in d and "gradient exchange" in d["error"] and d["n_gpus"] == 2


def test_graft_entry_smoke_runs_and_checks_against_the_oracle():
    """__graft_entry__.smoke(): one small fwd+bwd of the hot path on cuda:0 compared with the oracle (raises on mismatch)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
