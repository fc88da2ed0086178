"""GPU: bench.py's contract -- the JSON line of a small single-rank run, and the multi-rank path (one process per rank over
torch.distributed, barrier + max-over-ranks timing) exercised with two ranks sharing the one GPU of the test box
(WENET_BENCH_BACKEND=gloo), so that the driver's 8-GPU run is not the first run of that code."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _torchrun(nproc, port, bench_args, env, timeout):
    """one node, nproc ranks on 127.0.0.1; a rendezvous that fails (a port still held by an earlier test's store, eight HIP contexts coming up at once on one
    GPU) is tried once more on the next port -- the assertion is about the bench line, not about the launcher"""
    last = None
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port + 40 * attempt),
               os.path.join(ROOT, "bench.py")] + bench_args
        last = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env)
        if last.returncode == 0:
            break
    return last


def test_single_rank_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--captures", "24", "--seconds", "1", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    nsamp = d["config"]["samples_per_capture"]
    assert abs(d["value"] - 24 * nsamp * 2 / (d["ms_per_step"] * 2e-3) / 1e6) < 1e-3 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-6
    assert roof["kernel"].startswith("wenet_demod")                       # what the library reports it launched
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["packets_match_gpu"] is True and cb["value"] > 0
    if cb["kind"] == "reference":
        assert {"a_stats100", "b_stats_off", "c_all_cores"} <= set(cb["legs"])          # (+ the half-occupancy leg and the best of the two on a many-core host)
    ow = d["other_workloads"]
    assert {"single_stream", "host_fed", "slipping_100ppm"} <= set(ow)


def test_two_ranks_on_one_gpu_over_gloo():
    env = dict(os.environ, WENET_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _torchrun(2, 29517, ["--gpus", "2", "--captures", "16", "--seconds", "1", "--steps", "2", "--warmup", "1"], env, 900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)                                                   # rank 0 prints the one line
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None and "other_workloads" not in d
    nsamp = d["config"]["samples_per_capture"]
    # value = samples of ALL ranks / max-over-ranks time
    assert abs(d["value"] - 2 * 16 * nsamp * 2 / (d["ms_per_step"] * 2e-3) / 1e6) < 1e-3 * d["value"]
    assert d["packets_valid_per_step_rank0"] > 0


def test_fixed_capture_set_sharded_round_robin_over_two_ranks():
    """BASELINE configs 3 / 5 shape (a fixed set of captures, capture i on rank i mod n_gpus, Eb/N0 sweep 4..12 dB) through the code path the
    8-GPU run will take -- bench.py --total-captures over wenet_amd/shard.py -- here with two ranks sharing the one GPU over gloo."""
    env = dict(os.environ, WENET_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _torchrun(2, 29519, ["--gpus", "2", "--total-captures", "9", "--sweep", "--seconds", "1", "--steps", "2", "--warmup", "1"], env, 900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["captures_per_gpu"] == 5          # rank 0 owns captures 0, 2, 4, 6, 8
    nsamp = d["config"]["samples_per_capture"]
    assert abs(d["value"] - 9 * nsamp * 2 / (d["ms_per_step"] * 2e-3) / 1e6) < 1e-3 * d["value"]
    assert "sweep" in d["config"]["workload"] and d["packets_valid_per_step_rank0"] > 0


def test_config5_shape_eight_ranks_on_one_gpu_over_gloo():
    """BASELINE config 5 as the driver's 8-GPU run will launch it -- 128 channels dealt round-robin to eight ranks, sixteen each -- with the eight
    ranks sharing the one GPU of the test box (gloo): rendezvous, sharding, the barrier + max-over-ranks timing and rank 0's line with n_gpus 8."""
    env = dict(os.environ, WENET_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _torchrun(8, 29523, ["--gpus", "8", "--total-captures", "128", "--seconds", "1", "--steps", "2", "--warmup", "1"], env, 1200)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["captures_per_gpu"] == 16 and d["cpu_baseline"] is None
    nsamp = d["config"]["samples_per_capture"]
    assert abs(d["value"] - 128 * nsamp * 2 / (d["ms_per_step"] * 2e-3) / 1e6) < 1e-3 * d["value"]
    assert d["packets_valid_per_step_rank0"] > 0


def test_two_ranks_default_backend_falls_back_when_rccl_cannot_start():
    """The driver's multi-GPU run takes bench.py's default backend (nccl = RCCL).  Two ranks on ONE GPU is a set-up RCCL refuses (or, where it
    accepts it, runs): either way the run must finish and print its line -- over RCCL, or over gloo with `dist_note` saying so -- with every rank's
    own time and the all-reduced packet count in it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WENET_BENCH_BACKEND", None)
    r = _torchrun(2, 29527, ["--gpus", "2", "--captures", "16", "--seconds", "1", "--steps", "2", "--warmup", "1"], env, 900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and len(d["per_rank_ms"]) == 2 and all(x > 0 for x in d["per_rank_ms"])
    assert abs(max(d["per_rank_ms"]) - d["ms_per_step"]) < 0.05 * d["ms_per_step"]
    assert d["packets_valid_total"] >= 2 * d["packets_valid_per_step_rank0"] - 2          # both ranks decoded (same workload shape, different seeds)
    assert ("gloo" in d["launch"]) == ("dist_note" in d)


def test_single_process_two_handles_on_two_host_threads():
    """`--single-process --gpus 2`: two handles driven by two host threads of one process (one per device; the same device twice on a one-GPU box) --
    the bench-shaped caller of the per-device contexts (tests/test_gpu_units.py::test_two_handles_one_process_two_devices)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "2", "--captures", "16", "--seconds", "1",
                        "--steps", "2", "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and len(d["per_rank_ms"]) == 2 and "single process" in d["launch"] and d["cpu_baseline"] is None
    nsamp = d["config"]["samples_per_capture"]
    assert abs(d["value"] - 2 * 16 * nsamp * 2 / (d["ms_per_step"] * 2e-3) / 1e6) < 1e-3 * d["value"]
    assert d["packets_valid_total"] > d["packets_valid_per_step_rank0"] > 0


def test_live_channels_two_ranks_on_one_gpu_over_gloo():
    """BASELINE config 5 AS WRITTEN (concurrent channels): bench.py --live deals the channels to the ranks round-robin and every rank pushes its
    channels' 100 ms ticks through wenet_rx_push -- here 12 channels over two ranks sharing the one GPU (gloo); the line carries every rank's tick latency and packets,
    every CRC-valid packet is one that was sent on its channel, in order."""
    env = dict(os.environ, WENET_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _torchrun(2, 29531, ["--gpus", "2", "--live", "12", "--seconds", "2", "--warmup", "1"], env, 900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["channels"] == 12 and d["config"]["channels_per_gpu"] == 6 and d["steps"] == 20
    assert len(d["per_rank"]) == 2 and [p["rank"] for p in d["per_rank"]] == [0, 1]
    for p in d["per_rank"]:
        assert p["channels"] == 6 and p["packets_valid"] > 6 * 4 and p["packets_not_as_sent"] == 0 and p["ms_per_tick_mean"] > 0
    nsamp = d["config"]["samples_per_channel"]
    assert abs(d["value"] - 12 * nsamp / (d["ms_per_step"] * 20e-3) / 1e6) < 2e-3 * d["value"]
    assert d["decoder_repeats"] == 0


def test_live_channels_one_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--live", "16", "--seconds", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["config"]["channels_per_gpu"] == 16 and d["per_rank"][0]["packets_not_as_sent"] == 0 and d["per_rank"][0]["packets_valid"] > 16 * 2
