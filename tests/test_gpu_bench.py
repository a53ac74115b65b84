"""bench.py's contract on a real MI355X: the N = 1 JSON line (metric / roofline / cpu_baseline / parity fields) and the N = 2 code
path the driver launches on a whole node - rehearsed here with two ranks sharing GPU 0 over gloo (OFX_BENCH_REHEARSAL=1: its
throughput means nothing, its control flow - rendezvous, barrier, MAX-over-ranks timing, rank-0 JSON - is the real one)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line_has_the_contract_fields():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--outfits", "64", "--cpu-outfits", "2",
                        "--cpu-cfg2-outfits", "1", "--secondary", "", "--rung", ""], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["unit"] == "outfits/s" and j["scaling"] == "weak"
    assert abs(j["value"] - 64 * 3 / (j["ms_per_step"] * 3e-3)) <= 1e-3 * j["value"]
    assert j["parity_rel_err_vs_oracle"] < 1e-3                      # the headline mode is the 1e-3-compliant one
    assert j["config"]["launch"].startswith("one HIP graph launch per step")      # the default: steps replayed from the captured graph, logits checked equal
    lbl = j["launch_by_launch"]                                       # the same call with graph replay off, reported beside the headline
    assert lbl["bit_identical_to_replayed"] is True and lbl["ms_per_step"] > 0 and lbl["steps"] == 10
    sp = j["step_ms_spread"]
    assert len(sp["all"]) == 3 and len(sp["host_issue_ms"]) == 3 and sp["device_allocs_in_timed_region"] <= 2
    rf, cb = j["roofline"], j["cpu_baseline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["per_shape"]
    assert sum(t["launches"] for t in rf["per_shape"]) == rf["launches_per_step"]
    assert cb["kind"] == "port" and cb["threads"] >= 1 and cb["iters"] >= 5 and cb["median_ms"] > 0
    c1 = cb["cfg1_cp_forward_32_outfits"]
    assert c1["all_threads"]["iters"] >= 10 and c1["one_thread"]["iters"] >= 10 and c1["one_thread"]["outfits_per_s"] > 0


def test_bench_two_ranks_rehearsal():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OFX_BENCH_REHEARSAL="1")
    env.pop("OMP_NUM_THREADS", None)                               # bench.py caps its host threads per rank itself (host_cores() // world)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--outfits", "32",
                        "--cir-queries", "200", "--cir-pool", "20000"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and "cpu_baseline" not in j
    # whole-job aggregate: both ranks' outfits over the MAX-over-ranks time
    assert abs(j["value"] - 2 * 32 * 2 / (j["ms_per_step"] * 2e-3)) <= 1e-3 * j["value"]
    assert j["config"]["parallelism"].startswith("dp2")
    rk = j["ranks"]
    assert rk["batch_seeds"] == [1236, 1237]                       # rank-local batches (weak scaling: each rank scores its own outfits)
    assert len(rk["elapsed_s"]) == 2 and abs(max(rk["elapsed_s"]) - j["ms_per_step"] * 2e-3) <= 2e-3 * max(rk["elapsed_s"])     # MAX over ranks is the job's time
    assert all(t >= 1 for t in rk["host_threads"])
    assert rk["graph_launch"] == [True, True]                      # every rank replays its own captured step
    # the one data-path collective the north star names (CIR: pool row-sharded, ONE all-gather of the per-shard candidate lists) ran across the two
    # ranks and reproduced rank 0's unsharded top-k exactly
    assert rk["cir_equal"] is True and rk["cir_allgather_ms"] > 0 and rk["cir_local_topk_ms_rank0"] > 0
