"""`python bench.py --gpus 2` as the driver invokes it for N > 1 WITHOUT torch.distributed.run: the script starts its
own ranks, the headline is the strong-scaling configuration (global batch split over the ranks, BASELINE configs[3]),
weak scaling and the ingest-fed step ride along as side objects, and the line carries the collective figures.  The GPU
box has one device, so both ranks share cuda:0 and the collectives go over gloo (PNMN_BENCH_BACKEND / PNMN_BENCH_DEVICE
are test hooks the driver never sets); RCCL itself is exercised by the driver's multi-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_its_own_ranks_and_reports_the_strong_scaling_line():
    env = dict(os.environ, PNMN_BENCH_BACKEND="gloo", PNMN_BENCH_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64",
           "--settle", "2", "--fit-iters", "50", "--fit-target", "0.5", "--ingest-rows", "256", "--roofline-passes", "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["scaling"] == "strong" and line["config"]["global_batch"] == 64 and line["config"]["per_gpu_batch"] == 32
    assert line["config"]["parallelism"] == "dp2" and line["value"] > 0
    assert line["value"] == pytest.approx(64 / (line["ms_per_step"] * 1e-3), rel=1e-3)
    # collectives: the ranks really reduced (world size read back from the process group after all-reduces)
    assert line["rccl_ranks"] == 2 and line["collectives"]["backend"] == "gloo"
    assert line["allreduce_ms_per_step"] > 0 and 0.0 <= line["allreduce_hidden_frac"] <= 1.0
    c = line["collectives"]
    # FC early + two arena pieces + the small-tensor bucket; every trainable parameter travels once
    assert c["collectives_per_step"] == 4 and c["allreduce_bytes_per_step"] > 250e6
    assert 0 < c["cluster_cus"] <= 256 - 32  # data parallel: the recurrent grids leave CUs to the collectives
    # side objects
    weak = line["weak_scaling"]
    assert weak["scaling"] == "weak" and weak["global_batch"] == 128 and weak["value"] > 0
    ing = line["joint_training_ingest"]
    assert ing["value"] > 0 and ing["store_rows"] == 256 and ing["method"] == "resident"
    assert ing["pinned_host"]["value"] > 0 and ing["pinned_host"]["pcie_GBs_per_gpu"] > 0
    roof = line["roofline"]
    assert roof["kernel"] == "conv_nhwc" and roof["passes"] == 4 and len(roof["tflops_per_pass"]) == 4
    assert roof.get("suspect") or (0 < roof["frac"] < 1 and
                                   sum(k["ms_per_step"] for k in roof["kernels"].values()) <= 1.02 * roof["single_stream_step_ms"])
    assert line["cpu_baseline"] is None  # (N = 1 only)
