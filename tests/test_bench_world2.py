"""bench.py as the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N ...), at world size 2 on the ONE
GPU of the test box: both ranks on cuda:0, collectives over gloo (DXMI_BENCH_BACKEND / DXMI_BENCH_ONE_DEVICE: RCCL refuses two ranks on a
device).  What it guards: every rank walks the same sequence of collectives through the warm-up, the timed region, the hipGraph capture
with its cuts and the per-launch-event step (which must run on EVERY rank: its gradient exchanges are collectives), and rank 0 prints
one JSON line with the contract's keys, for weak and for strong scaling."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_on_one_gpu(scaling):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DXMI_BENCH_BACKEND="gloo", DXMI_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32",
           "--train-steps", "2", "--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    d = json.loads(lines[0])
    per_rank = 32 if scaling == "weak" else 16
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == scaling
    assert d["config"]["images_per_gpu_per_step"] == per_rank and d["config"]["global_batch"] == 2 * per_rank
    assert d["value"] > 0 and abs(d["value"] - 2 * per_rank * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert d["train_steps_per_sec"] > 0 and d["rccl"]["world"] == 2 and len(d["per_rank_images_per_sec"]) == 2
    assert d["hip_graph"] is True and not d.get("hip_graph_errors"), d.get("hip_graph_errors")
    assert "roofline" in d and d["roofline"]["kernel"].startswith("conv")
