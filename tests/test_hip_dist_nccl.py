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


def test_trainer_step_under_rccl():
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
    line = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["world"] == world
    assert line["rank_identical_parameters"] and line["finite"]
    assert line["flat_bytes"][0] == 5_134_595 * 4 and line["flat_bytes"][1] == (35_746_307 + 4) * 4
