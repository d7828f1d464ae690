"""Rank process of tests/test_hip_graph.py::test_graph_cut_runs_collective_between_segments: the nccl (= RCCL) group at
world size = visible GPUs, then the eager-vs-replayed train steps with the gradient exchange at a graph cut."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank = int(os.environ["RANK"])
    backend = os.environ.get("DXMI_TEST_BACKEND", "nccl")
    if backend == "gloo":
        # world-size-2 run on ONE GPU: RCCL refuses two ranks on a device, gloo moves device tensors through the host — the
        # collectives are real (two processes, different data), only the transport differs
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
    import test_hip_graph as tg
    out = tg.cut_check()
    if rank == 0:
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
