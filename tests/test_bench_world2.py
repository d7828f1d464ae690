"""bench.py as the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N ...), at world size 2 on the ONE
GPU of the test box: both ranks on cuda:0, collectives over gloo (DXMI_DIST_BACKEND / DXMI_DIST_ONE_DEVICE: RCCL refuses two ranks on a
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


@pytest.mark.parametrize("scaling,world", [("weak", 2), ("strong", 2), ("strong", 4)])
def test_bench_two_ranks_on_one_gpu(scaling, world):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DXMI_DIST_BACKEND="gloo", DXMI_DIST_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", "32",
           "--train-steps", "2", "--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    d = json.loads(lines[0])
    per_rank = 32 if scaling == "weak" else 32 // world
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == scaling
    assert d["config"]["images_per_gpu_per_step"] == per_rank and d["config"]["global_batch"] == world * per_rank
    assert d["value"] > 0 and abs(d["value"] - world * per_rank * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert d["train_steps_per_sec"] > 0 and d["rccl"]["world"] == world and len(d["per_rank_images_per_sec"]) == world
    assert d["hip_graph"] is True and not d.get("hip_graph_errors"), d.get("hip_graph_errors")
    assert "roofline" in d and d["roofline"]["kernel"].startswith("conv")


def _torchrun(pkg, script_args, timeout=900):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DXMI_DIST_BACKEND="gloo", DXMI_DIST_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PWD=pkg)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--"] + script_args        # "--": the scripts' own --run must not be read as torchrun's --run-path
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=pkg)


def test_cli_scripts_two_ranks_on_one_gpu():
    """The four drop-in scripts at world size 2 (torchrun, both ranks on cuda:0, gloo): train_cifar10.py splits the global batch
    (reference :298-301), exchanges gradients at the graph cuts of its replayed steps and rank 0 writes the checkpoint; generate_cifar10.py
    shards the images over the ranks (reference :193-204) and meets at its final barrier; the same for train_image_large.py (loss-scale
    bookkeeping and the all-reduce inside MixedPrecisionTrainer.optimize) and generate_large.py.  No rank may issue a collective the
    other does not (a hang is the failure mode: the timeouts are the assertion)."""
    import shutil
    import torch
    pkg = os.path.join(ROOT, "diffusion-by-maxentirl_amd")
    try:
        r = _torchrun(pkg, ["train_cifar10.py", "--config", "builtin:cifar10_T10", "--dataset", "builtin", "--run", "w2", "--synthetic_data",
                            "--max_iters", "3", "--training.batchsize", "8", "--training.n_epochs", "1", "--training.log_every", "1"])
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        logdir = os.path.join(pkg, "results", "cifar10", "cifar10_T10", "w2")
        ck = torch.load(os.path.join(logdir, "sampler_last.pth"), map_location="cpu")
        assert len(ck["state_dict"]) == 330 and ck["iter"] == 3 and all(torch.isfinite(v).all() for v in ck["state_dict"].values())
        r = _torchrun(pkg, ["generate_cifar10.py", "--log_dir", logdir, "-n", "8", "--batchsize", "4", "--epoch", "last", "--skip_fid"])
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert len([f for f in os.listdir(os.path.join(logdir, "generated")) if f.endswith(".png")]) == 8
        over = ["--diffusion.image_size", "32", "--diffusion.num_channels", "64", "--diffusion.num_res_blocks", "1",
                "--diffusion.channel_mult", "1,2", "--diffusion.attention_resolutions", "16", "--sampler.sample_shape", "[3,32,32]",
                "--sampler.n_timesteps", "4", "--trainer.n_timesteps", "4", "--trainer.skip_sampler_tau", "1",
                "--training.batchsize", "8", "--training.log_every", "1", "--data.image_size", "32"]
        r = _torchrun(pkg, ["train_image_large.py", "--config", "builtin:imagenet64_T10", "--dataset", "builtin", "--run", "w2",
                            "--synthetic_data", "--max_iters", "2"] + over)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        logdir = os.path.join(pkg, "results", "imagenet64", "imagenet64_T10", "w2")
        ck = torch.load(os.path.join(logdir, "sampler.pth"), map_location="cpu")
        assert ck["i_iter"] == 1 and all(torch.isfinite(v).all() for v in ck["state_dict"].values())
        r = _torchrun(pkg, ["generate_large.py", "--log_dir", logdir, "--n_sample", "8", "--batchsize", "4", "--skip_fid"])
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert len([f for f in os.listdir(os.path.join(logdir, "generated")) if f.endswith(".png")]) == 8
    finally:
        shutil.rmtree(os.path.join(pkg, "results"), ignore_errors=True)
