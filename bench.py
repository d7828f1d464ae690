"""bench.py — DxMI hot-path throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch: a complete T=10 VARSampler generation
(10 DDPM U-Net forwards + 10 fused sampler transitions) of 256 CIFAR-10-shaped images per GPU
(BASELINE.json configs[1]), bf16 MFMA operands / fp32 accumulate / fp32 sampler state, random-init
weights, synthetic Gaussian noise already resident in HBM.  Generation shards by images with no
data-path collective (each rank draws its own trajectories: reference generate_cifar10.py:193-204),
so scaling is "weak" and `value` = all ranks' images / max-over-ranks time.

The JSON line also carries
  roofline     — the dominant kernel (MFMA implicit-GEMM conv, template <1,8,32,6>): algorithmic
                 conv FLOPs per launch / average launch duration, measured with HIP events on the
                 launch stream inside the timed region, against the dense bf16 MFMA peak.
  cpu_baseline — the oracle (torch-CPU fp32 restatement of the reference) timed on this box's host
                 cores on a bounded sample (rank 0, N = 1 only).  Baseline only, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "diffusion-by-maxentirl_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table
UNET_KW = dict(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1,
               in_channels=3, resolution=32)  # reference configs/cifar10/T10.yaml:1-10


def kernel_name(kid):
    """dxmi_conv2d_kernel_id -> the template instantiation name rocprofv3 prints."""
    if kid >= 300000:
        return "conv_stem_kernel"
    if kid >= 200000:
        return "conv1x1_stream_kernel<2, 4, 4>"
    if kid >= 10000:
        nb, ks = (kid // 100) % 100, kid // 10000
        aq = (8 if nb == 2 else ((3 if kid % 100 == 4 else 4) if nb == 4 else 1)) if ks == 3 else 1   # queue depth chosen in conv_pipe.hip
        pmax = kid % 100
        sd = 2 if (ks == 3 and pmax == 4) else 1   # staging distance chosen in conv_pipe.hip
        return f"conv_pipe_kernel<{nb}, {pmax}, {ks}, 0, {aq}, {sd}>"
    return f"conv_igemm_kernel<{kid // 1000}, {(kid // 100) % 10}, 32, {kid % 100}>"


def pmc_traffic(kname):
    """HBM-side bytes per launch of `kname` from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py);
    PMC counters cannot be read from inside the process, so this is the profile of the same command."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            k = json.load(f)["kernels"].get(kname)
        return (k["hbm_bytes_per_launch"] if k else None), "profiles/r01_pmc_traffic.json"
    except OSError:
        return None, "profiles/r01_pmc_traffic.json (missing)"


def build_sampler(device, T):
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler
    torch.manual_seed(0)
    net = Model(**UNET_KW)
    sampler = VARSampler(net, T, [3, 32, 32], trainable_beta="fix_last")
    return sampler.to(device).eval()


def cpu_baseline(T, batch, reps):
    """Oracle (oracle/, torch-CPU fp32) generating `batch` images with T steps, `reps` times."""
    from oracle import unet_small as ounet
    from oracle import var_sampler as ovs
    from oracle import schedule as osched
    from oracle.weights import formula_tensor
    from models.DxMI.unet_small import Model
    shapes = {k: v.shape for k, v in Model(**UNET_KW).state_dict().items()}
    sd = {k: formula_tensor(k, s) for k, s in shapes.items()}
    s = osched.var_schedule(T)
    sched = {k: torch.from_numpy(v) for k, v in s.items() if k != "user_defined_eta"}
    cfg = ounet.UNetSmallConfig()
    g = torch.Generator().manual_seed(0)
    noise = [torch.randn(batch, 3, 32, 32, generator=g) for _ in range(T + 1)]
    best = float("inf")
    with torch.no_grad():
        for _ in range(reps):
            t0 = time.perf_counter()
            ovs.sample(lambda x, t: ounet.forward(sd, cfg, x, t), sched, sched["log_betas"], noise)
            best = min(best, time.perf_counter() - t0)
    return batch / best


def build_trainer(sampler, device, B, T):
    """DxMI trainer on the HIP path with the reference's CIFAR-10 hyper-parameters
    (configs/cifar10/T10.yaml:33-59; optimizer split train_cifar10.py:283-296)."""
    from models.DxMI.trainer import DxMI_Trainer
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    from dxmi_hip.dist import broadcast_parameters
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False,
                                            out_activation="linear", avg_pool_dim=1, learn_out_scale=True, nh=128)).to(device)
    net = sampler.net
    broadcast_parameters(net)
    broadcast_parameters(v)
    not_beta = [p for n, p in net.named_parameters() if "log_betas" not in n]
    opt = torch.optim.Adam([{"params": net.log_betas, "lr": 1e-5}, {"params": not_beta, "lr": 1e-7}])
    opt_v = torch.optim.Adam(v.parameters(), lr=1e-5)
    tr = DxMI_Trainer(batchsize=B, tau1=0.1, tau2=0.01, gamma=1, use_sampler_beta=True, time_cost=0, adavelreg=0.99,
                      entropy_in_value=None, velocity_in_value=None, time_cost_sig=True, n_timesteps=T)
    tr.set_models(f=None, v=v, sampler=sampler, optimizer=opt, optimizer_fstar=None, optimizer_v=opt_v)
    return tr


def train_step(tr, sampler, images, device):
    """One iteration of train_cifar10.py:162-193 (n_critic = n_generator = 1)."""
    from models.DxMI.trainer import append_buffer, reset_buffer
    sampler.eval()
    d_sample = sampler.sample(len(images), device=device)
    buf = append_buffer(reset_buffer(device), d_sample)
    d_energy = tr.update_f_v(images, d_sample, buf)
    d_sampler = tr.update_sampler(buf, 1)
    return d_energy, d_sampler


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--T", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-conv-events", action="store_true", help="skip the per-launch HIP events (roofline leg)")
    ap.add_argument("--train-steps", type=int, default=3, help="timed DxMI train steps (0 = skip the train leg)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")  # RCCL
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from dxmi_hip import ops
    ops.device_check()
    sampler = build_sampler(device, args.T)
    B, T = args.batch, args.T
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    # synthetic inputs resident in HBM before the timed region: x_T and one z per step
    noise = [torch.randn(B, 3, 32, 32, device=device, generator=g) for _ in range(T + 1)]

    def step():
        return sampler.sample(B, device=device, noise=noise)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # Roofline leg: HIP events bracket every conv launch (on the launch stream) during the FIRST of the timed steps
    # only: an event pair costs ~2 us of stream time and a step has ~3500 conv launches, so bracketing all K steps
    # would take ~10 % off the number being measured.  220 launches of the dominant kernel are averaged.
    prof = None if (args.no_conv_events or rank != 0) else ops.ConvProfiler()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.CONV_PROFILER = prof if i == 0 else None
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    ops.CONV_PROFILER = None
    assert torch.isfinite(out["sample"]).all()
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()

    # ---- second leg: DxMI train step (sample + value/energy update + policy update), same batch/GPU
    train_sps = None
    if args.train_steps > 0:
        tr = build_trainer(sampler, device, B, T)
        gimg = torch.Generator(device=device).manual_seed(112233 + rank)
        imgs = torch.rand(B, 3, 32, 32, device=device, generator=gimg) * 2 - 1
        train_step(tr, sampler, imgs, device)          # warm-up (weight packing, workspaces)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.train_steps):
            logs = train_step(tr, sampler, imgs, device)
        sync_all()
        t_train = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([t_train], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            t_train = tt.item()
        assert all(v == v for v in logs[0].values())   # no NaN
        train_sps = args.train_steps / t_train

    if rank != 0:
        return
    images = B * args.steps * world
    line = {
        "metric": "images/sec (CIFAR-10 DDPM T=10 generation)", "value": images / elapsed, "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"CIFAR-10 DDPM U-Net (35.7M params) VARSampler T={T} generation, "
                               f"{B} images/GPU/step, 3x32x32 (BASELINE configs[1])",
                   "images_per_gpu_per_step": B, "T": T, "parallelism": f"dp{world} (independent trajectories, no collective)"},
        "train_steps_per_sec": train_sps,
        "train_config": {"per_gpu_batch": B, "global_batch": B * world, "timed_steps": args.train_steps,
                         "step": "sample T + update_f_v (1 energy + T TD steps) + update_sampler, Adam, dropout 0.1",
                         "grad_sync": "flat fp32 all-reduce over RCCL" if world > 1 else "none (1 GPU)"},
    }
    if prof is not None:
        summ = prof.summary()
        if summ:
            kid, s = max(summ.items(), key=lambda kv: kv[1]["ms"])
            tflops = s["flops"] / (s["ms"] * 1e-3) / 1e12
            traffic, traffic_src = pmc_traffic(kernel_name(kid))
            line["roofline"] = {
                "bound": "mfma", "achieved": tflops, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
                "traffic_unit": "HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE), PMC passes in " + traffic_src,
                "algorithmic_bytes_per_launch": s["bytes"] / s["launches"],
                "kernel": kernel_name(kid),
                "launches": s["launches"], "avg_launch_us": 1e3 * s["ms"] / s["launches"],
                "algorithmic_gflop_per_launch": s["flops"] / s["launches"] / 1e9,
                "algorithmic_gbps": s["bytes"] / (s["ms"] * 1e-3) / 1e9,
                "share_of_step_time": s["ms"] * 1e-3 / (elapsed / args.steps),
                "events": "first timed step only",
            }
            line["conv_kernels"] = {kernel_name(k): {"launches": v["launches"], "ms": round(v["ms"], 3),
                                             "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12} for k, v in summ.items()}
    if world == 1 and not args.no_cpu_baseline:
        cb = 16
        v = cpu_baseline(T, cb, reps=2)
        line["cpu_baseline"] = {"value": v, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                                "sample": f"oracle (torch-CPU fp32 restatement), {cb} images x T={T}, best of 2 "
                                          f"(BASELINE configs[0] shape); host has {os.cpu_count()} logical cores"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
