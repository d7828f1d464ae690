"""bench.py — DxMI hot-path throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: launched once per rank by torchrun / torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment), or as a plain `python bench.py --gpus N` — the parent then spawns N fresh rank processes BEFORE it touches
the GPU (no exec, no fork of an initialised HIP context) and relays rank 0's JSON line.

One "step" = one pass of the hot path over one batch: a complete T=10 VARSampler generation (10 DDPM U-Net forwards +
10 fused sampler transitions, Gaussian draws included) of 256 CIFAR-10-shaped images per GPU (BASELINE.json configs[1]),
bf16 MFMA operands / fp32 accumulate / fp32 sampler state, random-init weights.  Generation shards by images with no
data-path collective (each rank draws its own trajectories: reference generate_cifar10.py:193-204), so scaling is "weak"
and `value` = all ranks' images / max-over-ranks time of exactly K steps between barrier + synchronize brackets.

The JSON line also carries
  roofline           the dominant kernel (3x3 MFMA implicit-GEMM conv): algorithmic FLOPs per launch / average launch
                     duration from HIP events on the launch stream, against the dense bf16 MFMA peak (2.5 PFLOP/s);
  roofline_classes   the same per kernel class — conv3x3 / conv1x1 (MFMA-graded), GroupNorm / attention / sampler step
                     (HBM-graded, algorithmic bytes / 8 TB/s), and for the train leg wgrad (MFMA), GroupNorm backward,
                     optimiser, replay gather (HBM); `traffic` = HBM-side bytes per launch of the class's main kernel and
                     `conv_kernels[*].traffic` per conv kernel, from the committed rocprofv3 --pmc passes (profiles/);
  edm                BASELINE configs[3] / [4] generation (ImageNet-64 T=10, LSUN-256 T=4) and the EDM train step;
  train_steps_per_sec  second leg: full DxMI train step (sample T + update_f_v + update_sampler) at the same batch;
  reference_eager_gpu  the reference's op sequence (the oracle's eager restatement: NCHW, unfused torch ops on
                     MIOpen / rocBLAS) on the same GPU, fp32 and bf16-autocast — the ">= 3x" comparison, measured here;
  cpu_baseline       the oracle (torch-CPU fp32) on this box's host cores, bounded sample, at 8 threads and at all
                     cores (rank 0, N = 1 only).  Baseline only, not the target.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "diffusion-by-maxentirl_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table
HBM_PEAK_GBPS = 8000.0                # same table (spec; ~6.3 TB/s achievable)
UNET_KW = dict(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1,
               in_channels=3, resolution=32)  # reference configs/cifar10/T10.yaml:1-10
PMC_FILE = "profiles/r06_pmc_traffic.json"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--T", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-reference", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the per-launch HIP events (roofline legs)")
    ap.add_argument("--train-steps", type=int, default=8, help="timed DxMI train steps (0 = skip the train leg)")
    ap.add_argument("--no-edm", action="store_true", help="skip the EDM legs (BASELINE configs[3], configs[4] at their per-GPU sizes)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch images per GPU whatever N (generation shards by images: generate_cifar10.py:193-204); strong: the "
                         "reference's TRAINING semantics, per-rank batch = --batch // N (train_cifar10.py:298-301: `batchsize // ngpus`)")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch from python (no hipGraph replay: dxmi_hip/graph.py); same as DXMI_GRAPH=0")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the 1-GPU legs at the per-rank batches of an 8-GPU / 4-GPU reference run (32 @ T=10, 128 @ T=4)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------- self-launch
def spawn_ranks(n):
    """Parent of a plain `python bench.py --gpus N`: start N rank processes (fresh interpreters, nothing GPU-related has
    run in this one), wait, print rank 0's line, exit with the worst return code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DXMI_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


# ----------------------------------------------------------------------------------------------- pieces
def kernel_name(kid):
    """dxmi_conv2d_kernel_id -> the template instantiation name rocprofv3 prints."""
    if kid >= 600000:
        return "conv_head_kernel"
    if kid >= 550000:
        return f"conv1x1_rw8_kernel<{(kid // 10) % 100}, {'true' if kid % 10 else 'false'}>"
    if kid >= 500000:
        return f"conv1x1_rw_kernel<{(kid // 1000) % 10}, {(kid // 10) % 100}, {'true' if kid % 10 else 'false'}>"
    if kid in (400008, 400009):      # 400009: OpProfiler's tag for a launch with the fused GroupNorm output (gn_out)
        return "conv_ws8_kernel<true>" if kid == 400009 else "conv_ws8_kernel<false>"
    if kid >= 450000:
        return f"conv_sm_kernel<{2 if (kid - 450000) // 100 == 4 else 3}, 8, {kid % 100}>"
    if kid >= 400000:
        return f"conv_ws_kernel<{kid - 400000}>"
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
    """HBM-side bytes per launch of `kname` from the committed rocprofv3 --pmc passes of this same command
    (tools/pmc_traffic.py); counters cannot be read in-process.  None when the profile does not hold this kernel name
    (i.e. the kernel changed after the passes were taken)."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            k = json.load(f)["kernels"].get(kname)
        if k is None:       # kernels of the train leg only (weight gradient, GroupNorm backward): the train leg's passes
            with open(os.path.join(ROOT, PMC_FILE.replace("pmc_traffic", "pmc_train_traffic"))) as f:
                k = json.load(f)["kernels"].get(kname)
        return (k["hbm_bytes_per_launch"] if k else None), PMC_FILE
    except (OSError, ValueError, KeyError):
        return None, PMC_FILE + " (missing)"


def build_sampler(device, T):
    import torch
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler
    torch.manual_seed(0)
    net = Model(**UNET_KW)
    sampler = VARSampler(net, T, [3, 32, 32], trainable_beta="fix_last")
    return sampler.to(device).eval()


def oracle_sampler(device, T):
    """The oracle's restatement of the reference's sampler on `device` (cpu, or cuda for the eager-reference leg)."""
    import torch
    from oracle import schedule as osched
    from oracle import unet_small as ounet
    from oracle import var_sampler as ovs
    from oracle.weights import formula_tensor
    from models.DxMI.unet_small import Model
    shapes = {k: v.shape for k, v in Model(**UNET_KW).state_dict().items()}
    sd = {k: formula_tensor(k, s).to(device) for k, s in shapes.items()}
    s = osched.var_schedule(T)
    sched = {k: torch.from_numpy(v).to(device) for k, v in s.items() if k != "user_defined_eta"}
    cfg = ounet.UNetSmallConfig()
    return lambda noise: ovs.sample(lambda x, t: ounet.forward(sd, cfg, x, t).float(), sched, sched["log_betas"], noise)


def usable_cores():
    """Host cores this process may actually use: scheduler affinity, capped by the cgroup CPU quota when one is set
    (a 256-thread pool on a quota of a few cores measures oversubscription, not the CPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(T, batch, threads, steps=None):
    """Oracle (oracle/, torch-CPU fp32) generating `batch` images on `threads` host threads: `steps` of the T sampler
    steps are run (every step costs the same: one U-Net forward + the transition) and the rate is scaled to T steps."""
    import torch
    steps = T if steps is None else steps
    torch.set_num_threads(threads)
    fn = oracle_sampler("cpu", steps)
    g = torch.Generator().manual_seed(0)
    noise = [torch.randn(batch, 3, 32, 32, generator=g) for _ in range(steps + 1)]
    with torch.no_grad():
        t0 = time.perf_counter()
        fn(noise)
        dt = time.perf_counter() - t0
    return batch / (dt * T / steps)


def eager_reference_gpu(device, T, batch):
    """The reference's op sequence run eagerly by torch on this GPU (the oracle restatement: same NCHW / fp32 / unfused
    torch.nn.functional ops the reference executes through MIOpen and rocBLAS), fp32 and under bf16 autocast."""
    import torch
    out = {}
    fn = oracle_sampler(device, T)       # formula weights are generated on the CPU, then moved
    prev = torch.get_default_device()
    torch.set_default_device(device)     # the oracle builds its small helper tensors on the default device
    try:
        g = torch.Generator(device=device).manual_seed(1)
        noise = [torch.randn(batch, 3, 32, 32, device=device, generator=g) for _ in range(T + 1)]
        for name, ctx in (("fp32", torch.autocast("cuda", enabled=False)), ("bf16_autocast", torch.autocast("cuda", dtype=torch.bfloat16))):
            with torch.no_grad(), ctx:
                fn(noise)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 2
                for _ in range(reps):
                    fn(noise)
                torch.cuda.synchronize()
                out[name] = batch * reps / (time.perf_counter() - t0)
    finally:
        torch.set_default_device(prev)
    return out


def build_trainer(sampler, device, B, T, value_resample=False):
    """DxMI trainer on the HIP path with the reference's CIFAR-10 hyper-parameters
    (configs/cifar10/T10.yaml:33-59; optimizer split train_cifar10.py:283-296)."""
    from dxmi_hip.dist import broadcast_parameters
    from dxmi_hip.optim import Adam
    from models.DxMI.trainer import DxMI_Trainer
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False,
                                            out_activation="linear", avg_pool_dim=1, learn_out_scale=True, nh=128)).to(device)
    net = sampler.net
    broadcast_parameters(net)
    broadcast_parameters(v)
    not_beta = [p for n, p in net.named_parameters() if "log_betas" not in n]
    opt = Adam([{"params": net.log_betas, "lr": 1e-5}, {"params": not_beta, "lr": 1e-7}])
    opt_v = Adam(v.parameters(), lr=1e-5)
    tr = DxMI_Trainer(batchsize=B, tau1=0.1, tau2=0.01, gamma=1, use_sampler_beta=True, time_cost=0, adavelreg=0.99,
                      entropy_in_value=None, velocity_in_value=None, time_cost_sig=True, n_timesteps=T, value_resample=value_resample)
    tr.set_models(f=None, v=v, sampler=sampler, optimizer=opt, optimizer_fstar=None, optimizer_v=opt_v)
    return tr


def train_step(tr, sampler, images, device, ring):
    """One iteration of train_cifar10.py:162-193 (n_critic = n_generator = 1)."""
    from models.DxMI.trainer import append_buffer, reset_buffer
    sampler.eval()
    d_sample = sampler.sample(len(images), device=device, out=ring.next_slot())
    buf = append_buffer(ring, d_sample)
    d_energy = tr.update_f_v(images, d_sample, buf)
    d_sampler = tr.update_sampler(buf, 1)
    reset_buffer(device, ring=ring)
    return d_energy, d_sampler


def edm_generation_leg(device, name, B, reps=2, events=True):
    """BASELINE configs[3] / configs[4] at their per-GPU batch (generate_large.py:22-44: `--batchsize` is per rank): the
    reference's EDM / ADM U-Net (models/cm/unet.py) under OpenAIDiffusion.sample (models/DxMI/openai_diffusion.py:101) on the HIP
    path, random-init weights of the named architecture (zero-initialised layers get weights so no work is skipped),
    x_T and the per-step draws inside the timed call.  Outside the headline timed region; one warm-up call + `reps` timed."""
    import torch
    import configs_builtin
    from dxmi_hip import ops
    from models.cm.script_util import create_model_and_diffusion
    from models.DxMI.openai_diffusion import OpenAIDiffusion
    cfg = configs_builtin.get(name)
    torch.manual_seed(0)
    with torch.device(device):            # parameters are created (and initialised) on the GPU: 0.3 / 0.5 G parameters
        net, diffusion = create_model_and_diffusion(**cfg.diffusion)
    with torch.no_grad():
        for p in net.parameters():
            if float(p.abs().max()) == 0:
                torch.nn.init.normal_(p, std=0.02)
    s = OpenAIDiffusion(net, diffusion, **cfg.sampler)
    net.to(device).eval()
    T = s.n_timesteps
    out = {"workload": f"{name}: EDM U-Net ({sum(p.numel() for p in net.parameters()) / 1e6:.1f}M params) OpenAIDiffusion T={T}, "
                       f"{B} images/GPU/call, 3x{cfg.diffusion.image_size}x{cfg.diffusion.image_size}, bf16 torso", "T": T, "batch": B}
    with torch.no_grad():
        d = s.sample(B, device=device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            d = s.sample(B, device=device)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        assert torch.isfinite(d["sample"]).all()
        out.update({"images_per_sec": round(B / dt, 1), "ms_per_call": round(dt * 1e3, 1), "timed_calls": reps})
        if events:
            prof = ops.OpProfiler()
            ops.PROFILER = prof
            s.sample(B, device=device)
            torch.cuda.synchronize()
            ops.PROFILER = None
            summ = prof.summary()
            out["roofline_classes"] = class_rooflines(summ, dt)
            convs = {k: v for k, v in summ.items() if k[0].startswith("conv")}
            (cls, kid), c = max(convs.items(), key=lambda kv: kv[1]["ms"])
            tf = c["flops"] / (c["ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": kernel_name(kid), "achieved": round(tf, 1), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "launches_per_call": c["launches"],
                               "avg_launch_us": round(1e3 * c["ms"] / c["launches"], 1), "share_of_call_time": round(c["ms"] * 1e-3 / dt, 4)}
            out["conv_kernels"] = {kernel_name(k[1]): {"launches": v["launches"], "ms": round(v["ms"], 2),
                                                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} for k, v in convs.items()}
    del s, net, d
    torch.cuda.empty_cache()
    return out


def edm_train_leg(device, name="imagenet64_T10", B=16, steps=4, graph=True):
    """One DxMI_Trainer_Cond step (models/DxMI/trainer.py:693-746 through MixedPrecisionTrainer, models/cm/fp16_util.py) on the
    ImageNet-64 EDM net at per-GPU batch `B` (synthetic images / labels): sample T + update_f_v + update_sampler_mixed_precision."""
    import torch
    import configs_builtin
    import dxmi_config
    from dxmi_hip.optim import Adam, RAdam
    from models.cm.fp16_util import MixedPrecisionTrainer
    from models.cm.script_util import create_model_and_diffusion
    from models.DxMI.openai_diffusion import OpenAIDiffusion
    from models.DxMI.replay import TransitionRing
    from models.DxMI.trainer import append_buffer, reset_buffer
    cfg = configs_builtin.get(name)
    torch.manual_seed(0)
    with torch.device(device):
        unet, diffusion = create_model_and_diffusion(**cfg.diffusion)
    with torch.no_grad():
        for p in unet.parameters():
            if float(p.abs().max()) == 0:
                torch.nn.init.normal_(p, std=0.02)
    sampler = OpenAIDiffusion(unet, diffusion, **cfg.sampler)
    unet.to(device)
    v = dxmi_config.instantiate(cfg.value).to(device)
    mp = MixedPrecisionTrainer(model=unet, use_fp16=True, initial_lg_loss_scale=20, special_key="log_betas")
    opt = RAdam([{"params": mp.master_params[1:], "lr": 1e-8}, {"params": mp.master_params[0:1], "lr": 1e-6}])
    opt_v = Adam(v.parameters(), lr=1e-5)
    trainer = dxmi_config.instantiate(cfg.trainer, batchsize=B)
    trainer.set_models(v=v, sampler=sampler, optimizer=opt, optimizer_v=opt_v)
    trainer.use_graphs = sampler.use_graph = graph      # the three phases of the iteration replay as hipGraphs (dxmi_hip/graph.py)
    res = cfg.diffusion.image_size
    g = torch.Generator(device=device).manual_seed(1)
    # the replay ring of train_image_large.py (one trajectory per iteration, generated in place by the sampler)
    ring = TransitionRing(1, trainer.n_timesteps, B, sampler.sample_shape, device, with_y=True, sigma_dims=1)

    def step():
        data = torch.rand(B, 3, res, res, device=device, generator=g) * 2 - 1
        y = torch.randint(0, 1000, (B,), device=device, generator=g)
        sampler.eval()
        d = sampler.sample(B, device=device, i_class=y, out=ring.next_slot())
        buf = append_buffer(ring, d)
        le = trainer.update_f_v(data, d, buf, y=y)
        ls = trainer.update_sampler_mixed_precision(buf, mp_trainer=mp)
        reset_buffer(device, ring=ring)
        return le, ls

    from dxmi_hip import ops
    ops.tune_for_throughput(True)              # train_image_large.py's setting: under-filled conv grids on smaller tiles
    for _ in range(3 if graph else 1):         # (graph: eager first call, capture, first replay)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        le, ls = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ops.tune_for_throughput(False)
    assert all(x == x for x in le.values())
    out = {"workload": f"{name}: DxMI_Trainer_Cond step, per-GPU batch {B}, T={sampler.n_timesteps}, fused RAdam / Adam, loss scale 2^20",
           "train_steps_per_sec": round(1 / dt, 3), "ms_per_step": round(dt * 1e3, 1), "timed_steps": steps, "hip_graph": graph,
           "lg_loss_scale_after": round(float(mp.lg_loss_scale), 3), "radam_step_count": int(opt.step_count())}
    del trainer, mp, opt, opt_v, sampler, unet, v
    torch.cuda.empty_cache()
    return out


def small_batch_leg(device, B, T, graph, value_resample=False, steps=6):
    """The DxMI train step and the generation call at a per-rank batch of the reference's multi-GPU runs (`batchsize // ngpus`,
    train_cifar10.py:298-301): B = 32 is configs[1] (global 256, T = 10) on 8 GPUs, B = 128 / T = 4 / value_resample is configs[2]
    (global 512, T4_ddgan.yaml) on 4.  Fresh nets; 3 warm-up steps (the second one is the capture when `graph`), `steps` timed.
    host_issue_ms = host time until the python calls of a step have returned (for the train step this includes its two log
    read-backs); gpu_ms = HIP-event time of the same steps on the launch stream."""
    import torch
    from dxmi_hip import ops
    from models.DxMI.replay import TransitionRing
    sampler = build_sampler(device, T)
    sampler.use_graph = graph
    out = {"per_gpu_batch": B, "T": T, "hip_graph": graph, "timed_steps": steps}

    def timed(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        issue = 0.0
        for _ in range(steps):
            ti = time.perf_counter()
            fn()
            issue += time.perf_counter() - ti
        e1.record()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, issue / steps, e0.elapsed_time(e1) / steps

    for _ in range(3):
        sampler.sample(B, device=device)
    wall, issue, gpu = timed(lambda: sampler.sample(B, device=device))
    out["gen"] = {"images_per_sec": round(B / wall, 1), "ms_per_call": round(1e3 * wall, 2), "host_issue_ms": round(1e3 * issue, 2), "gpu_ms": round(gpu, 2)}
    ops.tune_for_throughput(True)
    tr = build_trainer(sampler, device, B, T, value_resample=value_resample)
    tr.use_graphs = graph
    ring = TransitionRing(1, T, B, (3, 32, 32), device)
    imgs = torch.rand(B, 3, 32, 32, device=device) * 2 - 1
    for _ in range(3):
        train_step(tr, sampler, imgs, device, ring)
    wall, issue, gpu = timed(lambda: train_step(tr, sampler, imgs, device, ring))
    ops.tune_for_throughput(False)
    out["train"] = {"steps_per_sec": round(1 / wall, 2), "ms_per_step": round(1e3 * wall, 2), "host_issue_ms": round(1e3 * issue, 2), "gpu_ms": round(gpu, 2)}
    del tr, sampler, ring
    torch.cuda.empty_cache()
    return out


def class_rooflines(summ, step_seconds):
    """Per kernel class: MFMA-graded (conv*, attention also reported against MFMA, wgrad) or HBM-graded."""
    by_cls = {}
    for (cls, name), s in summ.items():
        c = by_cls.setdefault(cls, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        for k in c:
            c[k] += s[k]
    out = {}
    for cls, c in by_cls.items():
        sec = c["ms"] * 1e-3
        if sec <= 0:
            continue
        # conv_other = conv_out (128 -> 3 channels): one read of the activation per 3 outputs, AI ~26 FLOP/B -> HBM-graded
        # attention: graded by its own arithmetic intensity against the ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B): the CIFAR net's
        # single 256-token head (AI ~128) is HBM-bound, the ADM nets' 1024-token blocks (AI = T / 2 = 512) are MFMA-bound
        ridge = MFMA_BF16_DENSE_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBPS * 1e9)
        mfma = cls in ("conv3x3", "conv1x1", "wgrad") or (cls == "attention" and c["bytes"] > 0 and c["flops"] / c["bytes"] > ridge)
        e = {"bound": "mfma" if mfma else "hbm", "launches": c["launches"], "ms_per_step": round(c["ms"], 3),
             "share_of_step_time": round(sec / step_seconds, 4),
             "algorithmic_tflops": round(c["flops"] / sec / 1e12, 1), "algorithmic_gbps": round(c["bytes"] / sec / 1e9, 1)}
        e["frac"] = round(c["flops"] / sec / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS if mfma else c["bytes"] / sec / 1e9 / HBM_PEAK_GBPS, 4)
        if cls in ("conv1x1", "attention", "conv_other"):      # both rooflines shown
            e["frac_hbm"] = round(c["bytes"] / sec / 1e9 / HBM_PEAK_GBPS, 4)
            e["frac_mfma"] = round(c["flops"] / sec / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)
        # HBM-side bytes per launch of the class's main kernel from the committed PMC passes (null when the profile lacks it)
        main_kernel = CLASS_MAIN_KERNEL.get(cls)
        if main_kernel:
            e["traffic"] = {"kernel": main_kernel, "hbm_bytes_per_launch": pmc_traffic(main_kernel)[0]}
        out[cls] = e
    return out


# the kernel that carries most of a non-conv class's bytes (conv classes: see `conv_kernels`)
CLASS_MAIN_KERNEL = {"groupnorm": "gn_apply_kernel<4>", "attention": "attention256_kernel<true>", "wgrad": "conv_wgrad_ws_kernel<3>",
                     "groupnorm_bwd": "gn_silu_bwd_kernel<4, 16>", "conv_stem": "conv_stem_kernel", "conv_other": "conv_head_kernel"}


# ----------------------------------------------------------------------------------------------- main (one rank)
def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1 and "DXMI_BENCH_CHILD" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    from dxmi_hip import dist as hip_dist
    device = torch.device(hip_dist.rank_device(local_rank))     # cuda:<LOCAL_RANK>; DXMI_DIST_ONE_DEVICE=1 (tests): every rank on cuda:0
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = hip_dist.dist_backend()      # "nccl" IS RCCL; DXMI_DIST_BACKEND=gloo only for the world-2-on-one-GPU test of this file
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from dxmi_hip import graph as hip_graph
    from dxmi_hip import ops
    ops.device_check()
    sampler = build_sampler(device, args.T)
    use_graph = hip_graph.default_enabled() and not args.no_graph
    sampler.use_graph = use_graph          # the T-step loop of a fixed (batch, destination) replays as one hipGraph after its first call
    B, T = (args.batch // world if args.scaling == "strong" else args.batch), args.T
    torch.manual_seed(1234 + rank)      # seed + rank, as generate_cifar10.py:103-110
    torch.cuda.manual_seed(1234 + rank)

    def step():
        # x_T and the T per-step draws are generated inside the step, as the reference does (var_sampler.py:242, :285)
        return sampler.sample(B, device=device)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            tt = torch.tensor([x], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            return tt.item()
        return x

    graph_errors = {}
    try:
        for _ in range(args.warmup + (2 if use_graph else 0)):     # (+ the eager first call and the capture call of the hipGraph: never timed)
            step()
    except Exception as e:          # a capture that fails on this box must not cost the measurement: issue the launches from python
        if not use_graph:
            raise
        graph_errors["generation"] = repr(e)[:300]
        use_graph = sampler.use_graph = False
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
    # ---- timed region: exactly K steps between barrier + synchronize brackets; one event per step for the median
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sync_all()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = step()
        marks[i + 1].record()
    sync_all()
    my_elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(my_elapsed)
    rank_elapsed = [my_elapsed]
    if world > 1:       # every rank's own wall time over the same K steps (a scaling record should show which rank was slow)
        tt = torch.zeros(world, device=device, dtype=torch.float64)
        tt[rank] = my_elapsed
        torch.distributed.all_reduce(tt)
        rank_elapsed = tt.tolist()
    per_step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    assert torch.isfinite(out["sample"]).all()

    # ---- roofline leg (rank 0, outside the timed region): HIP events bracket every kernel launch of two more steps.
    # An event pair costs ~2 us of stream time and a step has ~2500 launches, so bracketing inside the timed region
    # would take ~8 % off the number being measured; the bracketed launches are the same kernels on the same shapes.
    gen_summ, gen_clock = None, None
    if rank == 0 and not args.no_events:
        prof = ops.OpProfiler()
        ops.PROFILER = prof
        sampler.use_graph = False             # per-launch events need the launches issued one by one
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ops.PROFILER = None
        sampler.use_graph = use_graph
        gen_summ = {k: {kk: vv / 2 for kk, vv in v.items()} for k, v in prof.summary().items()}     # per step
        gen_clock = ops.conv_ws_clock_ghz()        # shader clock during the last conv_ws_kernel launch of the generation step

    # ---- second leg: DxMI train step (sample + value/energy update + policy update), same batch/GPU
    train_sps, train_summ, t_train_step = None, None, None
    if args.train_steps > 0:
        from models.DxMI.replay import TransitionRing
        ops.tune_for_throughput(True)          # as train_cifar10.py / train_image_large.py do (the generation legs keep the defaults)
        tr = build_trainer(sampler, device, B, T)
        tr.use_graphs = use_graph             # update_f_v / update_sampler replay as hipGraphs (gradient exchanges at graph cuts)
        ring = TransitionRing(1, T, B, (3, 32, 32), device)
        gimg = torch.Generator(device=device).manual_seed(112233 + rank)
        imgs = torch.rand(B, 3, 32, 32, device=device, generator=gimg) * 2 - 1
        try:
            for _ in range(3):                               # warm-up (weight packing, workspaces, optimiser state, allocator growth; the second step is the graph capture)
                train_step(tr, sampler, imgs, device, ring)
        except Exception as e:
            if not use_graph:
                raise
            graph_errors["train"] = repr(e)[:300]
            use_graph = sampler.use_graph = tr.use_graphs = False
            torch.cuda.synchronize()
            ring.reset()
            for _ in range(2):
                train_step(tr, sampler, imgs, device, ring)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.train_steps):
            logs = train_step(tr, sampler, imgs, device, ring)
        sync_all()
        t_train = max_over_ranks(time.perf_counter() - t1)
        assert all(v == v for v in logs[0].values())   # no NaN
        train_sps = args.train_steps / t_train
        t_train_step = t_train / args.train_steps
        if not args.no_events:
            # EVERY rank runs this step (its gradient exchanges are collectives: a step on rank 0 alone would wait for ranks that are
            # already at the next barrier); the per-launch events are taken on rank 0 only
            prof = ops.OpProfiler() if rank == 0 else None
            ops.PROFILER = prof
            sampler.use_graph = tr.use_graphs = False
            train_step(tr, sampler, imgs, device, ring)
            torch.cuda.synchronize()
            ops.PROFILER = None
            sampler.use_graph = tr.use_graphs = use_graph
            train_summ = prof.summary() if prof is not None else None
        ops.tune_for_throughput(False)
        sync_all()

    if rank != 0:
        return
    images = B * args.steps * world
    line = {
        "metric": "images/sec (CIFAR-10 DDPM T=10 generation)", "value": images / elapsed, "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "median_ms_per_step": statistics.median(per_step_ms),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "train_steps_per_sec": train_sps, "hip_graph": use_graph,
        "config": {"workload": f"CIFAR-10 DDPM U-Net (35.7M params) VARSampler T={T} generation, "
                               f"{B} images/GPU/step, 3x32x32 (BASELINE configs[1]); noise drawn inside the step",
                   "images_per_gpu_per_step": B, "global_batch": B * world, "T": T,
                   "parallelism": f"dp{world} (independent trajectories, no collective)"},
        "train_config": {"per_gpu_batch": B, "global_batch": B * world, "timed_steps": args.train_steps,
                         "step": "sample T (in place in the replay ring) + update_f_v (1 energy + T TD steps) + update_sampler, "
                                 "fused Adam, device-side grad clip, dropout 0.1",
                         "grad_sync": "flat fp32 all-reduce over RCCL" if world > 1 else "none (1 GPU)"},
    }
    from dxmi_hip import _lib
    line["library"] = {"path": os.path.relpath(_lib.LIB_PATH, ROOT), "dxmi_version": int(_lib.load().dxmi_version()),
                       "selected_by_DXMI_LIB": bool(os.environ.get("DXMI_LIB"))}
    if world > 1:
        wire = getattr(getattr(tr, "sync_sampler", None), "wire_dtype", None) if args.train_steps > 0 else None
        line["rccl"] = {"world": world, "backend": torch.distributed.get_backend(),
                        "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if torch.distributed.get_backend() == "nccl" else None,
                        "wire_dtype": str(wire or torch.float32).replace("torch.", ""),
                        "collectives_in_generation": 0,
                        "collectives_per_train_step": f"{T + 1} value-net all-reduces (20.5 MB fp32) + the U-Net's ~32 MB buckets (143 MB fp32)"}
        line["per_rank_images_per_sec"] = [round(B * args.steps / e, 1) for e in rank_elapsed]
    if gen_summ:
        step_s = elapsed / args.steps
        convs = {k: v for k, v in gen_summ.items() if k[0].startswith("conv")}
        (cls, kid), s = max(convs.items(), key=lambda kv: kv[1]["ms"])
        tflops = s["flops"] / (s["ms"] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(kernel_name(kid))
        line["roofline"] = {
            "bound": "mfma", "achieved": tflops, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tflops / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
            "traffic_unit": "HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE), separate --pmc passes in " + traffic_src,
            "algorithmic_bytes_per_launch": s["bytes"] / s["launches"],
            "kernel": kernel_name(kid), "launches_per_step": s["launches"], "avg_launch_us": 1e3 * s["ms"] / s["launches"],
            "algorithmic_gflop_per_launch": s["flops"] / s["launches"] / 1e9,
            "algorithmic_gbps": s["bytes"] / (s["ms"] * 1e-3) / 1e9,
            "share_of_step_time": s["ms"] * 1e-3 / step_s,
            "events": "two extra steps after the timed region, every launch bracketed on its stream",
            # in-kernel shader clock (d s_memtime / d s_memrealtime of workgroup 0 over one conv_ws_kernel launch): the dense MFMA peak
            # is quoted at the 2.4 GHz boost clock (256 CUs x 4096 FLOP/clk), the chip holds less under this kernel's load, and the
            # boxes of the pool differ; frac_at_clock = achieved / (256 x 4096 x clock)
            "clock_ghz": None if gen_clock is None else round(gen_clock, 3),
            "frac_at_clock": None if not gen_clock else tflops / (256 * 4096 * gen_clock * 1e9 / 1e12),
        }
        line["roofline_classes"] = class_rooflines(gen_summ, step_s)
        # per conv kernel: time, rate, algorithmic bytes per launch and the HBM-side bytes per launch of the committed PMC passes
        line["conv_kernels"] = {kernel_name(k[1]): {"launches": v["launches"], "ms": round(v["ms"], 3),
                                                    "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                                                    "algorithmic_bytes_per_launch": round(v["bytes"] / v["launches"]),
                                                    "traffic": pmc_traffic(kernel_name(k[1]))[0]} for k, v in convs.items()}
    if train_summ:
        line["train_roofline_classes"] = class_rooflines(train_summ, t_train_step)
    def guarded(fn, *a, **k):
        """A leg with hipGraph replay; if the capture fails on this box the leg is measured without it (and says so)."""
        try:
            return fn(*a, **k)
        except Exception as e:
            if not k.get("graph"):
                raise
            graph_errors[fn.__name__] = repr(e)[:300]
            torch.cuda.synchronize()
            k["graph"] = False
            return fn(*a, **k)

    if world == 1 and not args.no_edm:
        # BASELINE configs[3] / [4] (EDM backbones) at their per-GPU sizes + the EDM train step: rank 0, outside the timed region
        line["edm"] = {"imagenet64_T10_b100": edm_generation_leg(device, "imagenet64_T10", 100, events=not args.no_events),
                       "lsun_bedroom_T4_b16": edm_generation_leg(device, "lsun_bedroom_T4", 16, events=not args.no_events),
                       "imagenet64_T10_train_b16": guarded(edm_train_leg, device, graph=use_graph)}
    if world == 1 and not args.no_small_batch:
        # the per-rank batches of the reference's multi-GPU runs (global batch // N), on this one GPU: hipGraph replay vs python issue
        line["small_batch"] = {
            "cifar10_T10_b32": {"what": "configs[1] at 8 GPUs: 256 // 8 images per rank, T = 10",
                                "graph": guarded(small_batch_leg, device, 32, 10, graph=use_graph), "eager": small_batch_leg(device, 32, 10, graph=False)},
            "cifar10_T4_b128": {"what": "configs[2] at 4 GPUs: 512 // 4 images per rank, T = 4, value_resample (T4_ddgan.yaml)",
                                "graph": guarded(small_batch_leg, device, 128, 4, graph=use_graph, value_resample=True),
                                "eager": small_batch_leg(device, 128, 4, graph=False, value_resample=True)}}
    if world == 1 and not args.no_eager_reference:
        eg = eager_reference_gpu(device, T, B)
        line["reference_eager_gpu"] = {
            "images_per_sec": {k: round(v, 1) for k, v in eg.items()}, "unit": "images/s",
            "speedup_vs_fp32": round(line["value"] / eg["fp32"], 2), "speedup_vs_bf16_autocast": round(line["value"] / eg["bf16_autocast"], 2),
            "what": "the reference's op sequence (oracle restatement: NCHW, unfused torch ops via MIOpen/rocBLAS) run eagerly "
                    f"on the same GPU, {B} images x T={T}, 2 timed repetitions after 1 warm-up; fp32 is the reference's precision"}
    if world == 1 and not args.no_cpu_baseline:
        cb = 16
        ncores = usable_cores()
        v8 = cpu_baseline(T, cb, 8)                      # the full configs[0] sample: 16 images x T steps
        # all usable cores: 1 of the T steps, scaled, in a child process with a 45 s limit (a 256-thread pool has been seen
        # to take 80 s per step on this pool's hosts: oversubscribed BLAS)
        try:
            r = subprocess.run([sys.executable, "-c", f"import bench; print(bench.cpu_baseline({T}, {cb}, {ncores}, steps=1))"],
                               cwd=ROOT, capture_output=True, text=True, timeout=45)
            vall = float(r.stdout.strip().splitlines()[-1])
        except (subprocess.TimeoutExpired, ValueError, IndexError):
            vall = None
        best, cores = (v8, 8) if (vall is None or v8 >= vall) else (vall, ncores)
        line["cpu_baseline"] = {"value": best, "unit": "images/s", "cores": cores, "kind": "port",
                                "at_8_threads": v8, "at_all_cores": vall, "all_cores": ncores, "logical_cpus": os.cpu_count(),
                                "sample": f"oracle (torch-CPU fp32 restatement), BASELINE configs[0] shape: {cb} images x T={T} at 8 threads "
                                          f"(whole sample) and at the {ncores} usable cores (1 of the {T} steps, rate scaled to T, null = "
                                          "not finished within 45 s); one repetition each; the better one is `value`"}
    # the scalars a reader of the driver's record needs, flat and EARLY in the line (its tail is cut)
    flat = {"train_steps_per_sec": train_sps, "hip_graph": use_graph}
    if graph_errors:
        line["hip_graph_errors"] = graph_errors
    if "edm" in line:
        flat.update({"c4_images_per_sec": line["edm"]["imagenet64_T10_b100"]["images_per_sec"],
                     "c5_images_per_sec": line["edm"]["lsun_bedroom_T4_b16"]["images_per_sec"],
                     "edm_train_steps_per_sec": line["edm"]["imagenet64_T10_train_b16"]["train_steps_per_sec"]})
    if "small_batch" in line:
        b32 = line["small_batch"]["cifar10_T10_b32"]
        flat.update({"train_b32_steps_per_sec": b32["graph"]["train"]["steps_per_sec"], "train_b32_ms_per_step": b32["graph"]["train"]["ms_per_step"],
                     "train_b32_host_issue_ms": b32["graph"]["train"]["host_issue_ms"], "train_b32_gpu_ms": b32["graph"]["train"]["gpu_ms"],
                     "train_b32_eager_ms_per_step": b32["eager"]["train"]["ms_per_step"],
                     "gen_b32_images_per_sec": b32["graph"]["gen"]["images_per_sec"], "gen_b32_host_issue_ms": b32["graph"]["gen"]["host_issue_ms"],
                     "train_b128_T4_steps_per_sec": line["small_batch"]["cifar10_T4_b128"]["graph"]["train"]["steps_per_sec"]})
    if "roofline" in line:
        flat["roofline_frac"] = round(line["roofline"]["frac"], 4)
    head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]
    ordered = {k: line[k] for k in head}
    ordered.update(flat)
    ordered.update({k: v for k, v in line.items() if k not in ordered})
    print(json.dumps(ordered))


if __name__ == "__main__":
    main()
