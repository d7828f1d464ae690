"""Full-size EDM DxMI train step on one GPU (synthetic data): python tools/edm_train_bench.py imagenet64_T10 16 [steps] [graph]
graph: the three phases replayed from hipGraphs on the device ring (train_image_large.py's default), else python-issued on a dict buffer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")]
import torch
from dxmi_hip import ops
from dxmi_hip.optim import Adam, RAdam          # the shipped optimisers (train_image_large.py)
import configs_builtin, dxmi_config
from models.cm.fp16_util import MixedPrecisionTrainer
from models.cm.script_util import create_model_and_diffusion
from models.DxMI.openai_diffusion import OpenAIDiffusion
from models.DxMI.trainer import append_buffer, reset_buffer

if os.environ.get("DXMI_BATCH_INVARIANT", "0") != "1":
    ops.tune_for_throughput()          # train_image_large.py's default
name, B = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
GRAPH = len(sys.argv) > 4 and sys.argv[4] == "graph"
dev = "cuda:0"
cfg = configs_builtin.get(name)
torch.manual_seed(0)
unet, diffusion = create_model_and_diffusion(**cfg.diffusion)
for p in unet.parameters():
    if p.abs().max() == 0:
        torch.nn.init.normal_(p, std=0.02)
sampler = OpenAIDiffusion(unet, diffusion, **cfg.sampler)
unet.to(dev)
v = dxmi_config.instantiate(cfg.value).to(dev)
mp = MixedPrecisionTrainer(model=unet, use_fp16=True, initial_lg_loss_scale=20, special_key="log_betas")
opt = RAdam([{"params": mp.master_params[1:], "lr": 1e-8}, {"params": mp.master_params[0:1], "lr": 1e-6}])
opt_v = Adam(v.parameters(), lr=1e-5)
trainer = dxmi_config.instantiate(cfg.trainer, batchsize=B)
trainer.set_models(v=v, sampler=sampler, optimizer=opt, optimizer_v=opt_v)
res = cfg.diffusion.image_size
g = torch.Generator(device=dev).manual_seed(1)
ring = None
if GRAPH:
    from models.DxMI.replay import TransitionRing
    trainer.use_graphs = sampler.use_graph = True
    ring = TransitionRing(1, trainer.n_timesteps, B, sampler.sample_shape, dev, with_y=bool(cfg.sampler.class_cond), sigma_dims=1)


def step():
    data = torch.rand(B, 3, res, res, device=dev, generator=g) * 2 - 1
    y = torch.randint(0, 1000, (B,), device=dev, generator=g) if cfg.sampler.class_cond else None
    t = [time.perf_counter()]
    sampler.eval()
    d = sampler.sample(B, device=dev, i_class=y, out=ring.next_slot() if ring is not None else None)
    torch.cuda.synchronize(); t.append(time.perf_counter())
    buf = append_buffer(ring if ring is not None else reset_buffer(dev), d)
    le = trainer.update_f_v(data, d, buf, y=y)
    torch.cuda.synchronize(); t.append(time.perf_counter())
    ls = trainer.update_sampler_mixed_precision(buf, mp_trainer=mp)
    torch.cuda.synchronize(); t.append(time.perf_counter())
    if ring is not None:
        reset_buffer(dev, ring=ring)
    return [b - a for a, b in zip(t[:-1], t[1:])], le, ls


for _ in range(3 if GRAPH else 1):
    step()
tot = [0, 0, 0]
for _ in range(steps):
    dt, le, ls = step()
    tot = [a + b for a, b in zip(tot, dt)]
tot = [x / steps for x in tot]
if os.environ.get("DXMI_HOST_COST", "0") == "1":
    # HOST cost of the same step: every dxmi_* kernel entry point stubbed out (returns DXMI_OK without launching; size / selection
    # queries still answer), torch's own small ops still run.  Compare with the real step: host-bound where the two are close.
    from dxmi_hip import _lib
    real = _lib.load()

    class Stub:
        def __getattr__(self, name):
            f = getattr(real, name)
            if name.endswith(("_bytes", "_partials", "_supported", "_slices")) or name in ("dxmi_mt_blocks", "dxmi_last_error", "dxmi_conv2d_kernel_id",
                                                                                                  "dxmi_device_check", "dxmi_version", "dxmi_get_tuning", "dxmi_set_tuning"):
                return f
            return lambda *a: 0
    _lib._lib = Stub()
    step()
    if os.environ.get("DXMI_HOST_PROFILE", "0") == "1":
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        step()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(30)
    th = [0, 0, 0]
    for _ in range(steps):
        dt, _, _ = step()
        th = [a + b for a, b in zip(th, dt)]
    _lib._lib = real
    th = [x / steps for x in th]
    print(f"host only (kernels stubbed): sample {th[0]*1e3:.0f} ms, update_f_v {th[1]*1e3:.0f} ms, update_sampler {th[2]*1e3:.0f} ms = {sum(th)*1e3:.0f} ms per step")
print(f"{name} B={B}: sample {tot[0]*1e3:.0f} ms, update_f_v {tot[1]*1e3:.0f} ms, update_sampler {tot[2]*1e3:.0f} ms -> {1/sum(tot):.3f} steps/s; "
      f"v_loss {le['ebm/v_loss_']:.4f} sampler_loss {ls['sampler/sampler_loss_']:.4f} lg_loss_scale {mp.lg_loss_scale:.3f} "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
