"""DxMI training on the EDM backbones (ImageNet-64, LSUN) on MI355X; CLI-compatible with the reference's
train_image_large.py:94-330 for the training path.

    torchrun --nproc_per_node=N train_image_large.py --config configs/imagenet64/T10.yaml \
        --dataset configs/imagenet64/imagenet64.yaml --run myrun [--training.lr 1e-8 ...]
    (or --config builtin:imagenet64_T10 --dataset builtin --synthetic_data for a self-contained run)

Same flow as the reference: config merge + `--a.b.c v` overrides, seeding with seed+rank, create_model_and_diffusion,
pretrained EDM weights when the file exists, OpenAIDiffusion, value net from `_target_`, MixedPrecisionTrainer with the
`log_betas` special key and RAdam on its master tensors (:152-168), Adam for the value net, per-rank batch =
batchsize // world, and per iteration sample -> append_buffer -> update_f_v -> update_sampler_mixed_precision ->
reset_buffer (:301-321).  Differences, all outside the accelerated path: gradients are exchanged by one flat RCCL
all-reduce per backward (dxmi_hip/dist.py) instead of DDP buckets; FID / wandb / tensorboard are skipped unless their
packages and statistics files exist; `--synthetic_data` feeds uniform images and random labels.
Checkpoints: `sampler.pth` ({'state_dict', 'fid', 'i_iter'}) and `value.pth`, the names generate_large.py reads.
"""
import argparse
import os
import random

import numpy as np
import torch
from dxmi_hip import dist as _dist
from dxmi_hip.optim import Adam, RAdam   # torch.optim subclasses: step() is one multi-tensor HIP kernel series

import cmd_utils as cmd
import dxmi_config
from dxmi_hip.dist import broadcast_parameters
from models.cm.fp16_util import MixedPrecisionTrainer
from models.cm.script_util import create_model_and_diffusion
from models.DxMI.openai_diffusion import OpenAIDiffusion
from models.DxMI.trainer import append_buffer, reset_buffer
from utils import mkdir_p, print0


def synthetic_batches(batchsize, image_size, class_cond, n_class, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    while True:
        data = torch.rand(batchsize, 3, image_size, image_size, device=device, generator=g) * 2 - 1
        cond = {"y": torch.randint(0, n_class, (batchsize,), device=device, generator=g)} if class_cond else {}
        yield data, cond


def load_config(config, dataset, overrides=None):
    """--config / --dataset / `--a.b.c v` overrides -> merged Cfg (reference train_image_large.py:120-127)."""
    if config.startswith("builtin:"):
        import configs_builtin
        cfg = configs_builtin.get(config.split(":", 1)[1])
    else:
        cfg = dxmi_config.merge(dxmi_config.load(config), dxmi_config.load(dataset))
    return dxmi_config.merge(cfg, overrides or {})


def build_optimizers(cfg, unet, v):
    """MixedPrecisionTrainer + RAdam on its master tensors with the log_betas learning rate split, Adam for the value
    net (reference train_image_large.py:152-168)."""
    lls = cfg.training.get("initial_log_loss_scale", 20)
    if cfg.training.get("beta_lr") is not None:
        mp_trainer = MixedPrecisionTrainer(model=unet, use_fp16=cfg.diffusion.use_fp16, initial_lg_loss_scale=lls, special_key="log_betas")
        if cfg.diffusion.use_fp16:
            groups = [{"params": mp_trainer.master_params[1:], "lr": cfg.training.lr},
                      {"params": mp_trainer.master_params[0:1], "lr": cfg.training.beta_lr}]
        else:   # masters are the model parameters themselves: split by name
            groups = [{"params": [p for n, p in unet.named_parameters() if "log_betas" not in n], "lr": cfg.training.lr},
                      {"params": [unet.log_betas], "lr": cfg.training.beta_lr}]
        opt = RAdam(groups, weight_decay=cfg.training.get("weight_decay", 0.0))
    else:
        mp_trainer = MixedPrecisionTrainer(model=unet, use_fp16=cfg.diffusion.use_fp16, initial_lg_loss_scale=lls)
        opt = RAdam(mp_trainer.master_params, lr=cfg.training.lr, weight_decay=cfg.training.get("weight_decay", 0.0))
    return mp_trainer, opt, Adam(v.parameters(), lr=cfg.training.v_lr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--dataset", type=str, required=True)
    ap.add_argument("--run", type=str, required=True)
    ap.add_argument("--synthetic_data", action="store_true")
    ap.add_argument("--max_iters", type=int, default=None, help="stop after this many iterations (smoke runs)")
    ap.add_argument("--batch_invariant", action="store_true",
                    help="keep one conv kernel per layer shape whatever the batch size (bitwise batch-independent results) instead of "
                         "routing under-filled grids to smaller tiles (dxmi_hip.ops.tune_for_throughput)")
    ap.add_argument("--fid_extractor", type=str, default=None,
                    help="'module:attribute' of the FID feature extractor (the reference builds pytorch_fid's InceptionV3, whose weights "
                         "this image cannot download); with --fid_stats it enables the periodic fid() of training.fid_every")
    ap.add_argument("--fid_stats", type=str, default=None, help="dataset statistics npz (`mu`, `sigma`; reference: datasets/VIRTUAL_*.npz)")
    ap.add_argument("--fid_dims", type=int, default=2048)
    ap.add_argument("--no_graph", action="store_true",
                    help="issue every kernel launch from python instead of replaying the three phases of an iteration as hipGraphs "
                         "(dxmi_hip/graph.py; DXMI_GRAPH=0 does the same)")
    args, unknown = ap.parse_known_args()
    d_cmd_cfg = cmd.parse_nested_args(cmd.parse_unknown_args(unknown))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if local_rank == 0:
        print("Overriding", d_cmd_cfg)

    cfg = load_config(args.config, args.dataset, d_cmd_cfg)

    device = _dist.rank_device(local_rank)
    torch.cuda.set_device(device)
    if not args.batch_invariant:
        from dxmi_hip import ops as _ops
        _ops.tune_for_throughput()
    seed = cfg.training.seed
    torch.manual_seed(seed + local_rank)
    np.random.seed(seed + local_rank)
    torch.cuda.manual_seed_all(seed + local_rank)
    random.seed(seed + local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend=_dist.dist_backend(), init_method="env://")   # RCCL

    unet, diffusion = create_model_and_diffusion(**cfg.diffusion)
    pre = cfg.training.get("pretrained_path")
    if pre and os.path.exists(pre):
        unet.load_state_dict(torch.load(pre, map_location="cpu"))
        print0(f"pretrained EDM weights loaded from {pre}")
    else:
        print0("no pretrained EDM checkpoint found: random initialisation (zero-initialised output layers stay zero)")
    sampler = OpenAIDiffusion(unet, diffusion, **cfg.sampler)
    unet.to(device)
    if cfg.diffusion.use_fp16:
        unet.convert_to_fp16()
    v = dxmi_config.instantiate(cfg.value)
    if cfg.training.get("value_ckpt") is not None:
        v.load_pretrained(torch.load(cfg.training.value_ckpt, map_location="cpu"))
    v.to(device)
    broadcast_parameters(unet)
    broadcast_parameters(v)

    mp_trainer, opt, opt_v = build_optimizers(cfg, unet, v)

    batchsize = cfg.training.batchsize // world
    class_cond = bool(cfg.data.get("class_cond", cfg.sampler.get("class_cond", False)))
    if args.synthetic_data:
        loader = synthetic_batches(batchsize, cfg.diffusion.image_size, class_cond, cfg.data.get("n_class", 1000), device, seed + local_rank)
    else:
        from models.cm.dxmi_util import infinite_loader, load_data   # the reference's image-folder pipeline (out of scope here)
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        ds = load_data(data_dir=cfg.data.data_dir, cachefile=cfg.data.cachefile, batch_size=cfg.training.batchsize,
                       image_size=cfg.data.image_size, class_cond=cfg.data.class_cond, deterministic=cfg.data.deterministic,
                       random_crop=False, random_flip=True)
        dsamp = DistributedSampler(ds) if world > 1 else None
        loader = infinite_loader(DataLoader(ds, batch_size=batchsize, shuffle=dsamp is None, sampler=dsamp, num_workers=4, drop_last=True))

    model_cfg_name = os.path.basename(args.config).split(".")[0].replace("builtin:", "")
    logdir = os.path.join(f"results/{cfg.data.name}/{model_cfg_name}", args.run)
    if local_rank == 0:
        mkdir_p(logdir)
        dxmi_config.save(cfg, os.path.join(logdir, "config.yaml"))

    trainer = dxmi_config.instantiate(cfg.trainer, batchsize=batchsize)
    trainer.set_models(v=v, sampler=sampler, optimizer=opt, optimizer_v=opt_v)
    # an iteration is ~20 k launches at a per-rank batch of 16: replayed from hipGraphs (sampling, value update, the K policy
    # iterations with the loss-scale bookkeeping on the device) the host no longer bounds it; gradient exchanges stay eager RCCL calls
    from dxmi_hip import graph as hip_graph
    trainer.use_graphs = sampler.use_graph = hip_graph.default_enabled() and not args.no_graph

    n_iter = cfg.training.n_iter if args.max_iters is None else min(cfg.training.n_iter, args.max_iters)
    # device-resident replay ring (one trajectory per iteration, generated in place by the sampler; sigma is 1-D here)
    from models.DxMI.replay import TransitionRing
    state_dict = TransitionRing(1, trainer.n_timesteps, batchsize, sampler.sample_shape, device, with_y=class_cond, sigma_dims=1)
    # periodic FID (reference :35-87, :256-257): n_fid_samples / sampling_batchsize / world batches per rank -> uint8 -> all ranks'
    # activations gathered over RCCL -> statistics on the device (dxmi_fid_stats) -> distance; the best sampler is kept
    fid_on = args.fid_extractor is not None and args.fid_stats is not None and cfg.training.get("fid_every") is not None
    best_fid = float("inf")
    if fid_on:
        from dxmi_hip import ops
        from pytorch_fid.fid_score import fid_from_images, load_extractor, load_statistics
        extractor = load_extractor(args.fid_extractor)
        extractor = extractor.to(device) if hasattr(extractor, "to") else extractor
        m2, s2 = load_statistics(args.fid_stats)

    def fid(it):
        nonlocal best_fid
        sb = cfg.training.sampling_batchsize
        sampler.eval()
        mine = [ops.quantize_u8(sampler.sample(sb, device=device, i_class=None, enable_grad=False)["sample"].contiguous().float(),
                                mode=1, nhwc=False) for _ in range(int(cfg.training.n_fid_samples / sb / world))]
        samples = torch.cat(mine)
        if world > 1:                                   # reference :50-53, :59: gather the images, take this rank's strided share
            parts = [torch.zeros_like(samples) for _ in range(world)]
            torch.distributed.all_gather(parts, samples)
            samples = torch.cat(parts)
        value = fid_from_images(samples[local_rank::world], extractor, m2, s2, batch_size=50, dims=args.fid_dims, device=device)
        print0(f"FID: {value}")
        if local_rank == 0 and value < best_fid:
            best_fid = value
            np.savez(os.path.join(logdir, "best_samples.npz"), samples.permute(0, 2, 3, 1).cpu().numpy())
            torch.save({"state_dict": unet.state_dict(), "fid": value, "i_iter": it}, os.path.join(logdir, "sampler.pth"))
            torch.save({"state_dict": v.state_dict()}, os.path.join(logdir, "value.pth"))
            print0(f"best FID: sampler saved at {os.path.join(logdir, 'sampler.pth')}")

    i_iter = -1
    for i_iter in range(n_iter):
        data, cond = next(loader)
        data = data.to(device)
        y = cond.get("y", None)
        y = y.to(device) if y is not None else None
        if fid_on and i_iter % cfg.training.fid_every == 0:
            fid(i_iter)
        sampler.eval()
        d_sample = sampler.sample(len(data), device=device, i_class=y, out=state_dict.next_slot() if len(data) == batchsize else None)
        append_buffer(state_dict, d_sample)
        d_energy = trainer.update_f_v(data, d_sample, state_dict, y=y)
        d_sampler = trainer.update_sampler_mixed_precision(state_dict, mp_trainer=mp_trainer, d_sample=d_sample)
        state_dict = reset_buffer(device, ring=state_dict)
        if (i_iter + 1) % cfg.training.log_every == 0:
            print0(f"iter {i_iter}: d_loss {d_energy['ebm/d_loss_']:.4f} v_loss {d_energy['ebm/v_loss_']:.4f} "
                   f"sampler_loss {d_sampler['sampler/sampler_loss_']:.4f} lg_loss_scale {mp_trainer.lg_loss_scale:.3f}")
    if local_rank == 0 and not (fid_on and best_fid < float("inf")):     # with FID tracking the best sampler is what stays on disk (reference :73-84)
        torch.save({"state_dict": unet.state_dict(), "fid": None, "i_iter": i_iter}, os.path.join(logdir, "sampler.pth"))
        torch.save({"state_dict": v.state_dict()}, os.path.join(logdir, "value.pth"))
        print0(f"saved {logdir}/sampler.pth and value.pth after {i_iter + 1} iterations")


if __name__ == "__main__":
    main()
