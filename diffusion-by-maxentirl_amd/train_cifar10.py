"""DxMI training for CIFAR-10 on MI355X (CLI-compatible with the reference's train_cifar10.py:208-462
for the training path).

    torchrun --nproc_per_node=N train_cifar10.py --config configs/cifar10/T10.yaml \
        --dataset configs/cifar10/cifar10.yaml --run myrun [--training.lr 1e-7 ...]
    (or --config builtin:cifar10_T10 --dataset builtin --synthetic_data for a self-contained run)

Same flow as the reference: config merge + `--a.b.c v` overrides, seeding with seed+rank, instantiate
net / sampler / value from `_target_`, Adam with the log_betas / rest split (reference :283-296),
per-rank batch = batchsize // world, epoch loop of `train_one_epoch` (:141-205), checkpoints with the
reference's file names and state-dict keys (:58-78, :459-462).  Differences, all outside the
accelerated path: gradients are exchanged by one flat RCCL all-reduce per backward instead of DDP
buckets (dxmi_hip/dist.py); FID / wandb / tensorboard are skipped unless their packages and the dataset
PNG folder exist; `--synthetic_data` feeds uniform images (benchmarks, smoke runs).
"""
import argparse
import os
import random

import numpy as np
import torch
from dxmi_hip import dist as _dist

import cmd_utils as cmd
import dxmi_config
from dxmi_hip.dist import broadcast_parameters
from dxmi_hip.optim import Adam          # torch.optim.Adam subclass: step() is one multi-tensor HIP kernel series
from models.DxMI.replay import TransitionRing
from models.DxMI.trainer import append_buffer, reset_buffer
from utils import fix_legacy_dict, mkdir_p, print0


def save_model(trainer, logdir, postfix, d_other_info=None):
    """sampler_{postfix}.pth / value_{postfix}.pth with reference key names (train_cifar10.py:58-78)."""
    d = {"state_dict": trainer.sampler.net.state_dict()}
    d.update(d_other_info or {})
    torch.save(d, os.path.join(logdir, f"sampler_{postfix}.pth"))
    if trainer.v is not None:
        torch.save({"state_dict": trainer.v.state_dict()}, os.path.join(logdir, f"value_{postfix}.pth"))


def synthetic_loader(batchsize, n_batches, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    for _ in range(n_batches):
        yield torch.rand(batchsize, 3, 32, 32, device=device, generator=g), None


def train_one_epoch(trainer, sampler, dataloader, n_critic, n_generator, device, log_every, state):
    """reference train_one_epoch (:141-205) without the FID / logging side paths."""
    # device-resident replay ring: n_critic trajectories of T transitions each, generated in place by the sampler
    buf = TransitionRing(n_critic, trainer.n_timesteps, trainer.batchsize, sampler.sample_shape, device)
    for step, (images, _) in enumerate(dataloader):
        sampler.eval()
        images = (2 * images - 1).to(device)
        in_ring = len(images) == trainer.batchsize
        d_sample = sampler.sample(len(images), device=device, out=buf.next_slot() if in_ring else None)
        append_buffer(buf, d_sample)
        d_energy = trainer.update_f_v(images, d_sample, buf)
        if (step + 1) % n_critic == 0:
            d_sampler = trainer.update_sampler(buf, n_generator)
            buf = reset_buffer(device, ring=buf)
            if (step + 1) % log_every == 0:
                print0(f"iter {state['i_iter']}: d_loss {d_energy['ebm/d_loss_']:.4f} v_loss {d_energy['ebm/v_loss_']:.4f} "
                       f"sampler_loss {d_sampler['sampler/sampler_loss_']:.4f}")
        state["i_iter"] += 1


def load_config(config, dataset, overrides=None):
    """--config / --dataset / `--a.b.c v` overrides -> merged Cfg (reference :228-233: OmegaConf.load x 2 + merge)."""
    if config.startswith("builtin:"):
        import configs_builtin
        cfg = configs_builtin.get(config.split(":", 1)[1])
    else:
        cfg = dxmi_config.merge(dxmi_config.load(config), dxmi_config.load(dataset))
    return dxmi_config.merge(cfg, overrides or {})


def build_optimizers(cfg, net, sampler, v):
    """Adam with the log_betas / rest learning-rate split, Adam for the value net (reference :283-296)."""
    tune_beta = bool(sampler.trainable_beta) and cfg.training.get("beta_lr") is not None
    if tune_beta:
        not_beta = [p for n, p in net.named_parameters() if "log_betas" not in n]
        optimizer = Adam([{"params": net.log_betas, "lr": cfg.training.beta_lr},
                          {"params": not_beta, "lr": cfg.training.lr}])
    else:
        optimizer = Adam(net.parameters(), lr=cfg.training.lr)
    return optimizer, Adam(v.parameters(), lr=cfg.training.v_lr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--dataset", type=str, required=True)
    ap.add_argument("--run", type=str, default="run")
    ap.add_argument("--synthetic_data", action="store_true")
    ap.add_argument("--max_iters", type=int, default=None, help="stop after this many iterations (smoke runs)")
    ap.add_argument("--batch_invariant", action="store_true",
                    help="keep one conv kernel per layer shape whatever the batch size (bitwise batch-independent results) instead of "
                         "routing under-filled grids to smaller tiles (dxmi_hip.ops.tune_for_throughput)")
    ap.add_argument("--no_graph", action="store_true",
                    help="issue every kernel launch from python instead of replaying the generation call and the two updates as hipGraphs "
                         "(dxmi_hip/graph.py; DXMI_GRAPH=0 does the same)")
    args, unknown = ap.parse_known_args()
    d_cmd_cfg = cmd.parse_nested_args(cmd.parse_unknown_args(unknown))
    print0("Overriding", d_cmd_cfg)

    cfg = load_config(args.config, args.dataset, d_cmd_cfg)

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    net = dxmi_config.instantiate(cfg.sampler_net)
    sampler = dxmi_config.instantiate(cfg.sampler, net=net).to(device)
    if cfg.training.sampler_ckpt and os.path.exists(cfg.training.sampler_ckpt):
        net.load_state_dict(fix_legacy_dict(torch.load(cfg.training.sampler_ckpt, map_location="cpu")), strict=False)
        print0(f"Sampler checkpoint loaded from {cfg.training.sampler_ckpt}")
    else:
        print0("no sampler checkpoint found: random initialisation")
    v = dxmi_config.instantiate(cfg.value).to(device)
    if cfg.training.get("value_ckpt") is not None:
        v.load_pretrained(torch.load(cfg.training.value_ckpt, map_location="cpu"))
    broadcast_parameters(net)
    broadcast_parameters(v)

    optimizer, optimizer_v = build_optimizers(cfg, net, sampler, v)

    batchsize = cfg.training.batchsize // world     # the reference divides by the visible device count (:298-301)
    trainer = dxmi_config.instantiate(cfg.trainer, batchsize=batchsize)
    trainer.set_models(f=None, v=v, sampler=sampler, optimizer=optimizer, optimizer_fstar=None, optimizer_v=optimizer_v)
    # per-rank batches of a multi-GPU run are small (batchsize // world): the step is replayed from hipGraphs so that the host does not
    # bound it; gradient exchanges stay eager RCCL calls at graph cuts (dxmi_hip/dist.py)
    from dxmi_hip import graph as hip_graph
    trainer.use_graphs = sampler.use_graph = hip_graph.default_enabled() and not args.no_graph

    model_cfg_name = os.path.basename(args.config).split(".")[0].replace("builtin:", "")
    logdir = os.path.join(f"results/{cfg.data.name}/{model_cfg_name}", args.run)
    if local_rank == 0:
        mkdir_p(logdir)
        dxmi_config.save(cfg, os.path.join(logdir, "config.yaml"))

    state = {"i_iter": 0}
    for epoch in range(cfg.training.n_epochs):
        if args.synthetic_data:
            n_batches = args.max_iters or 100
            loader = synthetic_loader(batchsize, n_batches, device, seed + local_rank + epoch)
        else:
            import loader as ref_loader  # the reference's torchvision CIFAR-10 pipeline (out of scope here)
            from torch.utils.data import DataLoader
            from torch.utils.data.distributed import DistributedSampler
            train_set = ref_loader.get_dataset(cfg.data.name, cfg.data.data_dir)
            ds = DistributedSampler(train_set) if world > 1 else None
            loader = DataLoader(train_set, batch_size=batchsize, shuffle=ds is None, sampler=ds, num_workers=4,
                                pin_memory=True, drop_last=True)
        train_one_epoch(trainer, sampler, loader, cfg.training.n_critic, cfg.training.n_generator, device,
                        cfg.training.log_every, state)
        if args.max_iters is not None and state["i_iter"] >= args.max_iters:
            break
    if local_rank == 0:
        save_model(trainer, logdir, "last", d_other_info={"epoch": epoch, "iter": state["i_iter"], "fid": None})
        print0(f"saved {logdir}/sampler_last.pth and value_last.pth after {state['i_iter']} iterations")


if __name__ == "__main__":
    main()
