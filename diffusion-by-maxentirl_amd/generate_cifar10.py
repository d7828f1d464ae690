"""Generate samples from a trained DxMI sampler on MI355X (CLI-compatible with the reference's
generate_cifar10.py:45-233 for the generation path).

    torchrun --nproc_per_node=N generate_cifar10.py --log_dir results/cifar10/T10/run -n 50000 --batchsize 100

Reads `config.yaml` + `sampler_{epoch}.pth` from --log_dir (as written by train_cifar10.py), shards the
images over ranks with seed+rank (reference :103-110, :193-204), runs the HIP sampler and writes
`{rank}_{i}.png` under <log_dir>/generated.  FID (pytorch_fid + Inception weights + the training PNGs)
is outside the accelerated path: it runs only when pytorch_fid and the dataset folder are present,
otherwise it is skipped with a message (`--skip_fid` forces that).  `--synthetic` builds the net from
the built-in config with random weights (benchmark / smoke use, no checkpoint needed).
"""
import argparse
import os
import random
import time

import numpy as np
import torch
from dxmi_hip import dist as _dist

import cmd_utils as cmd  # noqa: F401  (kept for CLI parity: unknown --a.b overrides are parsed the same way)
import dxmi_config
from utils import mkdir_p, print0, to_uint8_nhwc, write_png_batch


def rescale(X):
    return (X - (-1)) / 2


def save_png(img_chw, path):
    """uint8 PNG of a [3,H,W] tensor in [0,1] (torchvision.utils.save_image rounding: x*255+0.5)."""
    write_png_batch(to_uint8_nhwc(img_chw[None]), [path], workers=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log_dir", type=str, required=True)
    ap.add_argument("--batchsize", type=int, default=100)
    ap.add_argument("-n", "--n_generate", type=int, default=50000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--epoch", type=str, default="best")
    ap.add_argument("-save", "--save_images", type=bool, default=True)
    ap.add_argument("--guidance_scale", type=float, default=None)
    ap.add_argument("--stat", type=str, default=None)
    ap.add_argument("--skip_fid", action="store_true")
    ap.add_argument("--fid_extractor", type=str, default=None,
                    help="'module:attribute' of the feature extractor the FID uses (the reference builds pytorch_fid's InceptionV3, "
                         "whose weights this image cannot download): callable(batch in [0,1]) -> [features [B, 2048, h, w]]")
    ap.add_argument("--fid_stats", type=str, default=None, help="statistics npz (`mu`, `sigma`) instead of the dataset's PNG folder")
    ap.add_argument("--synthetic", type=str, default=None, help="builtin config name, e.g. cifar10_T10 (random weights)")
    ap.add_argument("--no_graph", action="store_true", help="issue every launch from python instead of replaying the T-step loop of a "
                                                           "batch as one hipGraph (dxmi_hip/graph.py; DXMI_GRAPH=0 does the same)")
    args, unknown = ap.parse_known_args()

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = _dist.rank_device(local_rank)
    torch.cuda.set_device(device)
    seed = args.seed
    torch.manual_seed(seed + local_rank)
    np.random.seed(seed + local_rank)
    torch.cuda.manual_seed_all(seed + local_rank)
    random.seed(seed + local_rank)
    assert args.n_generate % args.batchsize == 0, "n_generate must be a multiple of batchsize"

    if args.synthetic:
        import configs_builtin
        run_config = configs_builtin.get(args.synthetic)
    else:
        config_path = os.path.join(args.log_dir, "config.yaml")
        if not os.path.exists(config_path):
            raise ValueError(f"Config not found at {config_path}")
        run_config = dxmi_config.load(config_path)
    output_path = os.path.join(args.log_dir, "generated" if args.guidance_scale is None else f"generated_{args.guidance_scale}")
    mkdir_p(output_path)

    net = dxmi_config.instantiate(run_config.sampler_net)
    sampler = dxmi_config.instantiate(run_config.sampler, net=net).to(device)
    if not args.synthetic:
        sampler_path = os.path.join(args.log_dir, f"sampler_{args.epoch}.pth")
        if not os.path.exists(sampler_path) and os.path.exists(os.path.join(args.log_dir, "sampler.pth")):
            sampler_path = os.path.join(args.log_dir, "sampler.pth")
        if not os.path.exists(sampler_path):
            raise ValueError(f"Sampler not found at {sampler_path}")
        ckpt = torch.load(sampler_path, map_location=device)
        sampler.net.load_state_dict(ckpt["state_dict"])
        print0(f"Loaded sampler from {sampler_path} (epoch {ckpt.get('epoch')}, FID {ckpt.get('fid')})")
    sampler.eval()
    from dxmi_hip import graph as hip_graph
    sampler.use_graph = hip_graph.default_enabled() and not args.no_graph     # second batch onwards: one hipGraphLaunch per batch

    trainer = None
    if args.guidance_scale is not None:   # reference :160-191: value net from value_best.pth, trainer only as the sampling driver
        v = dxmi_config.instantiate(run_config.value).to(device)
        if not args.synthetic:
            value_path = os.path.join(args.log_dir, "value_best.pth")
            if not os.path.exists(value_path) and os.path.exists(os.path.join(args.log_dir, f"value_{args.epoch}.pth")):
                value_path = os.path.join(args.log_dir, f"value_{args.epoch}.pth")
            if not os.path.exists(value_path):
                raise ValueError(f"Value ftn not found at {value_path}")
            v.load_state_dict(torch.load(value_path, map_location=device)["state_dict"])
        v.eval()
        trainer = dxmi_config.instantiate(run_config.trainer, batchsize=args.batchsize)
        trainer.set_models(f=None, v=v, sampler=sampler, optimizer=None, optimizer_fstar=None, optimizer_v=None)

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend=_dist.dist_backend(), init_method="env://")  # RCCL; only the final barrier uses it

    n_batches = int(args.n_generate / args.batchsize / world)
    i_img = 0
    from utils import ImageWriter
    writer = ImageWriter()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_batches):
        with torch.no_grad():
            if trainer is not None:
                d_sample = trainer.sample_guidance(n_sample=args.batchsize, device=device, guidance_scale=args.guidance_scale)
            else:
                d_sample = sampler.sample(args.batchsize, device=device)
        if args.save_images:
            # rescale -> clamp -> save_image rounding on the device (dxmi_quantize_u8), pinned double-buffered copy on a side
            # stream, PNGs from a thread pool: the sampler never waits for the files
            n = d_sample["sample"].shape[0]
            writer.submit(d_sample["sample"], [os.path.join(output_path, f"{local_rank}_{i_img + k}.png") for k in range(n)])
            i_img += n
    writer.close()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        torch.distributed.barrier()
    print0(f"Generated {args.n_generate} samples at {output_path} "
           f"({n_batches * args.batchsize / dt:.1f} images/s/rank incl. PNG writing)")

    data_path = args.fid_stats or os.path.join("datasets", f"{run_config.data.name}_train_png")
    if args.skip_fid or local_rank != 0:
        return
    if args.fid_extractor is None:
        print0("FID skipped: pass --fid_extractor module:attr (an InceptionV3 pool3 extractor; its weights are not in this image)")
        return
    if not os.path.exists(data_path):
        print0(f"Dataset not found at {data_path}: FID skipped")
        return
    # reference :212-216: FID of the generated PNG folder against the dataset; the activation statistics run on the device
    from pytorch_fid.fid_score import calculate_fid_given_paths, load_extractor
    extractor = load_extractor(args.fid_extractor)
    if hasattr(extractor, "to"):
        extractor = extractor.to(device)
    fid = calculate_fid_given_paths([output_path, data_path], batch_size=args.batchsize, device=device, dims=2048, extractor=extractor)
    print(f"FID score: {fid}")


if __name__ == "__main__":
    main()
