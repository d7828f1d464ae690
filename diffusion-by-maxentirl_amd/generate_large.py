"""Generate samples from an EDM-backbone DxMI sampler (ImageNet-64, LSUN-256) on MI355X; CLI-compatible with
the reference's generate_large.py:85-190 for the generation path.

    torchrun --nproc_per_node=N generate_large.py --log_dir results/imagenet64/T10/run --n_sample 50000 --batchsize 100

Reads `config.yaml` + `sampler.pth` from --log_dir, builds models.cm U-Net + OpenAIDiffusion, seeds every rank
with seed+rank (reference :103-114) and generates n_sample / batchsize / world batches per rank; --batchsize is
PER RANK, as in the reference.  Ranks never exchange data while sampling (the reference wraps the net in DDP
only to broadcast weights; here every rank loads the same checkpoint), so the only collective is the final
gather of uint8 images for the FID npz.  FID itself (pytorch_fid + Inception weights + dataset statistics) is
outside the accelerated path and runs only when those are present; --skip_fid writes PNGs as images are produced.
`--synthetic NAME` builds the net of a built-in config with random weights (benchmark / smoke use).
"""
import argparse
import os
import random
import time

import numpy as np
import torch
from dxmi_hip import dist as _dist

import dxmi_config
from models.cm.script_util import create_model_and_diffusion
from models.DxMI.openai_diffusion import OpenAIDiffusion
from utils import mkdir_p, print0, to_uint8_nhwc, write_png_batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log_dir", type=str, required=True, help="path to logdir")
    ap.add_argument("--n_sample", type=int, required=True)
    ap.add_argument("--batchsize", type=int, default=100)
    ap.add_argument("--guidance_scale", type=float, default=None)
    ap.add_argument("--skip_fid", action="store_true")
    ap.add_argument("--fid_extractor", type=str, default=None,
                    help="'module:attribute' of the feature extractor (the reference builds pytorch_fid's InceptionV3, whose "
                         "weights this image cannot download): callable(batch in [0,1]) -> [features [B, dims, h, w]]")
    ap.add_argument("--fid_stats", type=str, default=None, help="npz with the dataset's `mu` / `sigma` (reference: datasets/VIRTUAL_*.npz)")
    ap.add_argument("--fid_dims", type=int, default=2048)
    ap.add_argument("--synthetic", type=str, default=None, help="builtin config name, e.g. imagenet64_T10 (random weights)")
    ap.add_argument("--no_graph", action="store_true", help="issue every launch from python instead of replaying the T-step loop of a "
                                                           "batch as one hipGraph (dxmi_hip/graph.py; DXMI_GRAPH=0 does the same)")
    args, unknown = ap.parse_known_args()

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = _dist.rank_device(local_rank)
    torch.cuda.set_device(device)
    if args.synthetic:
        import configs_builtin
        cfg = configs_builtin.get(args.synthetic)
    else:
        cfg = dxmi_config.load(os.path.join(args.log_dir, "config.yaml"))
    seed = cfg.training.seed
    torch.manual_seed(seed + local_rank)
    np.random.seed(seed + local_rank)
    torch.cuda.manual_seed_all(seed + local_rank)
    random.seed(seed + local_rank)

    unet, diffusion = create_model_and_diffusion(**cfg.diffusion)
    sampler = OpenAIDiffusion(unet, diffusion, **cfg.sampler)
    output_path = os.path.join(args.log_dir, "generated")
    mkdir_p(output_path)
    if not args.synthetic:
        ckpt_path = os.path.join(args.log_dir, "sampler.pth")
        ckpt = torch.load(ckpt_path, map_location="cpu")
        print0(f"checkpoint loaded from {ckpt_path} (FID {ckpt.get('fid')}, iter {ckpt.get('i_iter')})")
        sampler.net.load_state_dict(ckpt["state_dict"])
    sampler.net.to(device)
    if cfg.diffusion.use_fp16:
        unet.convert_to_fp16()
    sampler.eval()

    trainer = None
    if args.guidance_scale is not None:   # reference :132-148: value.pth + the trainer as the sampling driver
        v = dxmi_config.instantiate(cfg.value).to(device)
        if not args.synthetic:
            value_path = os.path.join(args.log_dir, "value.pth")
            if not os.path.exists(value_path):
                raise ValueError(f"Value ftn not found at {value_path}")
            v.load_state_dict(torch.load(value_path, map_location=device)["state_dict"])
        v.eval()
        trainer = dxmi_config.instantiate(cfg.trainer, batchsize=args.batchsize)
        trainer.set_models(v=v, sampler=sampler, optimizer=None, optimizer_v=None)

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend=_dist.dist_backend(), init_method="env://")  # RCCL

    from dxmi_hip import graph as hip_graph
    sampler.use_graph = hip_graph.default_enabled() and not args.no_graph        # second batch onwards: one hipGraphLaunch per batch
    n_batches = int(args.n_sample / args.batchsize / world)
    l_sample, i_img = [], 0
    from dxmi_hip import ops
    from utils import ImageWriter
    writer = ImageWriter()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_batches):
        if trainer is not None:
            d_sample = trainer.sample_guidance(n_sample=args.batchsize, device=device, guidance_scale=args.guidance_scale)
        else:
            d_sample = sampler.sample(args.batchsize, device=device, i_class=None, enable_grad=False)
        sample = d_sample["sample"]
        if args.skip_fid:
            # ((sample + 1) / 2).clamp(0, 1) -> save_image (:36-41) on the device, pinned copy on a side stream, PNG thread pool
            writer.submit(sample, [os.path.join(output_path, f"{local_rank}_{i_img + k}.png") for k in range(len(sample))])
            i_img += len(sample)
        else:
            l_sample.append(ops.quantize_u8(sample.contiguous().float(), mode=1, nhwc=False))      # ((x + 1) * 127.5).clamp.to(uint8), :43
    writer.close()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print0(f"generated {n_batches * args.batchsize} images/rank x {world} ranks, "
           f"{n_batches * args.batchsize / max(dt, 1e-9):.1f} images/s/rank")
    if args.skip_fid:
        return
    samples = torch.cat(l_sample)
    if world > 1:
        gathered = [torch.zeros_like(samples) for _ in range(world)]
        torch.distributed.all_gather(gathered, samples)
        samples = torch.cat(gathered)
    if local_rank == 0:
        np.savez(os.path.join(args.log_dir, f"samples_{len(samples)}.npz"), samples.permute(0, 2, 3, 1).cpu().numpy())
    if args.fid_extractor is None or args.fid_stats is None:
        print0("samples saved as npz; FID needs --fid_extractor module:attr (an InceptionV3 pool3 extractor: its weights are not in "
               "this image) and --fid_stats <dataset statistics npz>")
        return
    # reference fid() (:57-74): this rank's strided share of ALL samples -> activations -> all_gather -> statistics -> distance;
    # the statistics (np.mean / np.cov on the host in the reference) run on the device
    from pytorch_fid.fid_score import fid_from_images, load_extractor, load_statistics
    extractor = load_extractor(args.fid_extractor)
    if hasattr(extractor, "to"):
        extractor = extractor.to(device)
    m2, s2 = load_statistics(args.fid_stats)
    fid = fid_from_images(samples[local_rank::world], extractor, m2, s2, batch_size=50, dims=args.fid_dims, device=device)
    print0(f"FID from {len(samples)} samples: {fid}")


if __name__ == "__main__":
    main()
