"""Rank process of tests/test_hip_dist_nccl.py: one DxMI train step under the nccl (= RCCL) backend at
world size = visible GPUs, with per-rank data; prints a JSON line with parameter checksums."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "diffusion-by-maxentirl_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev)
    from dxmi_hip.dist import FlatGradSync, broadcast_parameters
    from dxmi_hip.optim import Adam
    from models.DxMI.replay import TransitionRing
    from models.DxMI.trainer import DxMI_Trainer, append_buffer
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    B, T = 4, 4
    torch.manual_seed(100 + rank)            # different initialisation per rank: the broadcast must equalise it
    net = Model(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.0, in_channels=3, resolution=32)
    sampler = VARSampler(net, T, [3, 32, 32], trainable_beta="fix_last").to(dev)
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False, out_activation="linear",
                                            avg_pool_dim=1, learn_out_scale=True, nh=128)).to(dev)
    broadcast_parameters(net)
    broadcast_parameters(v)
    not_beta = [p for n, p in net.named_parameters() if "log_betas" not in n]
    opt = Adam([{"params": net.log_betas, "lr": 1e-3}, {"params": not_beta, "lr": 1e-5}])
    opt_v = Adam(v.parameters(), lr=1e-3)
    tr = DxMI_Trainer(batchsize=B, tau1=0.1, tau2=0.01, gamma=1, use_sampler_beta=True, time_cost=0, adavelreg=0.99, time_cost_sig=1, n_timesteps=T)
    tr.set_models(f=None, v=v, sampler=sampler, optimizer=opt, optimizer_fstar=None, optimizer_v=opt_v)
    if world == 1:                           # a 1-rank group skips the exchange by default: force it through RCCL
        tr.sync_v, tr.sync_sampler = FlatGradSync(v, force=True), FlatGradSync(sampler, force=True)
    g = torch.Generator(device=dev).manual_seed(7 + rank)          # per-rank data
    img = torch.rand(B, 3, 32, 32, device=dev, generator=g) * 2 - 1
    # ---- the exchange itself against the single-process mean: value-net gradients of a per-rank loss, gathered RAW from every
    # rank (all_gather), averaged locally in rank order, vs what FlatGradSync leaves in .grad (RCCL's ring sums in another
    # order: 1e-6 relative; at world 1 the AVG over one rank must return the gradient bit for bit)
    for p in v.parameters():
        p.grad = None
    v(img, None).pow(2).sum().backward()
    raw = torch.cat([p.grad.detach().flatten() for p in v.parameters()])
    gathered = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(gathered, raw)
    want = torch.stack(gathered).sum(0) / world
    tr.sync_v()
    got = torch.cat([p.grad.detach().flatten() for p in v.parameters()])
    mean_err = ((got - want).norm() / want.norm()).item()
    mean_bitwise = bool(torch.equal(got, want))
    ranks_differ = bool(world == 1 or not torch.equal(gathered[0], gathered[-1]))     # the per-rank data really differed
    for p in v.parameters():
        p.grad = None
    ring = TransitionRing(1, T, B, (3, 32, 32), dev)
    sampler.eval()
    d = sampler.sample(B, device=dev, out=ring.next_slot())
    buf = append_buffer(ring, d)
    torch.manual_seed(5)
    le = tr.update_f_v(img, d, buf)
    ls = tr.update_sampler(buf, 1)
    torch.cuda.synchronize()
    chk = torch.tensor([sum(p.detach().double().sum().item() for p in net.parameters()),
                        sum(p.detach().double().sum().item() for p in v.parameters()),
                        sum(p.detach().double().abs().sum().item() for p in net.parameters())], device=dev, dtype=torch.float64)
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    same = all(torch.equal(allc[0], c) for c in allc)
    fin = all(x == x for x in list(le.values()) + list(ls.values()))
    if rank == 0:
        print(json.dumps({"world": world, "rank_identical_parameters": same if world > 1 else None, "finite": fin, "backend": dist.get_backend(),
                          "mean_rel_err": mean_err, "mean_bitwise": mean_bitwise, "ranks_differ": ranks_differ,
                          "nccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()),
                          "flat_bytes": [(tr.sync_v.flat.numel() - tr.sync_v.nflags) * 4, (tr.sync_sampler.flat.numel() - tr.sync_sampler.nflags) * 4],
                          "flags": [tr.sync_v.nflags, tr.sync_sampler.nflags],
                          "raw_bytes": [sum(p.numel() for p in s_.params) * 4 for s_ in (tr.sync_v, tr.sync_sampler)],
                          "padded_bytes": [sum((p.numel() + 3) // 4 * 4 for p in s_.params) * 4 for s_ in (tr.sync_v, tr.sync_sampler)],
                          "slices_16B_aligned": all(v.data_ptr() % 16 == 0 for s_ in (tr.sync_v, tr.sync_sampler) for v in s_.views),
                          "grads_alias_flat": all(p.grad is None or p.grad.data_ptr() == v.data_ptr()
                                                  for s_ in (tr.sync_v, tr.sync_sampler) for p, v in zip(s_.params, s_.views))}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
