"""hipGraph replay of the DxMI steps (dxmi_hip/graph.py) against the eager HIP path on the same seeds: the captured step must
walk the same sequence — device RNG, CPU randperm, dropout seeds, Adam step counts — and give the same numbers."""
import pytest
import torch

pytestmark = pytest.mark.gpu

UNET_KW = dict(ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=[16], dropout=0.1,
               in_channels=3, resolution=32)


def _models(T, seed=0, device="cuda:0"):
    from models.DxMI.unet_small import Model
    from models.DxMI.var_sampler import VARSampler
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    from oracle.weights import formula_tensor
    torch.manual_seed(seed)
    net = Model(**UNET_KW)
    sampler = VARSampler(net, T, [3, 32, 32], trainable_beta="fix_last")
    net.load_state_dict({k: (v if k in ("log_betas", "std") else formula_tensor(k, v.shape)) for k, v in net.state_dict().items()})
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False,
                                            out_activation="linear", avg_pool_dim=1, learn_out_scale=True, nh=128))
    v.load_state_dict({k: formula_tensor(k, t.shape) for k, t in v.state_dict().items()})
    return net, sampler.to(device), v.to(device)


def test_sampler_graph_matches_eager():
    """sample() replayed from a hipGraph draws the noise an eager call draws (torch's Philox offset advances per replay) and
    returns the same trajectory, bit for bit; a replay does not repeat the previous call's images."""
    T, B = 4, 8
    outs = {}
    for mode in ("eager", "graph"):
        _, s, _ = _models(T)
        s.eval()
        s.use_graph = mode == "graph"
        torch.cuda.manual_seed(77)
        outs[mode] = [{k: (torch.stack(v).clone() if isinstance(v, list) else v.clone())
                       for k, v in s.sample(B, device="cuda:0").items() if k in ("sample", "l_sample", "logp", "mean", "sigma")}
                      for _ in range(4)]
        if mode == "graph":
            g = next(iter(s._graphs.values()))
            assert g.replays == 2 and len(g.segments) == 1
    for i in range(4):
        for k in outs["eager"][i]:
            assert torch.equal(outs["eager"][i][k], outs["graph"][i][k]), (i, k)
    assert not torch.equal(outs["graph"][2]["sample"], outs["graph"][3]["sample"])


def _train(mode, steps, T=4, B=8, resample=False, dist_force=False, device="cuda:0", data_seed=9):
    from dxmi_hip.optim import Adam
    from models.DxMI.replay import TransitionRing
    from models.DxMI.trainer import DxMI_Trainer, append_buffer, reset_buffer
    net, sampler, v = _models(T, device=device)
    net.dropout_seed = 1234
    opt = Adam([{"params": net.log_betas, "lr": 1e-3}, {"params": [p for n, p in net.named_parameters() if "log_betas" not in n], "lr": 1e-5}])
    opt_v = Adam(v.parameters(), lr=1e-4)
    tr = DxMI_Trainer(batchsize=B, tau1=0.1, tau2=0.01, gamma=1, use_sampler_beta=True, time_cost=0, adavelreg=0.99,
                      time_cost_sig=True, n_timesteps=T, value_resample=resample)
    tr.set_models(f=None, v=v, sampler=sampler, optimizer=opt, optimizer_fstar=None, optimizer_v=opt_v)
    if dist_force:
        from dxmi_hip.dist import FlatGradSync
        tr.sync_v = FlatGradSync(v, force=True)
        tr.sync_sampler = FlatGradSync(sampler, force=True)
    tr.use_graphs = sampler.use_graph = mode == "graph"
    ring = TransitionRing(1, T, B, (3, 32, 32), device)
    torch.manual_seed(5)
    torch.cuda.manual_seed(6)
    g = torch.Generator(device=device).manual_seed(data_seed)
    logs = []
    for _ in range(steps):
        img = torch.rand(B, 3, 32, 32, device=device, generator=g) * 2 - 1
        sampler.eval()
        d = sampler.sample(B, device=device, out=ring.next_slot())
        buf = append_buffer(ring, d)
        le = tr.update_f_v(img, d, buf)
        ls = tr.update_sampler(buf, 1)
        reset_buffer(device, ring=ring)
        logs.append({**le, **ls})
    torch.cuda.synchronize()
    state = {"unet": {k: t.detach().clone() for k, t in net.state_dict().items()},
             "v": {k: t.detach().clone() for k, t in v.state_dict().items()},
             "betas_for_q": tr.betas_for_q.clone(), "opt_step": float(opt.state[net.conv_in.weight]["step"]),
             "opt_v_step": float(opt_v.state[v.net.conv1.weight]["step"]), "dropout_calls": net._dropout_calls}
    return logs, state, tr


@pytest.mark.parametrize("resample", [False, True])
def test_train_step_graph_matches_eager(resample):
    """Four DxMI train steps (sample into the ring, update_f_v, update_sampler) replayed from hipGraphs against four eager steps:
    every logged scalar, every parameter of both nets, the q-beta EMA, the optimiser step counters and the dropout call counter.
    Steps 3 and 4 are pure replays (step 1 is the eager warm-up, step 2 the capture)."""
    steps = 4
    le, se, _ = _train("eager", steps, resample=resample)
    lg, sg, tr = _train("graph", steps, resample=resample)
    graphs = tr._graphs
    assert len(graphs) == 2 and all(g.replays == steps - 2 and len(g.segments) == 1 for g in graphs.values())
    for i in range(steps):
        assert le[i].keys() == lg[i].keys()
        for k in le[i]:
            assert le[i][k] == lg[i][k] or abs(le[i][k] - lg[i][k]) <= 1e-6 * max(1.0, abs(le[i][k])), (i, k, le[i][k], lg[i][k])
    assert se["opt_step"] == sg["opt_step"] == steps and se["opt_v_step"] == sg["opt_v_step"]
    assert se["dropout_calls"] == sg["dropout_calls"]
    assert torch.equal(se["betas_for_q"], sg["betas_for_q"])
    for name in ("unet", "v"):
        for k in se[name]:
            assert torch.equal(se[name][k], sg[name][k]), (name, k, (se[name][k] - sg[name][k]).abs().max().item())


def test_graph_survives_eager_update_in_between():
    """An eager optimiser step between two replays leaves the packed weights stale on the host's books: the next replay
    refreshes them first (StepGraph checks its modules' versions), so the replayed generation uses the updated weights."""
    T, B = 4, 8
    _, s, _ = _models(T)
    s.eval()
    s.use_graph = True
    for _ in range(3):
        s.sample(B, device="cuda:0")
    net = s.net
    with torch.no_grad():
        net.conv_out.weight.mul_(0.5)          # version bump: the packs are stale now
    torch.cuda.manual_seed(3)
    a = s.sample(B, device="cuda:0")["sample"].clone()
    s.use_graph = False
    torch.cuda.manual_seed(3)
    b = s.sample(B, device="cuda:0")["sample"].clone()
    assert torch.equal(a, b)


def cut_check():
    """Body of the collective-at-a-cut test; runs inside a rank process (tests/_graph_cut_worker.py) with the nccl (= RCCL) group
    up.  At world 1 the exchange is forced through the 1-rank communicator."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    T, steps = 4, 3
    dev = f"cuda:{torch.cuda.current_device()}"
    le, se, _ = _train("eager", steps, T=T, dist_force=True, device=dev, data_seed=9 + rank)
    lg, sg, tr = _train("graph", steps, T=T, dist_force=True, device=dev, data_seed=9 + rank)
    segs = {k[0]: [kind for kind, _ in g.segments] for k, g in tr._graphs.items()}
    out = {"world": world, "f_v_eager_cuts": segs["update_f_v"].count("eager"), "f_v_graphs": segs["update_f_v"].count("graph"),
           "sampler_eager_cuts": segs["update_sampler"].count("eager"), "sampler_graphs": segs["update_sampler"].count("graph"), "T": T}
    out["logs_close"] = all(abs(le[i][k] - lg[i][k]) <= 1e-6 * max(1.0, abs(le[i][k])) for i in range(steps) for k in le[i])
    out["params_equal"] = all(torch.equal(se[n][k], sg[n][k]) for n in ("unet", "v") for k in se[n])
    chk = torch.tensor([sum(t.double().sum().item() for t in sg[n].values() if t.is_floating_point()) for n in ("unet", "v")],
                       device=dev, dtype=torch.float64)
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    out["ranks_identical"] = all(torch.equal(allc[0], c) for c in allc)
    if world > 1:
        # the EDM iteration too: MixedPrecisionTrainer exchanges the U-Net's gradients inside optimize() — one cut per policy iteration
        ee, es, _ = _edm_train("eager", 3, data_seed=9 + rank)
        eg, gs, tr2 = _edm_train("graph", 3, data_seed=9 + rank)
        segs2 = {k[0]: [kind for kind, _ in g.segments] for k, g in tr2._graphs.items()}
        out["edm_sampler_cuts"] = segs2["update_sampler_mp"].count("eager")
        out["edm_params_equal"] = all(torch.equal(es[n][k], gs[n][k]) for n in ("unet", "v") for k in es[n]) and es["lg"] == gs["lg"] and es["steps"] == gs["steps"]
        chk = torch.tensor([sum(t.double().sum().item() for t in gs[n].values() if t.is_floating_point()) for n in ("unet", "v")], device=dev, dtype=torch.float64)
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        out["edm_ranks_identical"] = all(torch.equal(allc[0], c) for c in allc)
    return out


def test_graph_cut_two_ranks_on_one_gpu_gloo():
    """The captured train step with a REAL two-rank exchange on the one GPU of the test box: two processes share cuda:0 and
    all-reduce their (different) gradients over gloo at the graph cuts.  Every rank must end the replayed steps with the parameters
    the python-issued two-rank run gives (bitwise) and with its peer's parameters."""
    import json
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DXMI_TEST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "_graph_cut_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    T = line["T"]
    assert line["world"] == 2 and line["f_v_eager_cuts"] == T + 1 and line["sampler_eager_cuts"] == 1
    assert line["logs_close"] and line["params_equal"] and line["ranks_identical"]
    assert line["edm_sampler_cuts"] == 4 and line["edm_params_equal"] and line["edm_ranks_identical"]        # T * B / B = 4 policy iterations


def test_graph_cut_runs_collective_between_segments():
    """The multi-GPU form of the captured step, at world size = visible GPUs (1 on the driver's test box: the exchange is then
    forced through a 1-rank RCCL communicator): FlatGradSync issues its all-reduces at a cut — eagerly, between two graph
    launches — so update_f_v is T + 2 graph segments around T + 1 exchanges, update_sampler 2 around 1, and the results equal
    the eager step's.  Rank processes are fresh interpreters (tests/_graph_cut_worker.py)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    world = torch.cuda.device_count()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "_graph_cut_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    T = line["T"]
    assert line["f_v_eager_cuts"] == T + 1 and line["f_v_graphs"] == T + 2
    assert line["sampler_eager_cuts"] == 1 and line["sampler_graphs"] == 2
    assert line["logs_close"] and line["params_equal"] and line["ranks_identical"]


EDM_KW = dict(image_size=32, class_cond=True, learn_sigma=False, num_channels=64, num_res_blocks=1, channel_mult="1,2",
              num_heads=4, num_head_channels=64, num_heads_upsample=-1, attention_resolutions="16", dropout=0.0,
              use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=True, use_fp16=True,
              use_new_attention_order=False, weight_schedule="uniform")
EDM_SAMPLER_KW = dict(n_timesteps=4, sample_shape=(3, 32, 32), class_cond=True, num_classes=1000, trainable_beta="fix_last",
                      stochastic_last=True, rho=4.0)
EDM_TRAINER_KW = dict(tau1=0.1, tau2=0.01, gamma=1, n_timesteps=4, use_sampler_beta=True, adavelreg=0.99, entropy_in_value=None,
                      velocity_in_value=None, value_grad_clip=True, time_cost=0, skip_sampler_tau=1, time_cost_sig=1)


def _edm_train(mode, steps, lg0=20.0, B=4, data_seed=9):
    from dxmi_hip.optim import Adam, RAdam
    from models.cm.fp16_util import MixedPrecisionTrainer
    from models.cm.script_util import create_model_and_diffusion
    from models.DxMI.openai_diffusion import OpenAIDiffusion
    from models.DxMI.replay import TransitionRing
    from models.DxMI.trainer import DxMI_Trainer_Cond, append_buffer, reset_buffer
    from models.modules import IGEBMEncoderV2
    from models.value import TimeIndependentValue
    from oracle.weights import formula_tensor
    dev = "cuda:0"
    net, diffusion = create_model_and_diffusion(**EDM_KW)
    net.load_state_dict({k: formula_tensor(k, v.shape) for k, v in net.state_dict().items()})
    sampler = OpenAIDiffusion(net, diffusion, **EDM_SAMPLER_KW)
    net.to(dev)
    v = TimeIndependentValue(IGEBMEncoderV2(in_chan=3, out_chan=1, use_spectral_norm=False, keepdim=False, out_activation="linear",
                                            avg_pool_dim=1, learn_out_scale=True, nh=128))
    v.load_state_dict({k: formula_tensor(k, t.shape) for k, t in v.state_dict().items()})
    v.to(dev)
    mp = MixedPrecisionTrainer(model=net, use_fp16=True, initial_lg_loss_scale=lg0, special_key="log_betas")
    opt = RAdam([{"params": mp.master_params[1:], "lr": 1e-6}, {"params": mp.master_params[0:1], "lr": 1e-4}])
    opt_v = Adam(v.parameters(), lr=1e-5)
    tr = DxMI_Trainer_Cond(batchsize=B, **EDM_TRAINER_KW)
    tr.set_models(v=v, sampler=sampler, optimizer=opt, optimizer_v=opt_v)
    tr.use_graphs = sampler.use_graph = mode == "graph"
    T = EDM_SAMPLER_KW["n_timesteps"]
    ring = TransitionRing(1, T, B, (3, 32, 32), dev, with_y=True, sigma_dims=1)
    torch.manual_seed(5)
    torch.cuda.manual_seed(6)
    g = torch.Generator(device=dev).manual_seed(data_seed)
    logs = []
    for _ in range(steps):
        img = torch.rand(B, 3, 32, 32, device=dev, generator=g) * 2 - 1
        y = torch.randint(0, 1000, (B,), device=dev, generator=g)
        sampler.eval()
        d = sampler.sample(B, device=dev, i_class=y, out=ring.next_slot())
        buf = append_buffer(ring, d)
        le = tr.update_f_v(img, d, buf, y=y)
        ls = tr.update_sampler_mixed_precision(buf, mp_trainer=mp)
        reset_buffer(dev, ring=ring)
        logs.append({**le, **ls})
    torch.cuda.synchronize()
    state = {"unet": {k: t.detach().clone() for k, t in net.state_dict().items()}, "v": {k: t.detach().clone() for k, t in v.state_dict().items()},
             "lg": mp.lg_loss_scale, "steps": opt.step_count(), "log": dict(mp.log), "betas_for_q": tr.betas_for_q.clone()}
    return logs, state, tr


@pytest.mark.parametrize("lg0", [20.0, 129.5])
def test_edm_train_step_graph_matches_eager(lg0):
    """DxMI_Trainer_Cond + MixedPrecisionTrainer + fused RAdam on a shrunken EDM net: the three phases of an iteration replayed
    from hipGraphs against eager iterations.  lg0 = 129.5 makes the first loss scales overflow fp32 (2^129.5, 2^128.5 = inf):
    those iterations are skipped ON THE DEVICE inside the replay — the device counter picks the rows of the host tables for
    'j overflows so far' — and the host bookkeeping after the replay must land on the eager run's loss scale, step count and
    parameters."""
    steps = 4
    le, se, _ = _edm_train("eager", steps, lg0=lg0)
    lg, sg, tr = _edm_train("graph", steps, lg0=lg0)
    assert {k[0] for k in tr._graphs} == {"update_f_v", "update_sampler_mp"}
    assert all(g.replays == steps - 2 for g in tr._graphs.values())
    assert se["lg"] == sg["lg"] and se["steps"] == sg["steps"], (se["lg"], sg["lg"], se["steps"], sg["steps"])
    if lg0 > 100:
        assert se["steps"] < steps * 4           # some iterations really overflowed
    for i in range(steps):
        for k in le[i]:
            a, b = le[i][k], lg[i][k]
            assert a == b or (a != a and b != b) or abs(a - b) <= 1e-6 * max(1.0, abs(a)), (i, k, a, b)
    assert torch.equal(se["betas_for_q"], sg["betas_for_q"])
    for name in ("unet", "v"):
        for k in se[name]:
            assert torch.equal(se[name][k], sg[name][k]), (name, k)
    for k in ("grad_norm", "param_norm", "lg_loss_scale"):
        assert se["log"].get(k) == sg["log"].get(k) or abs(se["log"][k] - sg["log"][k]) <= 1e-6 * abs(se["log"][k]), k
