"""N > 1 path on CPU: world_size-2 gloo processes exercise the flat gradient all-reduce and the
parameter broadcast that replace the reference's DDP wrappers."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dxmi_hip.dist import FlatGradSync, broadcast_parameters
    torch.manual_seed(100 + rank)                      # different init per rank
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
    m.register_buffer("buf", torch.full((3,), float(rank)))
    broadcast_parameters(m, src=0)
    w0 = m[0].weight.detach().clone()
    x = torch.full((4, 8), float(rank + 1))
    m(x).sum().backward()
    local = [p.grad.clone() for p in m.parameters()]
    FlatGradSync(m)()
    # numpy copies travel by value; tensors would travel as shared-memory handles that die with this process
    q.put((rank, w0.numpy(), m.buf.clone().numpy(), [g.numpy() for g in local], [p.grad.clone().numpy() for p in m.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_and_broadcast_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:       # a port the kernel says is free right now (a pid-derived one collided once)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = [(r, torch.from_numpy(w), torch.from_numpy(b), [torch.from_numpy(x) for x in lo], [torch.from_numpy(x) for x in sy])
           for r, w, b, lo, sy in res]
    (_, w_a, buf_a, loc_a, syn_a), (_, w_b, buf_b, loc_b, syn_b) = res
    assert torch.equal(w_a, w_b) and torch.equal(buf_a, buf_b) and float(buf_b[0]) == 0.0   # broadcast from rank 0
    for la, lb, sa, sb in zip(loc_a, loc_b, syn_a, syn_b):
        assert torch.allclose(sa, (la + lb) / 2, atol=1e-6) and torch.equal(sa, sb)          # mean over ranks, identical


def test_single_process_is_noop():
    sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
    from dxmi_hip.dist import FlatGradSync, is_distributed
    assert not is_distributed()
    m = torch.nn.Linear(3, 2)
    m(torch.ones(1, 3)).sum().backward()
    g = m.weight.grad.clone()
    FlatGradSync(m)()
    assert torch.equal(m.weight.grad, g)


def _mp_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dxmi_hip.dist import broadcast_parameters
    from models.cm.fp16_util import MixedPrecisionTrainer
    torch.manual_seed(7 + rank)
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.SiLU(), torch.nn.Linear(5, 2))
    m.register_parameter("log_betas", torch.nn.Parameter(torch.zeros(3)))
    broadcast_parameters(m, src=0)
    mp_tr = MixedPrecisionTrainer(model=m, use_fp16=True, initial_lg_loss_scale=8, special_key="log_betas")
    opt = torch.optim.SGD(mp_tr.master_params, lr=0.5)
    x = torch.full((3, 6), float(rank + 1))
    mp_tr.zero_grad()
    mp_tr.backward(m(x).sum() * (rank + 1) + m.log_betas.sum() * (rank + 1))
    ok = mp_tr.optimize(opt)          # all-reduces the model gradients (mean over ranks) before the master step
    q.put((rank, ok, [p.detach().tolist() for p in m.parameters()], mp_tr.lg_loss_scale))   # plain lists: no shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


def test_mixed_precision_trainer_syncs_gradients_gloo():
    """EDM sampler update on 2 ranks: identical parameters after one optimise() although the ranks saw different data."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, ok_a, pa, ls_a), (_, ok_b, pb, ls_b) = res
    assert ok_a and ok_b and ls_a == ls_b
    assert pa == pb
    assert torch.allclose(torch.tensor(pa[0]), torch.full((3,), -0.5 * 1.5))      # log_betas (root parameter, listed first): mean gradient (1 + 2) / 2, lr 0.5


def _ragged_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dxmi_hip.dist import FlatGradSync, broadcast_parameters
    torch.manual_seed(3)
    m = torch.nn.ModuleDict({"a": torch.nn.Linear(4, 4), "b": torch.nn.Linear(4, 4), "unused": torch.nn.Linear(4, 4)})
    x = torch.ones(2, 4)
    # branch "unused" gets a gradient on NO rank: its .grad must stay None (the optimiser then skips it as torch does)
    # rank 1 never uses branch b: its parameters have grad None there (the collective must not shrink)
    (m["a"](x).sum() + (m["b"](x).sum() if rank == 0 else 0.0)).backward()
    want_b = m["b"].weight.grad.clone() / world if rank == 0 else None
    sync = FlatGradSync(m)
    sync()
    assert m["unused"].weight.grad is None and m["unused"].bias.grad is None
    assert m["b"].weight.grad is not None             # produced on rank 0 only: every rank receives the mean
    ptr = sync.flat.data_ptr()
    sync()                                            # persistent buffer: no reallocation on the second exchange
    assert m["unused"].weight.grad is None
    # broadcast AFTER a version-keyed cache was filled must be visible through the version counter
    v_before = m["a"].weight._version
    broadcast_parameters(m, src=0)
    q.put((rank, m["b"].weight.grad.clone().numpy(), None if want_b is None else want_b.numpy(), sync.flat.data_ptr() == ptr,
           m["a"].weight._version > v_before))
    dist.barrier()
    dist.destroy_process_group()


def test_fixed_parameter_set_when_a_rank_has_no_gradient_gloo():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, gb0, want, same0, bumped0), (_, gb1, _, same1, bumped1) = res
    # second sync averages the already-averaged gradient again: (g/2 + g/2)/2 on both ranks = g/2
    assert torch.allclose(torch.from_numpy(gb0), torch.from_numpy(want)) and (gb0 == gb1).all()
    assert same0 and same1 and bumped1            # rank 0 is the source: nothing is written there


def _overlap_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dxmi_hip.dist import FlatGradSync
    torch.manual_seed(5)
    m = torch.nn.ModuleDict({"a": torch.nn.Linear(6, 6), "b": torch.nn.Linear(6, 6), "c": torch.nn.Linear(6, 2)})
    sync = FlatGradSync(m, bucket_mb=4e-5)            # ~10 elements per bucket: three buckets, launched DURING backward
    n_buckets = len(sync.buckets)
    x = torch.full((3, 6), float(rank + 1))

    def loss():
        h = m["a"](x)
        if rank == 0:                                 # rank 1 never uses branch b: its bucket completes only inside sync()
            h = h + m["b"](x)
        return m["c"](h).pow(2).sum()

    out = {}
    # (1) hooks + buckets
    loss().backward()
    launched_in_backward = sync.next
    local = {k: (None if p.grad is None else p.grad.clone()) for k, p in m.named_parameters()}
    sync()
    out["round1"] = {k: p.grad.clone().numpy() for k, p in m.named_parameters()}
    out["local1"] = {k: (None if g is None else g.numpy()) for k, g in local.items()}
    # (2) second exchange reuses the buffer and the state machine
    ptr = sync.flat.data_ptr()
    for p in m.parameters():
        p.grad = None
    loss().backward()
    sync()
    out["same_buffer"] = sync.flat.data_ptr() == ptr
    out["round2_equal"] = all(torch.equal(p.grad, torch.from_numpy(out["round1"][k])) for k, p in m.named_parameters())
    # (3) two backward passes before one sync: blocking fallback on the ACCUMULATED gradients
    for p in m.parameters():
        p.grad = None
    loss().backward()
    loss().backward()
    acc = {k: (None if p.grad is None else p.grad.clone().numpy()) for k, p in m.named_parameters()}
    sync()
    out["round3"] = {k: p.grad.clone().numpy() for k, p in m.named_parameters()}
    out["acc3"] = acc
    q.put((rank, n_buckets, launched_in_backward, out))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_overlapped_sync_gloo():
    """FlatGradSync built BEFORE the backward: gradients are packed by hooks, buckets all-reduced in index order while autograd is
    still running; an unused branch on one rank, buffer reuse and the two-backward fallback."""
    import socket
    import numpy as np
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, nb0, lib0, o0), (_, nb1, lib1, o1) = res
    assert nb0 == nb1 and nb0 >= 3
    assert lib0 >= 1                                   # rank 0 had complete buckets before backward returned
    zero = lambda g, like: np.zeros_like(like) if g is None else g
    for k in o0["round1"]:
        want = (zero(o0["local1"][k], o0["round1"][k]) + zero(o1["local1"][k], o0["round1"][k])) / 2
        assert np.allclose(o0["round1"][k], want, atol=1e-6) and np.array_equal(o0["round1"][k], o1["round1"][k]), k
        want3 = (zero(o0["acc3"][k], o0["round3"][k]) + zero(o1["acc3"][k], o0["round3"][k])) / 2
        assert np.allclose(o0["round3"][k], want3, atol=1e-6) and np.array_equal(o0["round3"][k], o1["round3"][k]), k
    assert o0["same_buffer"] and o1["same_buffer"] and o0["round2_equal"] and o1["round2_equal"]
