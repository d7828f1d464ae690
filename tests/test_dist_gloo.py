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


# ---- round 4: the train step's whole collective SEQUENCE at world size 4 ---------------------------------------------
def _sequence_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dxmi_hip.dist as dd
    log = []
    real_all_reduce = dist.all_reduce

    def logged(t, *a, **k):             # every collective this rank issues, in issue order
        log.append((int(t.numel()), str(k.get("op", a[0] if a else "SUM")).split(".")[-1], bool(k.get("async_op", False))))
        return real_all_reduce(t, *a, **k)
    dd.dist.all_reduce = logged
    T = 3
    torch.manual_seed(11)               # same initial parameters everywhere (what broadcast_parameters establishes)
    value = torch.nn.Sequential(torch.nn.Linear(12, 24), torch.nn.LeakyReLU(0.2), torch.nn.Linear(24, 1))
    unet = torch.nn.ModuleDict({"down": torch.nn.Linear(12, 32), "mid": torch.nn.Linear(32, 32), "attn": torch.nn.Linear(32, 32),
                                "up": torch.nn.Linear(32, 12)})
    unet.register_parameter("log_betas", torch.nn.Parameter(torch.zeros(T)))
    never = torch.nn.Linear(4, 4)       # a module whose parameters get no gradient on ANY rank (empty foreach lists)
    sync_v, sync_u, sync_n = dd.FlatGradSync(value), dd.FlatGradSync(unet, bucket_mb=2e-3), dd.FlatGradSync(never)
    g = torch.Generator().manual_seed(1000 + rank)      # rank-local data, as the rank-local replay ring
    x = torch.randn(8, 12, generator=g)
    local = {"value": [], "unet": None}
    synced = {"value": [], "unet": None}
    # update_f_v: one energy step + T TD steps, each: backward through the value net, all-reduce, optimiser step (trainer.py:230-346)
    for step in range(T + 1):
        for p in value.parameters():
            p.grad = None
        (value(x * (step + 1)).pow(2).mean()).backward()
        local["value"].append([p.grad.clone().numpy() for p in value.parameters()])
        sync_v()
        # fp32 wire: the means are not copied back, .grad IS the parameter's slice of the flat buffer
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync_v.params, sync_v.views))
        synced["value"].append([p.grad.clone().numpy() for p in value.parameters()])
    # update_sampler: one backward through value net INTO the U-Net; rank 2 does not touch the attention branch (missing gradients)
    h = unet["mid"](torch.tanh(unet["down"](x)))
    if rank != 2:
        h = h + unet["attn"](h)
    out = unet["up"](h) * torch.exp(unet.log_betas).sum()
    for p in value.parameters():        # the policy step freezes the value net (trainer.py `_Frozen`): its hooks must not fire
        p.requires_grad_(False)
    value(out).sum().backward()
    for p in value.parameters():
        p.requires_grad_(True)
    launched_during_backward = sync_u.next
    local["unet"] = {k: (None if p.grad is None else p.grad.clone().numpy()) for k, p in unet.named_parameters()}
    sync_u()
    synced["unet"] = {k: p.grad.clone().numpy() for k, p in unet.named_parameters()}
    sync_n()                            # nothing to exchange, nothing raised, .grad stays None
    assert all(p.grad is None for p in never.parameters())
    q.put((rank, log, local, synced, len(sync_u.buckets), launched_during_backward))
    dist.barrier()
    dist.destroy_process_group()


def test_train_step_collective_sequence_world4_gloo():
    """The DxMI train step's exchange pattern (train_cifar10.py:298-309 + trainer.py:230-408) on 4 ranks: T+1 value-net
    all-reduces, then the U-Net's bucketed reduce with one rank missing a branch's gradients, then a module nobody touched.
    Every rank must issue the SAME collective sequence (sizes, ops, order) and end with the single-process mean of the per-rank
    gradients, bit-identical across ranks."""
    import socket
    import numpy as np
    world = 4
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sequence_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    logs = [r[1] for r in res]
    for r in range(1, world):
        assert [(n, op) for n, op, _ in logs[r]] == [(n, op) for n, op, _ in logs[0]], f"rank {r} issued a different collective sequence"
    nb = res[0][4]
    assert nb >= 3 and all(r[4] == nb for r in res)
    # sequence (round 5: the "used on any rank" flags ride in the tail of a module's LAST bucket, no collective of their own):
    #   (T+1) x [value net: one bucket = gradients + 4 flags]  +  nb U-Net buckets  +  1 bucket of the untouched module
    # collectives per train step = (T + 1) + n_buckets, exactly (+ 1 here for the module nobody touched)
    T = 3
    pad4 = lambda n: (n + 3) // 4 * 4          # every parameter's slice of the flat buffer starts on a 16-byte boundary (fp32 wire)
    nv = sum(pad4(p.numel()) for p in torch.nn.Sequential(torch.nn.Linear(12, 24), torch.nn.LeakyReLU(0.2), torch.nn.Linear(24, 1)).parameters())
    sizes = [n for n, _, _ in logs[0]]
    assert sizes[:T + 1] == [nv + 4] * (T + 1)
    assert len(sizes) == (T + 1) + nb + 1
    nu = 12 * 32 + 32 + 2 * (32 * 32 + 32) + 32 * 12 + 12 + pad4(T)
    assert sum(sizes[T + 1:T + 1 + nb]) == nu + 9          # the U-Net's buckets: every gradient once + 9 flags in the last one
    assert sizes[-1] == 4 * 4 + 4 + 2
    assert res[0][5] >= 1 and res[2][5] < nb           # buckets launched DURING backward; rank 2 holds one back until sync()
    for step in range(T + 1):
        for i in range(4):
            want = sum(res[r][2]["value"][step][i] for r in range(world)) / world
            for r in range(world):
                assert np.allclose(res[r][3]["value"][step][i], want, atol=1e-6)
                assert np.array_equal(res[r][3]["value"][step][i], res[0][3]["value"][step][i])
    for k in res[0][3]["unet"]:
        like = res[0][3]["unet"][k]
        want = sum((np.zeros_like(like) if res[r][2]["unet"][k] is None else res[r][2]["unet"][k]) for r in range(world)) / world
        for r in range(world):
            assert np.allclose(res[r][3]["unet"][k], want, atol=1e-6), k
            assert np.array_equal(res[r][3]["unet"][k], like), k
    assert res[2][2]["unet"]["attn.weight"] is None    # the missing gradients really were missing on rank 2


def _captured_sync_worker(rank, world, port, q):
    """FlatGradSync's exchange of a step that is being captured into hipGraphs (dist.py `_sync_captured`), with the graph replaced
    by a stand-in whose `cut(fn)` just runs `fn` (there are no graphs on the CPU): the pack / collectives-at-the-cut / re-bind
    sequence must give what the eager, overlapped sync gives, with one collective per bucket in bucket order."""
    sys.path.insert(0, os.path.join(ROOT, "diffusion-by-maxentirl_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dxmi_hip import graph as hip_graph
    from dxmi_hip.dist import FlatGradSync
    torch.manual_seed(3)
    m = torch.nn.Sequential(torch.nn.Linear(8, 300), torch.nn.ReLU(), torch.nn.Linear(300, 7), torch.nn.ReLU(), torch.nn.Linear(7, 1))
    fs = FlatGradSync(m, bucket_mb=0.004)          # ~1 k elements per bucket: several buckets
    x = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + rank))
    m(x).sum().backward()
    local = [p.grad.clone() for p in m.parameters()]
    fs()                                            # eager, overlapped: allocates the flat buffer; reference result
    eager = [p.grad.clone() for p in m.parameters()]
    for p in m.parameters():
        p.grad = None

    class Cap:
        cuts = 0

        def cut(self, fn):
            Cap.cuts += 1
            prev, hip_graph._CURRENT = hip_graph._CURRENT, None      # the cut's function runs un-captured
            try:
                fn()
            finally:
                hip_graph._CURRENT = prev
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (calls.append(int(t.numel())), orig(t, *a, **k))[1]
    cap = Cap()
    hip_graph._CURRENT = cap
    try:
        m(x).sum().backward()                      # hooks must stay quiet while capturing
        assert not calls
        fs()
    finally:
        hip_graph._CURRENT = None
        dist.all_reduce = orig
    got = [p.grad.clone() for p in m.parameters()]
    aliased = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(fs.params, fs.views))
    q.put((rank, [g.numpy() for g in local], [g.numpy() for g in eager], [g.numpy() for g in got], calls, Cap.cuts,
           [b[1] - b[0] for b in fs.buckets], aliased))
    dist.barrier()
    dist.destroy_process_group()


def test_captured_sync_matches_eager_sync_gloo():
    import socket
    import numpy as np
    world = 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_captured_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        _, local, eager, got, calls, cuts, bucket_sizes, aliased = res[r]
        assert cuts == 1 and calls == bucket_sizes and len(bucket_sizes) >= 2 and aliased
        for i in range(len(got)):
            want = sum(res[k][1][i] for k in range(world)) / world
            assert np.allclose(got[i], want, atol=1e-6)
            assert np.array_equal(got[i], eager[i]) and np.array_equal(got[i], res[0][3][i])
