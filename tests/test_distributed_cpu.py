"""world_size-2 data-parallel path on CPU (gloo), with oracle/libskd_ref.so as the C-ABI double:
cross-rank InPlaceABNSync statistics, bucketed gradient averaging, and a whole NetModel step whose
result must equal the reference's DP semantics (full-batch BN statistics, per-shard losses, mean of
the shard losses -- utils/parallel.py:155, libs/functions.py:185-209,263-280)."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import cref
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    _lib.install_test_backend(cref.load(_lib.SIGNATURES))
    P.init_distributed("gloo")
    try:
        out = globals()[fn_name](rank, world)
        torch.save(out, os.path.join(outdir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), fn_name, d), nprocs=world, join=True)
        return [torch.load(os.path.join(d, "r%d.pt" % r)) for r in range(world)]


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ---------------------------------------------------------------------------------------------------
def _sync_abn(rank, world):
    from structure_knowledge_distillation_amd import libs
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, 5, 3, generator=g) * 2 + 1
    gz = torch.randn(4, 6, 5, 3, generator=g)
    w, b = torch.randn(6, generator=g), torch.randn(6, generator=g)
    from structure_knowledge_distillation_amd.utils import parallel as P
    mod = libs.InPlaceABNSync(6, activation="leaky_relu").train()
    with torch.no_grad():
        mod.weight.copy_(w); mod.bias.copy_(b)
    xs = x[rank * 2:(rank + 1) * 2].clone().requires_grad_(True)
    P.comm_timer.enable()
    z = mod(xs * 1.0)
    (z * gz[rank * 2:(rank + 1) * 2]).sum().backward()
    spans = P.comm_timer.disable()
    assert spans["syncabn"][1] == 2 and spans["syncabn"][0] >= 0.0      # one exchange forward, one backward (bench.py "comm")
    return {"z": z.detach(), "dx": xs.grad, "dw": mod.weight.grad, "db": mod.bias.grad,
            "rm": mod.running_mean.clone(), "rv": mod.running_var.clone()}


def test_sync_abn_two_ranks_equals_one_rank_on_the_whole_batch():
    from oracle import abn_torch
    outs = _run("_sync_abn")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, 5, 3, generator=g) * 2 + 1
    gz = torch.randn(4, 6, 5, 3, generator=g)
    w, b = torch.randn(6, generator=g), torch.randn(6, generator=g)
    xo = x.double().requires_grad_(True)
    wo, bo = w.double().requires_grad_(True), b.double().requires_grad_(True)
    rm, rv = torch.zeros(6, dtype=torch.float64), torch.ones(6, dtype=torch.float64)
    zo = abn_torch.abn_autograd(xo, wo, bo, rm, rv, True, 0.1, 1e-5, "leaky_relu", 0.01)
    (zo * gz.double()).sum().backward()
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        assert rel(outs[r]["z"], zo[sl]) < 1e-5
        assert rel(outs[r]["dx"], xo.grad[sl]) < 1e-4
        assert rel(outs[r]["rm"], rm) < 1e-6 and rel(outs[r]["rv"], rv) < 1e-6   # n = N*S*world (functions.py:177,209)
    # parameter gradients: the rank average (what the gradient all-reduce produces) is half the whole-batch gradient
    assert rel(0.5 * (outs[0]["dw"] + outs[1]["dw"]), 0.5 * wo.grad) < 1e-4
    assert rel(0.5 * (outs[0]["db"] + outs[1]["db"]), 0.5 * bo.grad) < 1e-4


# ---------------------------------------------------------------------------------------------------
def _sync_stem(rank, world):
    """Round 6: the fused training stem (bn3 -> relu3 -> maxpool, libs.modules.forward_relu_maxpool) with cross-replica statistics:
    statistics -> exchange -> normalise-rectify-pool; gather-reduce -> exchange -> gather-dx."""
    from structure_knowledge_distillation_amd import libs
    g = torch.Generator().manual_seed(3)
    n = 2 * world
    x = (torch.randn(n, 8, 9, 11, generator=g) * 2 + 0.5)
    gz = torch.randn(n, 8, 5, 6, generator=g)
    w, b = torch.randn(8, generator=g), torch.randn(8, generator=g)
    sl = slice(2 * rank, 2 * rank + 2)
    pool = torch.nn.MaxPool2d(3, 2, 1, ceil_mode=True)
    mod = libs.InPlaceABNSync(8, activation="none").train()
    with torch.no_grad():
        mod.weight.copy_(w); mod.bias.copy_(b)
    xs = x[sl].contiguous(memory_format=torch.channels_last).requires_grad_(True)
    z = mod.forward_relu_maxpool(xs * 1.0, pool)
    (z * gz[sl]).sum().backward()
    return {"z": z.detach().contiguous(), "dx": xs.grad.contiguous(), "dw": mod.weight.grad, "db": mod.bias.grad,
            "rm": mod.running_mean.clone(), "rv": mod.running_var.clone()}


def test_fused_stem_two_ranks_equals_one_rank_on_the_whole_batch():
    from oracle import abn_torch
    outs = _run("_sync_stem")
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4, 8, 9, 11, generator=g) * 2 + 0.5)
    gz = torch.randn(4, 8, 5, 6, generator=g)
    w, b = torch.randn(8, generator=g), torch.randn(8, generator=g)
    xo = x.double().requires_grad_(True)
    wo, bo = w.double().requires_grad_(True), b.double().requires_grad_(True)
    rm, rv = torch.zeros(8, dtype=torch.float64), torch.ones(8, dtype=torch.float64)
    yo = torch.relu(abn_torch.abn_autograd(xo, wo, bo, rm, rv, True, 0.1, 1e-5, "none", 0.01))        # pspnet_combine.py:176-180
    zo = torch.nn.functional.max_pool2d(yo, 3, 2, 1, ceil_mode=True)
    (zo * gz.double()).sum().backward()
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        assert rel(outs[r]["z"], zo[sl]) < 1e-5 and rel(outs[r]["dx"], xo.grad[sl]) < 1e-4
        assert rel(outs[r]["rm"], rm) < 1e-6 and rel(outs[r]["rv"], rv) < 1e-6        # pooled n = N * S * world
    assert rel(0.5 * (outs[0]["dw"] + outs[1]["dw"]), 0.5 * wo.grad) < 1e-4
    assert rel(0.5 * (outs[0]["db"] + outs[1]["db"]), 0.5 * bo.grad) < 1e-4


# ---------------------------------------------------------------------------------------------------
def _sync_abn_nhwc(rank, world):
    """Channels-last tensors take the ONE-CALL synchronised entries (skd_abn_forward_train_nhwc_sync / skd_abn_backward_nhwc_sync /
    skd_abn_relu_backward_nhwc_sync, include/skd.h section 12) whenever the group has mailboxes: in-place leaky form, fused
    BN + ReLU with and without residual."""
    from structure_knowledge_distillation_amd import libs
    from structure_knowledge_distillation_amd.utils import parallel as P
    g = torch.Generator().manual_seed(0)
    n = 2 * world
    x = torch.randn(n, 8, 5, 3, generator=g) * 2 + 1
    r = torch.randn(n, 8, 5, 3, generator=g)
    gz = torch.randn(n, 8, 5, 3, generator=g)
    w, b = torch.randn(8, generator=g), torch.randn(8, generator=g)
    sl = slice(2 * rank, 2 * rank + 2)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    assert P.SyncMailbox.get(dist.group.WORLD, torch.device("cpu")) is not None
    out = {}
    for kind in ("leaky", "relu", "relu_res"):
        mod = libs.InPlaceABNSync(8, activation="leaky_relu" if kind == "leaky" else "none").train()
        with torch.no_grad():
            mod.weight.copy_(w); mod.bias.copy_(b)
        xs = x[sl].clone().requires_grad_(True)
        rs = r[sl].clone().requires_grad_(True)
        P.comm_timer.enable()
        if kind == "leaky":
            z = mod(cl(xs * 1.0))
        else:
            z = mod.forward_relu(cl(xs * 1.0), cl(rs * 1.0) if kind == "relu_res" else None)
        (z * gz[sl]).sum().backward()
        spans = P.comm_timer.disable()
        assert spans["syncabn_fused"][1] == 2 and "syncabn" not in spans     # one call forward, one backward, exchange inside
        out[kind] = {"z": z.detach().contiguous(), "dx": xs.grad, "dr": rs.grad, "dw": mod.weight.grad, "db": mod.bias.grad,
                     "rm": mod.running_mean.clone(), "rv": mod.running_var.clone()}
    return out


@pytest.mark.parametrize("world", [2, 8])
def test_sync_abn_channels_last_one_call_entries(world):
    from oracle import abn_torch
    outs = _run("_sync_abn_nhwc", world)
    g = torch.Generator().manual_seed(0)
    n = 2 * world
    x = torch.randn(n, 8, 5, 3, generator=g) * 2 + 1
    r = torch.randn(n, 8, 5, 3, generator=g)
    gz = torch.randn(n, 8, 5, 3, generator=g)
    w, b = torch.randn(8, generator=g), torch.randn(8, generator=g)
    for kind in ("leaky", "relu", "relu_res"):
        xo, ro = x.double().requires_grad_(True), r.double().requires_grad_(True)
        wo, bo = w.double().requires_grad_(True), b.double().requires_grad_(True)
        rm, rv = torch.zeros(8, dtype=torch.float64), torch.ones(8, dtype=torch.float64)
        zo = abn_torch.abn_autograd(xo, wo, bo, rm, rv, True, 0.1, 1e-5, "leaky_relu" if kind == "leaky" else "none", 0.01)
        if kind != "leaky":
            zo = torch.relu(zo + ro if kind == "relu_res" else zo)
        (zo * gz.double()).sum().backward()
        for rk in range(world):
            sl = slice(2 * rk, 2 * rk + 2)
            o = outs[rk][kind]
            assert rel(o["z"], zo[sl]) < 1e-5, (kind, rk)
            assert rel(o["dx"], xo.grad[sl]) < 1e-4, (kind, rk)
            if kind == "relu_res":
                assert rel(o["dr"], ro.grad[sl]) < 1e-5, (kind, rk)
            assert rel(o["rm"], rm) < 1e-6 and rel(o["rv"], rv) < 1e-6, (kind, rk)       # n = N * S * world
            assert torch.equal(o["rm"], outs[0][kind]["rm"]) and torch.equal(o["rv"], outs[0][kind]["rv"])   # replicas agree bit for bit
        # parameter gradients: the mean over ranks (what the gradient all-reduce produces) is 1 / world of the whole-batch gradient
        assert rel(sum(outs[rk][kind]["dw"] for rk in range(world)), wo.grad) < 1e-4, kind
        assert rel(sum(outs[rk][kind]["db"] for rk in range(world)), bo.grad) < 1e-4, kind


# ---------------------------------------------------------------------------------------------------
def _mailbox_vs_collectives(rank, world):
    """The one-hop mailbox exchange (include/skd.h section 12; here oracle/sync_ref.c over POSIX shared memory) against the
    torch.distributed collectives it replaces, on the same data, bit for bit: forward statistics (+ running update) and
    backward statistics, with and without per-rank sample weights, several exchanges in a row (parity alternation)."""
    import importlib
    IA = importlib.import_module("structure_knowledge_distillation_amd.libs.inplace_abn")
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    lib, group = _lib.get(), dist.group.WORLD
    out = {"rounds": []}
    for rnd in range(5):
        C = (8, 64, 512, 2048, 12)[rnd]
        weighted = rnd % 2 == 1
        if weighted:
            P.set_replica_batch(3 if rank == 0 else 1, "cpu")
        else:
            P.clear_replica_batch()
        res = {}
        for mode in ("1", "0"):
            os.environ["SKD_SYNC_IPC"] = mode
            P.SyncMailbox.reset()
            mb = P.SyncMailbox.get(group, torch.device("cpu"))
            assert (mb is not None) == (mode == "1")
            stat = torch.randn(2, C, generator=torch.Generator().manual_seed(100 * rnd + rank)).abs() + 0.1
            rm, rv = torch.zeros(C), torch.ones(C)
            count = 7 * (3 if rank == 0 else 1) if weighted else 7          # per-rank sample count, consistent with the weights
            mean, var = IA._sync_stats(stat.clone(), C, count, group, rm, rv, 0.1, lib, None)
            gstat = torch.randn(2, C, generator=torch.Generator().manual_seed(7000 + 100 * rnd + rank))
            for _ in range(3):                       # consecutive exchanges: parity 0 / 1 / 0
                IA._sync_grad_stats(gstat, group)
            res[mode] = (mean.clone(), var.clone(), rm, rv, gstat)
        for a, b in zip(res["1"], res["0"]):
            if world == 2:          # two addends: any summation order gives the same bits
                assert torch.equal(a, b), "mailbox exchange differs from the collectives (round %d)" % rnd
            else:                   # the mailbox kernels add in rank order, a ring all-reduce in ring order
                assert rel(a, b) < 1e-6, "mailbox exchange differs from the collectives (round %d)" % rnd
        out["rounds"].append(res["1"])
    os.environ["SKD_SYNC_IPC"] = "1"
    P.SyncMailbox.reset()
    return out


def _mailbox_fallback(rank, world):
    """One rank cannot open its peer's mailbox: the WHOLE group must fall back to the collectives, quickly (no rank left
    spinning in the self-test), and the synchronised statistics must still be right."""
    import importlib
    import time
    IA = importlib.import_module("structure_knowledge_distillation_amd.libs.inplace_abn")
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    real = _lib.get()
    if rank == 1:
        class Broken:
            def __getattr__(self, name):
                if name == "skd_sync_connect":
                    return lambda ctx, handles: 0
                return getattr(real, name)
        _lib.install_test_backend(Broken())
    t0 = time.perf_counter()
    mb = P.SyncMailbox.get(dist.group.WORLD, torch.device("cpu"))
    took = time.perf_counter() - t0
    _lib.install_test_backend(real)
    stat = torch.tensor([[1.0 + rank, 2.0], [0.5, 0.25 * (rank + 1)]])
    rm, rv = torch.zeros(2), torch.ones(2)
    mean, var = IA._sync_stats(stat.clone(), 2, 10, dist.group.WORLD, rm, rv, 0.1, real, None)
    P.SyncMailbox.reset()
    return {"mailbox": mb is not None, "took": took, "mean": mean.clone(), "var": var.clone()}


def test_mailbox_setup_failure_on_one_rank_sends_the_group_to_the_collectives():
    outs = _run("_mailbox_fallback")
    assert not outs[0]["mailbox"] and not outs[1]["mailbox"]
    assert max(o["took"] for o in outs) < 4.0, "no rank may sit in the 5 s self-test time-outs"
    for o in outs:                                   # functions.py:196-197 on the two replicas' [mean, var]
        assert torch.allclose(o["mean"], torch.tensor([1.5, 2.0])) and torch.allclose(o["var"], torch.tensor([0.75, 0.375]))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_mailbox_exchange_is_bit_identical_to_the_collectives(world):
    outs = _run("_mailbox_vs_collectives", world)
    for r in range(1, world):
        for a, b in zip(outs[0]["rounds"], outs[r]["rounds"]):
            for ta, tb in zip(a, b):
                assert torch.equal(ta, tb), "replicas must hold identical pooled statistics"


# ---------------------------------------------------------------------------------------------------
def _sync_abn_unequal(rank, world):
    """Rank 0 holds 3 samples, rank 1 holds 1 (a short last batch): pooled statistics through the per-rank weights."""
    from structure_knowledge_distillation_amd import libs
    from structure_knowledge_distillation_amd.utils import parallel as P
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, 5, 3, generator=g) * 2 + 1
    gz = torch.randn(4, 6, 5, 3, generator=g)
    sl = slice(0, 3) if rank == 0 else slice(3, 4)
    mod = libs.InPlaceABNSync(6, activation="leaky_relu").train()
    P.set_replica_batch(sl.stop - sl.start, torch.device("cpu"))
    xs = x[sl].clone().requires_grad_(True)
    z = mod(xs * 1.0)
    (z * gz[sl]).sum().backward()
    P.set_replica_batch(2, torch.device("cpu"))            # equal shards again: weights become uniform
    w_eq = P.replica_weights().clone()
    return {"z": z.detach(), "dx": xs.grad, "rm": mod.running_mean.clone(), "rv": mod.running_var.clone(), "w_eq": w_eq}


def test_sync_abn_unequal_shards_pool_by_sample_count():
    from oracle import abn_torch
    outs = _run("_sync_abn_unequal")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, 5, 3, generator=g) * 2 + 1
    gz = torch.randn(4, 6, 5, 3, generator=g)
    xo = x.double().requires_grad_(True)
    wo, bo = torch.ones(6, dtype=torch.float64), torch.zeros(6, dtype=torch.float64)
    rm, rv = torch.zeros(6, dtype=torch.float64), torch.ones(6, dtype=torch.float64)
    zo = abn_torch.abn_autograd(xo, wo, bo, rm, rv, True, 0.1, 1e-5, "leaky_relu", 0.01)
    (zo * gz.double()).sum().backward()
    for r, sl in ((0, slice(0, 3)), (1, slice(3, 4))):
        assert rel(outs[r]["z"], zo[sl]) < 1e-5, "forward must use the statistics of the WHOLE batch"
        assert rel(outs[r]["dx"], xo.grad[sl]) < 1e-4
        assert rel(outs[r]["rm"], rm) < 1e-6 and rel(outs[r]["rv"], rv) < 1e-6      # pooled n = 4 * 15
        assert torch.allclose(outs[r]["w_eq"], torch.full((2,), 0.5))


def test_per_rank_batch_helper():
    from structure_knowledge_distillation_amd.utils import parallel as P
    assert P.per_rank_batch(8) == 8 and P.replica_weights() is None        # single process: the global batch


# ---------------------------------------------------------------------------------------------------
def _reducer(rank, world):
    from structure_knowledge_distillation_amd.utils.parallel import GradientAllReducer
    torch.manual_seed(1)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (5, 300, 7, 1000, 3)]
    red = GradientAllReducer(params, bucket_bytes=1024)
    assert len(red.buckets) >= 2
    red.arm()
    loss = sum(((rank + 1) * (i + 1)) * p.sum() for i, p in enumerate(params) if i != 2)   # params[2] unused
    loss.backward()
    red.finish()
    out = [None if p.grad is None else p.grad.clone() for p in params]
    # un-armed backward leaves gradients local
    for p in params:
        p.grad = None
    sum(p.sum() * (rank + 1) for p in params).backward()
    red.finish()
    return {"avg": out, "local": params[0].grad.clone()}


@pytest.mark.parametrize("world", [2, 8])
def test_gradient_allreducer_buckets_and_unused_params(world):
    outs = _run("_reducer", world)
    mean_rank = (world + 1) / 2.0                        # mean over ranks of (rank + 1)
    for r in range(world):
        for i, g in enumerate(outs[r]["avg"]):
            want = 0.0 if i == 2 else mean_rank * (i + 1)
            assert g is not None and torch.allclose(g, torch.full_like(g, want)), (r, i)
        assert torch.allclose(outs[r]["local"], torch.full((5,), float(r + 1)))


# ---------------------------------------------------------------------------------------------------
def _eval_batches():
    g = torch.Generator().manual_seed(5)
    out = []
    for i in range(5):
        image = torch.randn(1, 3, 64, 96, generator=g) * 57
        label = torch.randint(0, 7, (1, 64, 96), generator=g)
        label[0, :3] = 255
        out.append((image, label, torch.tensor([[64 - i, 96 - 2 * i, 3]]), ["v%d" % i]))
    return out


class _TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.c = torch.nn.Conv2d(3, 7, 8, 8)
        with torch.no_grad():
            g = torch.Generator().manual_seed(6)
            self.c.weight.copy_(torch.randn(7, 3, 8, 8, generator=g) * 0.02)
            self.c.bias.copy_(torch.randn(7, generator=g))

    def forward(self, x):
        return [self.c(x)]


def _sharded_eval(rank, world):
    from structure_knowledge_distillation_amd.networks.evaluate import evaluate_main
    m, iu = evaluate_main(_TinyNet(), _eval_batches(), "0", "512,512", 7, whole=True, rank=rank, world=world)
    return {"mean": m, "iu": torch.as_tensor(iu)}


def test_evaluation_is_sharded_over_ranks_and_all_reduced():
    """evalute_model no longer evaluates on rank 0 alone -- batches are dealt round-robin, confusion counts
    all-reduced; every rank gets exactly the single-process result."""
    from oracle import cref
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.networks.evaluate import evaluate_main
    outs = _run("_sharded_eval")
    _lib.install_test_backend(cref.load(_lib.SIGNATURES))
    try:
        want_m, want_iu = evaluate_main(_TinyNet(), _eval_batches(), "0", "512,512", 7, whole=True)
    finally:
        _lib.install_test_backend(None)
    for r in range(2):
        assert outs[r]["mean"] == want_m and torch.equal(outs[r]["iu"], torch.as_tensor(want_iu))


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[3] in miniature: Pi + Pa + Ho on 2 ranks.  The discriminator needs 65 x 65 logits
# (sagan_models.py:131,163), i.e. 512 x 512 images; one image per rank keeps the CPU cost at ~1 minute.
_B, _HW = 1, 512


def _snap(mod):
    return {k: v.detach().clone() for k, v in mod.state_dict().items()}


def _netmodel_step(rank, world):
    from oracle import step_torch as O
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    torch.manual_seed(10 + rank)       # different init per rank: construction must broadcast rank 0's weights
    model = NetModel(default_args(batch_size=_B * world, ho=True, device=torch.device("cpu"), weight_decay=5e-4, lambda_pa=0.5))
    for m in model.student.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    with torch.no_grad():              # attention branches live (gamma is 0 at init)
        model.D_model.attn1.gamma.fill_(0.25)
        model.D_model.attn2.gamma.fill_(-0.5)
    init, teacher, d_init = _snap(model.student), _snap(model.teacher), _snap(model.D_model)
    x, y = O.synthetic_batch(_B * world, _HW, _HW, seed=3)
    alpha = torch.rand(_B * world, 1, 1, 1, generator=torch.Generator().manual_seed(17))
    sl = slice(rank * _B, (rank + 1) * _B)
    model.gp_alpha = alpha[sl]
    model.set_input((x[sl], y[sl], None, None))
    model.optimize_parameters()        # kd_model.py:167-173 incl. the discriminator step
    assert all(p.requires_grad for p in model._d_params)
    grads = {k: p.grad.clone() for k, p in model.student.named_parameters()}            # SGD leaves .grad in place
    d_grads = {k: p.grad.clone() for k, p in model.D_model.named_parameters() if p.grad is not None}
    return {"init": init, "teacher": teacher, "d_init": d_init, "grads": grads, "d_grads": d_grads,
            "losses": {k: getattr(model, k) for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss", "D_loss")},
            "after": _snap(model.student), "d_after": _snap(model.D_model)}


@pytest.mark.timeout(900)
def test_netmodel_step_two_ranks_matches_reference_dp_semantics():
    """Pi + Pa + Ho on two ranks (gloo, C-ABI double) == the reference's DataParallel semantics
    (oracle.step_torch.distillation_step_sharded, fp64): whole-batch student BN statistics, per-shard criteria,
    mean of the shard losses, the discriminator's BatchNorm local to the replica, spectral-norm u / v identical
    everywhere, student AND discriminator gradients averaged by the bucketed all-reduce."""
    from oracle import step_torch as O
    outs = _run("_netmodel_step")
    for name in ("init", "d_init"):
        for k in outs[0][name]:
            assert torch.equal(outs[0][name][k], outs[1][name][k]), "replicas must start identical: %s" % k
    dbl = lambda P: {k: v.double() if v.is_floating_point() else v.clone() for k, v in P.items()}
    PS, PT, PD = dbl(outs[0]["init"]), dbl(outs[0]["teacher"]), dbl(outs[0]["d_init"])
    x, y = O.synthetic_batch(_B * 2, _HW, _HW, seed=3)
    alpha = torch.rand(_B * 2, 1, 1, 1, generator=torch.Generator().manual_seed(17)).double()
    shards = [slice(r * _B, (r + 1) * _B) for r in range(2)]
    cfg = O.StepConfig(weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
    want = O.distillation_step_sharded(PS, PT, PD, x.double(), y, cfg, shards, [alpha[sl] for sl in shards])
    for r in range(2):
        for k, ref in want["shards"][r].items():
            got = outs[r]["losses"][k]
            print("rank %d %-10s product %.8g  sharded oracle %.8g  rel %.2e" % (r, k, got, ref, abs(got - ref) / abs(ref)))
            assert abs(got - ref) <= 1e-4 * abs(ref), (r, k, got, ref)
    assert outs[0]["losses"] != outs[1]["losses"]                 # different shards, different local losses
    worst = {}
    for what, gkey, ref in (("student", "grads", want["grads_S"]), ("D", "d_grads", want["grads_D"])):
        for k, g in ref.items():
            if g is None:
                assert k not in outs[0][gkey] or float(outs[0][gkey][k].abs().max()) == 0.0, k
                continue
            g0, g1 = outs[0][gkey][k], outs[1][gkey][k]
            assert torch.equal(g0, g1), "averaged %s gradients must be identical on every rank: %s" % (what, k)
            err = float((g0.double() - g).norm()) / (float(g.norm()) + 1e-30)
            if float(g.norm()) > 1e-9:                             # (a conv bias in front of a BN has an exactly-zero gradient)
                worst[what] = max(worst.get(what, (0.0, "")), (err, k))
            # fp32 product vs fp64 oracle: backbone gradients are ill-conditioned
            assert float((g0.double() - g).norm()) <= 3e-2 * float(g.norm()) + 1e-6, (what, k, err)
    print("worst gradient error vs the fp64 sharded oracle:", worst)
    for k in outs[0]["after"]:
        assert torch.equal(outs[0]["after"][k], outs[1]["after"][k]), "student replicas diverged: %s" % k
        if "running" in k:
            assert rel(outs[0]["after"][k], PS[k]) < 1e-5, k
    local_bn = 0
    for k in outs[0]["d_after"]:
        a, b = outs[0]["d_after"][k], outs[1]["d_after"][k]
        if k.startswith("preprocess_additional.running"):
            local_bn += int(not torch.equal(a, b))                # sagan_models.py:148: plain BatchNorm2d, NOT synchronised
            for r in range(2):
                assert rel(outs[r]["d_after"][k], want["PD_shards"][r][k]) < 1e-5, (r, k)
            continue
        assert torch.equal(a, b), "discriminator replicas diverged: %s" % k
        if k.endswith(("weight_u", "weight_v")):
            assert rel(a, PD[k]) < 1e-4, k
            assert torch.equal(want["PD_shards"][0][k], want["PD_shards"][1][k]), k
        elif a.is_floating_point() and k in want["grads_D"] and want["grads_D"][k] is not None:
            # parameters after D's SGD step: moved by lr_d x (the averaged gradient, itself within 3e-2 of the oracle's)
            g = want["grads_D"][k]
            assert float((a.double() - PD[k]).norm()) <= 1e-6 * float(PD[k].norm()) + 3e-2 * cfg.lr_d * float(g.norm()) + 1e-12, k
    assert local_bn == 2, "the discriminator's BatchNorm statistics must stay local to the replica"


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[3]'s world size: EIGHT ranks.  Pi + Pa on 8 x 1 images of 128 x 128 (the
# discriminator needs 65 x 65 logits, i.e. 512 x 512 images: the Ho path is covered at world 2 above and on the GPU).
def _netmodel_step_w8(rank, world):
    from oracle import step_torch as O
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    torch.manual_seed(20 + rank)
    model = NetModel(default_args(batch_size=world, ho=False, device=torch.device("cpu"), weight_decay=5e-4, lambda_pa=0.5))
    for m in model.student.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    init, teacher = _snap(model.student), _snap(model.teacher)
    x, y = O.synthetic_batch(world, 128, 128, seed=5)
    model.set_input((x[rank:rank + 1], y[rank:rank + 1], None, None))
    model.optimize_parameters()
    return {"init": init if rank == 0 else None, "teacher": teacher if rank == 0 else None,
            "first": {k: v for k, v in list(init.items())[:3]},
            "grads": {k: p.grad.clone() for k, p in model.student.named_parameters()},
            "losses": {k: getattr(model, k) for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss")},
            "running": {k: v for k, v in _snap(model.student).items() if "running" in k}}


@pytest.mark.timeout(900)
def test_netmodel_step_eight_ranks_matches_reference_dp_semantics():
    """Eight ranks (gloo, C-ABI double): replica broadcast, SyncABN over 8 replicas in every student BN (pooled n = 8 N S),
    bucketed gradient averaging over 8 ranks == the reference's DataParallel semantics (sharded fp64 oracle, 8 shards)."""
    from oracle import step_torch as O
    world = 8
    outs = _run("_netmodel_step_w8", world)
    for r in range(1, world):
        for k, v in outs[0]["first"].items():
            assert torch.equal(v, outs[r]["first"][k]), "replicas must start identical: %s" % k
    dbl = lambda P: {k: v.double() if v.is_floating_point() else v.clone() for k, v in P.items()}
    PS, PT = dbl(outs[0]["init"]), dbl(outs[0]["teacher"])
    x, y = O.synthetic_batch(world, 128, 128, seed=5)
    cfg = O.StepConfig(ho=False, weight_decay=5e-4, lambda_pa=0.5, dropout_p=0.0)
    want = O.distillation_step_sharded(PS, PT, None, x.double(), y, cfg, [slice(r, r + 1) for r in range(world)])
    for r in range(world):
        for k, ref in want["shards"][r].items():
            got = outs[r]["losses"][k]
            assert abs(got - ref) <= 1e-4 * abs(ref) + 1e-12, (r, k, got, ref)
    assert len({round(o["losses"]["G_loss"], 6) for o in outs}) == world         # eight different shards
    worst = (0.0, "")
    for k, g in want["grads_S"].items():
        if g is None:
            continue
        for r in range(1, world):
            assert torch.equal(outs[0]["grads"][k], outs[r]["grads"][k]), "averaged gradients must be identical on every rank: %s" % k
        err = float((outs[0]["grads"][k].double() - g).norm())
        if float(g.norm()) > 1e-9:
            worst = max(worst, (err / float(g.norm()), k))
        assert err <= 3e-2 * float(g.norm()) + 1e-6, (k, err, float(g.norm()))     # fp32 product vs fp64 oracle
    print("world 8: worst gradient error vs the fp64 sharded oracle:", worst)
    for k, v in outs[0]["running"].items():
        for r in range(1, world):
            assert torch.equal(v, outs[r]["running"][k]), k
        assert rel(v, PS[k]) < 1e-5, k


# ---------------------------------------------------------------------------------------------------
# bench.py --gpus N: the warm-up's safety net.  Rehearsed on the CPU double: rank 1 is held back in the first
# warm-up step of the first form for longer than the warm-up's in-kernel wait limit, so rank 0's exchanges time out, raise the
# device status word and poison the step; the ranks must AGREE on that after the step, drop the model and the mailboxes together,
# come back in the next form and produce finite losses there -- and the record that bench.py prints as `comm` must say so.
def _bench_fallback(rank, world):
    import time
    sys.path.insert(0, ROOT)
    import bench
    from oracle import step_torch as O
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    dev = torch.device("cpu")
    args = default_args(batch_size=world, ho=False, device=dev, weight_decay=5e-4, lambda_pa=0.5)
    x, y = O.synthetic_batch(world, 256, 256, seed=5)
    data = (x[rank:rank + 1], y[rank:rank + 1], None, None)
    builds = []

    def build():
        torch.manual_seed(1234)
        model = NetModel(args)
        builds.append(dict(os.environ))
        first_form = len(builds) == 1

        student_forward = model._student_forward

        def late_student_forward():
            time.sleep(4.0)                         # the peer that never arrives in time (limit below: 1.5 s); set_input's all-gather
            return student_forward()                # and the teacher are behind it, the first synchronised layer in front

        def step(i):
            model._student_forward = late_student_forward if (first_form and i == 1 and rank == 1) else student_forward     # (step 0 sets the mailboxes up: collective)
            model.set_input(data)
            model.optimize_parameters()
            return (model.G_loss, model.mc_G_loss, model.pi_G_loss, model.pa_G_loss)
        return model, step

    os.environ.pop("SKD_ABN_SYNC_FUSED", None)
    os.environ.pop("SKD_SYNC_IPC", None)
    P.set_sync_fused(None)
    model, step, info = bench.warm_up_with_fallback(build, 3, world, dev, warm_timeout_s=1.5, run_timeout_s=20.0)
    losses = step(3)                                 # the "timed region": the surviving form works
    return {"info": info, "builds": len(builds), "losses": [float(v) for v in losses], "status": _lib.device_status(),
            "env": {k: os.environ.get(k) for k in ("SKD_ABN_SYNC_FUSED", "SKD_SYNC_IPC", "SKD_SYNC_TIMEOUT_S")},
            "sync_fused": P.sync_fused(),
            "mailbox": P.SyncMailbox.active()}


@pytest.mark.timeout(900)
def test_bench_warm_up_falls_back_to_the_next_exchange_form_on_a_device_status_word():
    outs = _run("_bench_fallback")
    for r, o in enumerate(outs):
        assert o["builds"] == 2 and o["info"]["attempts"] == 2, (r, o["builds"], o["info"])
        assert "three launches per pass" in o["info"]["form"], o["info"]["form"]          # mailboxes kept, no in-kernel exchange
        assert o["info"]["fallback_reason"] and "as configured" in o["info"]["fallback_reason"], o["info"]
        # the fallback changed LIBRARY state (skd_abn_set_sync_fused), not the process environment
        assert o["env"] == {"SKD_ABN_SYNC_FUSED": None, "SKD_SYNC_IPC": None, "SKD_SYNC_TIMEOUT_S": "20.0"}, o["env"]
        assert o["sync_fused"] is False
        assert o["mailbox"] and not any(o["status"])
        assert all(v == v and abs(v) < 1e6 for v in o["losses"]), o["losses"]
    # the reason is a timed-out exchange on at least one rank (the other may only have heard of it through the all-reduce)
    assert any("timed out" in o["info"]["fallback_reason"] or "status words" in o["info"]["fallback_reason"] for o in outs)
    assert outs[0]["losses"][0] != outs[1]["losses"][0]          # different shards


# ---------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` started WITHOUT torchrun: the N = 1 command with the number changed must start its
# own ranks (bench.self_launch: torch.distributed.run on 127.0.0.1, a free port), and rank 0's ONE JSON line must come out of the
# parent's stdout with the parent's exit code 0.  Rehearsed on the CPU double through tests/integration/bench_cpu_double.py (the
# wrapper only installs the double; bench.py refuses --device cpu without one).
def _launch_bench(extra, env_extra=None, entry=None, timeout=800):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    entry = entry or os.path.join(ROOT, "tests", "integration", "bench_cpu_double.py")
    p = subprocess.run([sys.executable, entry] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, [json.loads(l) for l in lines]


@pytest.mark.timeout(900)
def test_bench_gpus_2_without_torchrun_launches_its_own_ranks_and_prints_one_line():
    p, lines = _launch_bench(["--gpus", "2", "--device", "cpu", "--size", "256", "--batch", "1", "--losses", "pi,pa",
                              "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout                                 # rank 0 only
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 2 and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and abs(line["value"] - 2 * 1e3 / line["ms_per_step"]) < 1e-2 * line["value"]      # whole-job images/s
    assert line["comm"]["ranks"] == 2 and line["comm"]["backend"] == "gloo" and line["comm"]["form"]
    assert line["comm"]["fallback_reason"] is None and "rehearsal" in line
    assert all(v == v for v in line["config"]["losses_last_step"].values())
    assert "torch.distributed.run" in p.stderr                       # the parent says what it started


def test_bench_gpus_8_on_a_box_without_gpus_fails_with_a_device_count_message():
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("needs a box with fewer than 8 GPUs")
    p, lines = _launch_bench(["--gpus", "8", "--steps", "20", "--warmup", "5"], entry=os.path.join(ROOT, "bench.py"),
                             env_extra={"SKD_DIST_BACKEND": ""}, timeout=300)
    assert p.returncode == 2
    assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["n_gpus"] == 8
    assert "GPU(s) visible" in lines[0]["error"] and "torchrun" not in lines[0]["error"]


# ---------------------------------------------------------------------------------------------------
# SKD_DIST_SOLO=1 (utils.parallel.solo_rehearsal): a process group of ONE rank takes the N > 1 form of the step -- what a 1-GPU box
# runs over RCCL to show the plumbing (tools/gpu_session.sh solo).  Here: gloo + the C double.
def _solo(rank, world):
    from structure_knowledge_distillation_amd import libs
    from structure_knowledge_distillation_amd.utils import parallel as P
    assert world == 1 and P.world_size() == 1 and P.replicated() and P.comm_form() != "single rank"
    out = _reducer(rank, world)
    params = [torch.nn.Parameter(torch.randn(4))]
    red = P.GradientAllReducer(params)
    assert red.active and len(red._handles) == 1            # the hooks are registered: the bucketed asynchronous form runs
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, 5, 3, generator=g) * 2 + 1
    gz = torch.randn(4, 6, 5, 3, generator=g)
    res = {}
    for name, cls in (("sync", libs.InPlaceABNSync), ("plain", libs.InPlaceABN)):
        mod = cls(6, activation="leaky_relu").train()
        xs = x.clone().requires_grad_(True)
        P.comm_timer.enable()
        z = mod(xs * 1.0)
        (z * gz).sum().backward()
        spans = P.comm_timer.disable()
        res[name] = (z.detach(), xs.grad, mod.weight.grad, mod.bias.grad, mod.running_mean.clone(), mod.running_var.clone(),
                     spans.get("syncabn", (0.0, 0))[1])
    out["abn"] = res
    return out


def test_solo_rehearsal_group_of_one_takes_the_replicated_form(monkeypatch):
    monkeypatch.setenv("SKD_DIST_SOLO", "1")
    out = _run("_solo", world=1)[0]
    for i, g in enumerate(out["avg"]):                      # the one-rank average is the rank's own gradient
        assert torch.allclose(g, torch.full_like(g, 0.0 if i == 2 else float(i + 1)))
    sync, plain = out["abn"]["sync"], out["abn"]["plain"]
    assert sync[6] == 2 and plain[6] == 0                   # the synchronised layer DID exchange (forward + backward), the plain one not
    for a, b in zip(sync[:6], plain[:6]):                   # and a group of one changes nothing
        assert rel(a, b) < 1e-6


def test_without_the_switch_a_group_of_one_is_a_single_rank(monkeypatch):
    monkeypatch.delenv("SKD_DIST_SOLO", raising=False)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    from structure_knowledge_distillation_amd.utils import parallel as P
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert not P.replicated() and P.comm_form() == "single rank"
        assert not P.GradientAllReducer([torch.nn.Parameter(torch.zeros(3))]).active
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_bench_solo_rehearsal_prints_a_line_marked_as_rehearsal():
    p, lines = _launch_bench(["--gpus", "1", "--device", "cpu", "--size", "256", "--batch", "1", "--losses", "pi,pa",
                              "--steps", "2", "--warmup", "1"], env_extra={"SKD_DIST_SOLO": "1", "SKD_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout
    line = lines[0]
    assert line["n_gpus"] == 1 and line["comm"]["ranks"] == 1 and line["comm"]["backend"] == "gloo" and "rehearsal" in line
    assert line["comm"]["form"] != "single rank" and line["comm"]["fallback_reason"] is None


# ---------------------------------------------------------------------------------------------------
# GradientAllReducer, round 6: a bucket segment has its parameter's strides and BECOMES p.grad after finish() (no unpack copy).
def _reducer_views(rank, world):
    from structure_knowledge_distillation_amd.utils.parallel import GradientAllReducer
    torch.manual_seed(3)
    w_cl = torch.nn.Parameter(torch.randn(6, 4, 3, 3).contiguous(memory_format=torch.channels_last))    # a channels-last weight
    w_1x1 = torch.nn.Parameter(torch.randn(5, 4, 1, 1).contiguous(memory_format=torch.channels_last))   # ambiguous strides (size-1 dims)
    b = torch.nn.Parameter(torch.randn(6))
    params = [w_cl, w_1x1, b]
    red = GradientAllReducer(params, bucket_bytes=1 << 20)
    x = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(10 + rank))
    out = {}
    for step, set_to_none in enumerate((True, False)):
        if step == 0:
            for p in params:
                p.grad = None
        else:
            for p in params:
                p.grad.zero_()                                  # zero_grad(set_to_none=False): the gradient already IS its bucket segment
        red.arm()
        y = torch.nn.functional.conv2d(x, w_cl, b, padding=1)
        z = torch.nn.functional.conv2d(x, w_1x1)
        ((rank + 1.0) * (y.square().sum() + z.square().sum())).backward()
        red.finish()
        out[step] = {"grads": [p.grad.clone() for p in params], "strides": [tuple(p.grad.stride()) for p in params],
                     "is_view": [p.grad.data_ptr() == v.data_ptr() for p, v in zip(red.buckets[0].params, red.buckets[0].views)]}
    out["param_strides"] = [tuple(p.stride()) for p in params]
    # the same gradients computed locally, for the expected average
    local = []
    for r in range(world):
        ps = [p.detach().clone().requires_grad_(True) for p in params]
        xr = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(10 + r))
        yr = torch.nn.functional.conv2d(xr, ps[0], ps[2], padding=1)
        zr = torch.nn.functional.conv2d(xr, ps[1])
        ((r + 1.0) * (yr.square().sum() + zr.square().sum())).backward()
        local.append([p.grad for p in ps])
    out["want"] = [sum(g[i] for g in local) / world for i in range(3)]
    return out


def test_gradient_allreducer_segments_have_the_parameters_strides_and_become_the_gradients():
    outs = _run("_reducer_views", 2)
    for o in outs:
        for step in (0, 1):
            assert all(o[step]["is_view"]), "p.grad must BE its bucket segment after finish()"
            for got, want in zip(o[step]["grads"], o["want"]):
                assert rel(got, want) < 1e-5
        assert o[0]["strides"][0] == o["param_strides"][0] and o[0]["strides"][0] != torch.empty(6, 4, 3, 3).stride()   # channels-last kept
        assert o[0]["strides"][2] == (1,)
    for a, b in zip(outs[0][1]["grads"], outs[1][1]["grads"]):
        assert torch.equal(a, b)                                 # replicas hold identical averages
