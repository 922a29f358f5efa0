"""-m gpu: the world_size-2 data-parallel path with the REAL HIP kernels.  Two processes share cuda:0 and
talk through gloo (RCCL refuses two ranks on one device; the driver's 2/4/8-GPU runs use backend "nccl"
through exactly the same code): cross-rank InPlaceABNSync statistics (both the in-place leaky-ReLU form
and the fused BN+ReLU form), bucketed gradient averaging, replica consistency of a NetModel step."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kinks import assert_only_rounding_flips  # noqa: E402

pytestmark = pytest.mark.gpu
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
                      LOCAL_RANK="0", MIOPEN_LOG_LEVEL="3")
    os.environ.setdefault("SKD_SYNC_TIMEOUT_S", "20")      # tests: a protocol mistake must cost seconds of GPU-box time, not 600 s per exchange
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _sync_abn(rank, world):
    from structure_knowledge_distillation_amd import libs
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, 65, 65, generator=g) * 2 + 1
    gz = torch.randn(4, 6, 65, 65, generator=g)
    w, b = torch.randn(6, generator=g), torch.randn(6, generator=g)
    out = {}
    for kind in ("leaky", "relu"):
        mod = libs.InPlaceABNSync(6, activation="leaky_relu" if kind == "leaky" else "none").to(dev).train()
        with torch.no_grad():
            mod.weight.copy_(w); mod.bias.copy_(b)
        xs = x[rank * 2:(rank + 1) * 2].to(dev).requires_grad_(True)
        z = mod(xs * 1.0) if kind == "leaky" else mod.forward_relu(xs * 1.0)
        (z * gz[rank * 2:(rank + 1) * 2].to(dev)).sum().backward()
        out[kind] = {"z": z.detach().cpu(), "dx": xs.grad.cpu(), "dw": mod.weight.grad.cpu(), "db": mod.bias.grad.cpu(),
                     "rm": mod.running_mean.cpu(), "rv": mod.running_var.cpu()}
    return out


def test_sync_abn_two_ranks_hip_kernels():
    from oracle import abn_torch
    outs = _run("_sync_abn")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, 65, 65, generator=g) * 2 + 1
    gz = torch.randn(4, 6, 65, 65, generator=g)
    w, b = torch.randn(6, generator=g), torch.randn(6, generator=g)
    for kind in ("leaky", "relu"):
        xo = x.double().requires_grad_(True)
        wo, bo = w.double().requires_grad_(True), b.double().requires_grad_(True)
        rm, rv = torch.zeros(6, dtype=torch.float64), torch.ones(6, dtype=torch.float64)
        zo = abn_torch.abn_autograd(xo, wo, bo, rm, rv, True, 0.1, 1e-5, "leaky_relu" if kind == "leaky" else "none", 0.01)
        if kind == "relu":
            zo = torch.relu(zo)
        (zo * gz.double()).sum().backward()
        for r in range(2):
            sl = slice(2 * r, 2 * r + 2)
            assert rel(outs[r][kind]["z"], zo[sl]) < 1e-5, kind
            assert rel(outs[r][kind]["dx"], xo.grad[sl]) < 1e-4, kind
            assert rel(outs[r][kind]["rm"], rm) < 1e-5 and rel(outs[r][kind]["rv"], rv) < 1e-5, kind
        assert rel(outs[0][kind]["dw"] + outs[1][kind]["dw"], wo.grad) < 1e-4, kind
        assert rel(outs[0][kind]["db"] + outs[1][kind]["db"], bo.grad) < 1e-4, kind


# ---------------------------------------------------------------------------------------------------
def _sync_abn_nhwc_forms(rank, world):
    """Channels-last InPlaceABNSync through the one-call entries (include/skd.h section 12) in the three ways two ranks can
    take an exchange: both in the ONE-launch form (statistics exchanged by the channel blocks' last arrivers inside the
    register-resident kernel), both in the three-launch form, and MIXED (rank 0 one launch, rank 1 three) -- the protocol
    (one sequence number, per-channel-block flag words) must not care."""
    from structure_knowledge_distillation_amd import libs, _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    dev = torch.device("cuda", 0)
    mb = P.SyncMailbox.get(dist.group.WORLD, dev)
    assert mb is not None and mb.lib.skd_sync_set_timeout(mb.ctx, 8.0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    cap = _lib.get().skd_abn_set_fused_max_workgroups(-1)           # query: what SyncMailbox's device-sharing detection left in place
    out = {"cap": cap, "cus": cus}
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    for C, hw in ((64, 65), (256, 33), (128, 65)):
        g = torch.Generator().manual_seed(C)
        x = torch.randn(4, C, hw, hw, generator=g) * 2 + 1
        r = torch.randn(4, C, hw, hw, generator=g)
        gz = torch.randn(4, C, hw, hw, generator=g)
        w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
        sl = slice(rank * 2, rank * 2 + 2)
        for form in ("fused", "fused_again", "three", "mixed"):
            P.set_sync_fused({"fused": True, "fused_again": True, "three": False, "mixed": rank == 0}[form])     # library state, not the environment
            for kind in ("leaky", "relu", "relu_res"):
                mod = libs.InPlaceABNSync(C, activation="leaky_relu" if kind == "leaky" else "none").to(dev).train()
                with torch.no_grad():
                    mod.weight.copy_(w); mod.bias.copy_(b)
                xs = x[sl].to(dev).requires_grad_(True)
                rs = r[sl].to(dev).requires_grad_(True)
                if kind == "leaky":
                    z = mod(cl(xs * 1.0))
                else:
                    z = mod.forward_relu(cl(xs * 1.0), cl(rs * 1.0) if kind == "relu_res" else None)
                (z * gz[sl].to(dev)).sum().backward()
                torch.cuda.synchronize()
                _lib.raise_on_device_errors()          # a timed-out in-kernel wait: stop here, not 50 exchanges x the time limit later
                out[(C, form, kind)] = {"z": z.detach().contiguous().cpu(), "dx": xs.grad.cpu(), "dr": None if rs.grad is None else rs.grad.cpu(),
                                        "dw": mod.weight.grad.cpu(), "db": mod.bias.grad.cpu(),
                                        "rm": mod.running_mean.cpu(), "rv": mod.running_var.cpu()}
    P.set_sync_fused(None)
    out["status"] = _lib.device_status()
    out["forms"] = _lib.sync_form_counts()
    return out


def test_sync_abn_one_launch_form_two_ranks_and_mixed_forms():
    from oracle import abn_torch
    outs = _run("_sync_abn_nhwc_forms")
    # two ranks on one device: SyncMailbox must have split the compute units between them (grid barrier + in-kernel wait for the peer)
    assert outs[0]["cap"] <= (outs[0]["cus"] - 16) // 2 and outs[0]["cap"] == outs[1]["cap"], (outs[0]["cap"], outs[0]["cus"])
    assert not any(outs[0]["status"]) and not any(outs[1]["status"]), "an in-kernel wait timed out"
    # 3 shapes x 3 kinds x (forward + backward) = 18 synchronised calls per pass: rank 0 ran 54 as one launch (fused twice + mixed)
    # and 18 as three launches, rank 1 36 / 36 -- the forms really differed between the ranks in the mixed pass
    assert outs[0]["forms"] == (54, 18) and outs[1]["forms"] == (36, 36), (outs[0]["forms"], outs[1]["forms"])
    for C, hw in ((64, 65), (256, 33), (128, 65)):
        g = torch.Generator().manual_seed(C)
        x = torch.randn(4, C, hw, hw, generator=g) * 2 + 1
        r = torch.randn(4, C, hw, hw, generator=g)
        gz = torch.randn(4, C, hw, hw, generator=g)
        w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
        for kind in ("leaky", "relu", "relu_res"):
            xo, ro = x.double().requires_grad_(True), r.double().requires_grad_(True)
            wo, bo = w.double().requires_grad_(True), b.double().requires_grad_(True)
            rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
            zo = abn_torch.abn_autograd(xo, wo, bo, rm, rv, True, 0.1, 1e-5, "leaky_relu" if kind == "leaky" else "none", 0.01)
            if kind != "leaky":
                zo = torch.relu(zo + ro if kind == "relu_res" else zo)
            (zo * gz.double()).sum().backward()
            for form in ("fused", "fused_again", "three", "mixed"):
                o = [outs[rk][(C, form, kind)] for rk in range(2)]
                for rk in range(2):
                    sl = slice(2 * rk, 2 * rk + 2)
                    assert rel(o[rk]["z"], zo[sl]) < 1e-5, (C, form, kind)
                    assert rel(o[rk]["dx"], xo.grad[sl]) < 1e-4, (C, form, kind)
                    if kind == "relu_res":
                        assert rel(o[rk]["dr"], ro.grad[sl]) < 1e-5, (C, form, kind)
                    assert rel(o[rk]["rm"], rm) < 1e-5 and rel(o[rk]["rv"], rv) < 1e-5, (C, form, kind)
                # every replica combined the same exchanged numbers in the same order: identical running statistics, bit for bit
                assert torch.equal(o[0]["rm"], o[1]["rm"]) and torch.equal(o[0]["rv"], o[1]["rv"]), (C, form, kind)
                assert rel(o[0]["dw"] + o[1]["dw"], wo.grad) < 1e-4 and rel(o[0]["db"] + o[1]["db"], bo.grad) < 1e-4, (C, form, kind)
            # the one-launch form against the three-launch form: the same numbers up to the rounding of the partial sums; against
            # ITSELF (same inputs again): the same bits -- fixed-order reductions, rank-ordered combine, no atomics
            for rk in range(2):
                a, c, a2 = outs[rk][(C, "fused", kind)], outs[rk][(C, "three", kind)], outs[rk][(C, "fused_again", kind)]
                for k in ("z", "dx", "dw", "db", "rm", "rv"):
                    assert rel(a[k], c[k]) < 5e-6, (C, kind, k)
                    assert torch.equal(a[k], a2[k]), "the one-launch synchronised pass is not reproducible: %s" % ((C, kind, k),)


def _sync_abn_world8(rank, world):
    """Eight ranks -- BASELINE configs[3]'s world size -- on the real kernels: eight processes share the one MI355X (30 workgroups
    of the grid-barrier launches each), the mailboxes hold eight writers' slots, every exchange waits for eight flag words per
    channel block.  Forms: all one-launch, all three-launch, and alternating by rank."""
    from structure_knowledge_distillation_amd import libs, _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    dev = torch.device("cuda", 0)
    mb = P.SyncMailbox.get(dist.group.WORLD, dev)
    assert mb is not None and mb.lib.skd_sync_set_timeout(mb.ctx, 15.0)
    out = {"cap": _lib.get().skd_abn_set_fused_max_workgroups(-1)}
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    for C, hw in ((64, 17), (256, 9)):
        g = torch.Generator().manual_seed(C)
        x = torch.randn(world, C, hw, hw, generator=g) * 2 + 1
        gz = torch.randn(world, C, hw, hw, generator=g)
        for form in ("fused", "three", "mixed"):
            P.set_sync_fused({"fused": True, "three": False, "mixed": rank % 2 == 0}[form])
            mod = libs.InPlaceABNSync(C, activation="leaky_relu").to(dev).train()
            xs = x[rank:rank + 1].to(dev).requires_grad_(True)
            z = mod(cl(xs * 1.0))
            (z * gz[rank:rank + 1].to(dev)).sum().backward()
            torch.cuda.synchronize()
            _lib.raise_on_device_errors()
            out[(C, form)] = {"z": z.detach().contiguous().cpu(), "dx": xs.grad.cpu(), "rm": mod.running_mean.cpu(), "rv": mod.running_var.cpu()}
    P.set_sync_fused(None)
    out["forms"] = _lib.sync_form_counts()
    return out


def test_sync_abn_eight_ranks_on_the_real_kernels():
    from oracle import abn_torch
    world = 8
    outs = _run("_sync_abn_world8", world)
    assert all(o["cap"] == outs[0]["cap"] and 4 <= o["cap"] <= 32 for o in outs), [o["cap"] for o in outs]
    # 2 shapes x (forward + backward) = 4 calls per form: even ranks 8 one-launch + 4 three-launch, odd ranks 4 + 8
    assert [o["forms"] for o in outs] == [(8, 4) if r % 2 == 0 else (4, 8) for r in range(world)]
    for C, hw in ((64, 17), (256, 9)):
        g = torch.Generator().manual_seed(C)
        x = torch.randn(world, C, hw, hw, generator=g) * 2 + 1
        gz = torch.randn(world, C, hw, hw, generator=g)
        xo = x.double().requires_grad_(True)
        rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
        zo = abn_torch.abn_autograd(xo, torch.ones(C, dtype=torch.float64), torch.zeros(C, dtype=torch.float64), rm, rv, True, 0.1, 1e-5,
                                    "leaky_relu", 0.01)
        (zo * gz.double()).sum().backward()
        for form in ("fused", "three", "mixed"):
            for r in range(world):
                o = outs[r][(C, form)]
                assert rel(o["z"], zo[r:r + 1]) < 1e-5 and rel(o["dx"], xo.grad[r:r + 1]) < 1e-4, (C, form, r)
                assert rel(o["rm"], rm) < 1e-5 and rel(o["rv"], rv) < 1e-5, (C, form, r)          # pooled n = 8 N S
                assert torch.equal(o["rm"], outs[0][(C, form)]["rm"]) and torch.equal(o["rv"], outs[0][(C, form)]["rv"]), (C, form, r)


def _sync_timeout(rank, world):
    """A peer that never arrives: the exchange gives up after the context's time limit, the statistics are NaN AND the device
    status word is raised -- _lib.raise_on_device_errors() turns it into an exception (no silent NaN)."""
    import importlib
    import time
    IA = importlib.import_module("structure_knowledge_distillation_amd.libs.inplace_abn")
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    dev = torch.device("cuda", 0)
    mb = P.SyncMailbox.get(dist.group.WORLD, dev)
    assert mb is not None and mb.lib.skd_sync_set_timeout(mb.ctx, 1.0)
    out = {"before": _lib.device_status()}
    if rank == 0:
        stat = torch.ones(2, 8, device=dev)
        t0 = time.perf_counter()
        IA._sync_grad_stats(stat, dist.group.WORLD)          # rank 1 does not call it
        torch.cuda.synchronize()
        out["took"] = time.perf_counter() - t0
        out["nan"] = bool(torch.isnan(stat).all())
        out["after"] = _lib.device_status()
        # a SECOND exchange nobody answers gives up early (the word is still raised) and must NOT overwrite it: the word names the
        # exchange that timed out FIRST (csrc/sync_dev.hpp raise_status_first)
        stat2 = torch.ones(2, 8, device=dev)
        t1 = time.perf_counter()
        IA._sync_grad_stats(stat2, dist.group.WORLD)
        torch.cuda.synchronize()
        out["took2"] = time.perf_counter() - t1
        out["after2"] = _lib.device_status()
        try:
            _lib.raise_on_device_errors()
            out["raised"] = None
        except _lib.SkdDeviceError as e:
            out["raised"] = str(e)
        out["cleared"] = _lib.device_status()
    else:
        time.sleep(3.0)
    dist.barrier()
    P.SyncMailbox.reset()
    return out


def test_exchange_timeout_raises_a_device_status_word_and_an_exception():
    outs = _run("_sync_timeout")
    o = outs[0]
    assert not any(o["before"]) and o["nan"] and 0.9 < o["took"] < 6.0, o          # the 1 s limit, plus launch / sync latency of a busy host
    assert o["after"][0] != 0 and o["raised"] is not None and "timed out waiting for a peer" in o["raised"], o
    assert o["after2"][0] == o["after"][0] and o["took2"] < 0.9, o          # first failure kept; the second wait gave up after ~0.1 s
    assert not any(o["cleared"]) and not any(outs[1]["before"])


# ---------------------------------------------------------------------------------------------------
def _mailbox_vs_collectives(rank, world):
    """csrc/sync.hip with REAL HIP IPC (two processes, one device: each rank's mailbox is the other process's allocation):
    bit-identical to the all_gather + skd_abn_combine_stats / mul_ + all_reduce path it replaces, and timed against it."""
    import importlib
    import time
    IA = importlib.import_module("structure_knowledge_distillation_amd.libs.inplace_abn")
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    dev = torch.device("cuda", 0)
    lib, group = _lib.get(), dist.group.WORLD
    st = torch.cuda.current_stream(dev).cuda_stream
    out = {"rounds": [], "us": {}}
    for rnd in range(5):
        C = (8, 64, 512, 2048, 12)[rnd]
        weighted = rnd % 2 == 1
        if weighted:
            P.set_replica_batch(3 if rank == 0 else 1, dev)
        else:
            P.clear_replica_batch()
        res = {}
        for mode in ("1", "0"):
            os.environ["SKD_SYNC_IPC"] = mode
            P.SyncMailbox.reset()
            mb = P.SyncMailbox.get(group, dev)
            assert (mb is not None) == (mode == "1"), "IPC mailboxes could not be set up between two processes on one device"
            stat = (torch.randn(2, C, generator=torch.Generator().manual_seed(100 * rnd + rank)).abs() + 0.1).to(dev)
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            count = 7 * (3 if rank == 0 else 1) if weighted else 7
            mean, var = IA._sync_stats(stat.clone(), C, count, group, rm, rv, 0.1, lib, st)
            gstat = torch.randn(2, C, generator=torch.Generator().manual_seed(7000 + 100 * rnd + rank)).to(dev)
            for _ in range(3):
                IA._sync_grad_stats(gstat, group)
            torch.cuda.synchronize()
            res[mode] = tuple(t.cpu() for t in (mean, var, rm, rv, gstat))
            if C == 512:                                # latency of one backward exchange at the student's widest layer
                for timed in (False, True):
                    torch.cuda.synchronize(); dist.barrier()
                    t0 = time.perf_counter()
                    for _ in range(200):
                        IA._sync_grad_stats(gstat, group)
                    torch.cuda.synchronize()
                    if timed:
                        out["us"][mode] = (time.perf_counter() - t0) / 200 * 1e6
        for a, b in zip(res["1"], res["0"]):
            assert torch.equal(a, b), "mailbox exchange differs from the collectives (round %d)" % rnd
        out["rounds"].append(res["1"])
    os.environ["SKD_SYNC_IPC"] = "1"
    P.SyncMailbox.reset()
    return out


def test_mailbox_exchange_over_hip_ipc_is_bit_identical_to_the_collectives():
    outs = _run("_mailbox_vs_collectives")
    for a, b in zip(outs[0]["rounds"], outs[1]["rounds"]):
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb), "replicas must hold identical pooled statistics"
    us = outs[0]["us"]
    print("SyncABN exchange of 2 x 512 floats, two ranks on one MI355X, host-to-host per call: mailbox kernel %.1f us, "
          "gloo all_reduce %.1f us" % (us["1"], us["0"]))
    assert us["1"] < us["0"]


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[3] (DP + SyncABN + Ho) on two ranks sharing the MI355X: the whole optimize_parameters()
# with the discriminator step, against the sharded oracle (oracle.step_torch.distillation_step_sharded =
# utils/parallel.py:155 + libs/functions.py:185-209 + sagan_models.py:148 semantics).  The oracle's fp64 / fp32
# results are NOT computed here: they are records in tests/golden/gpu_suite_oracle.pt["sharded2"], written by
# tests/golden/make_golden_gpu_suite.py.  All four variants of the step run in ONE pair of
# worker processes (one import of torch, one HIP context, one warm MIOpen per rank instead of four).
GRAD_BOUND, GRAD_FLOOR = 3.0, 5e-3        # the ONE gradient bound of tests/test_step_gpu.py (reason stated there)
_B = 2
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
# variant -> environment of the step: D stream off / on in the default mode (channels-last student: SyncABN = the one-launch
# form with the exchange inside the kernel), the same with the three-launch form; the deterministic mode (no atomics anywhere;
# PyTorch's im2col convolutions return NCHW tensors, so the student's ABN layers take the NCHW kernels and exchange from Python)
# with the statistics through the IPC mailboxes or through torch.distributed
VARIANTS = {"d0": {"SKD_D_STREAM": "0"}, "d1": {"SKD_D_STREAM": "1"},
            "d1_three": {"SKD_D_STREAM": "1", "SKD_ABN_SYNC_FUSED": "0"},
            "det_ipc1": {"SKD_DETERMINISTIC": "1", "SKD_D_STREAM": "1", "SKD_SYNC_IPC": "1"},
            "det_ipc0": {"SKD_DETERMINISTIC": "1", "SKD_D_STREAM": "1", "SKD_SYNC_IPC": "0"}}


def _generator():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_gpu_suite", os.path.join(GOLDEN_DIR, "make_golden_gpu_suite.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _snap(mod):
    return {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}


def _netmodel_step_once(rank, world, gen):
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    from structure_knowledge_distillation_amd.utils import parallel as P
    dev = torch.device("cuda", 0)
    P.SyncMailbox.reset()                  # SKD_SYNC_IPC is read when the group's mailbox context is built
    from structure_knowledge_distillation_amd import _lib as L
    forms0 = L.sync_form_counts()
    torch.manual_seed(10 + rank)           # different init per rank: construction must broadcast rank 0's weights
    model = NetModel(default_args(batch_size=_B * world, ho=True, device=dev, weight_decay=5e-4, lambda_pa=0.5))
    assert (model._d_stream is not None) == (os.environ.get("SKD_D_STREAM", "1") == "1")
    assert model.deterministic == (os.environ.get("SKD_DETERMINISTIC", "0") == "1")
    for m in model.student.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    built = {k: v for k, v in list(_snap(model.student).items())[:4] + list(_snap(model.D_model).items())[:4]}
    # ... then every rank loads the fixture generator's seeded weights (what the recorded oracle started from)
    PS, PT, PD = gen.init_nets("sharded2")
    model.student.load_state_dict(PS)
    model.teacher.load_state_dict(PT)
    model.D_model.load_state_dict(PD)
    x, y, alpha, shards = gen.sharded2_inputs()
    sl = shards[rank]
    model.gp_alpha = alpha[sl].to(dev)
    model.set_input((x[sl], y[sl], None, None))
    model.optimize_parameters()            # two-stream or serial per SKD_D_STREAM; D's all-reduces start inside its stream
    torch.cuda.synchronize()
    assert all(p.requires_grad for p in model._d_params)
    forms1 = L.sync_form_counts()
    return {"built": built, "ipc": P.SyncMailbox.get(dist.group.WORLD, dev) is not None,     # cached by now: not collective
            "forms": (forms1[0] - forms0[0], forms1[1] - forms0[1]),      # synchronised ABN calls: (one launch, three launches)
            "grads": {k: p.grad.detach().cpu() for k, p in model.student.named_parameters()},
            "d_grads": {k: p.grad.detach().cpu() for k, p in model.D_model.named_parameters() if p.grad is not None},
            "losses": {k: getattr(model, k) for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss", "D_loss")},
            "logits": (model.preds_S[0].detach().cpu(), model.preds_T[0].detach().cpu()),
            "after": _snap(model.student), "d_after": _snap(model.D_model)}


def _netmodel_variants(rank, world):
    import traceback
    torch.set_num_threads(4)               # two ranks x 128 OpenMP threads oversubscribe the GPU box's host
    gen = _generator()
    out = {}
    for name, env in VARIANTS.items():
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        from structure_knowledge_distillation_amd.utils import parallel as _P
        _P.set_sync_fused(None if "SKD_ABN_SYNC_FUSED" not in env else env["SKD_ABN_SYNC_FUSED"] == "1")   # library state: the
        try:                                                                                               # environment is read once
            out[name] = _netmodel_step_once(rank, world, gen)
        except Exception:                  # reported by the test that asks for this variant; the others still run
            out[name] = {"error": traceback.format_exc()}
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            _P.set_sync_fused(None)
            torch.backends.cudnn.enabled = True            # NetModel's deterministic mode is process-wide
            torch.use_deterministic_algorithms(False)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        dist.barrier()
    return out


_SESSION = {}


def _variant(name):
    """Outputs [rank 0, rank 1] of one variant of the two-rank step; all variants are produced by one spawn per session."""
    if "outs" not in _SESSION:
        _SESSION["outs"] = _run("_netmodel_variants")
    outs = [o[name] for o in _SESSION["outs"]]
    for r, o in enumerate(outs):
        assert "error" not in o, "rank %d, variant %s:\n%s" % (r, name, o.get("error"))
    return outs


def _bound_report(what, rows, bound, floor):
    scored = sorted(((err / (bound * base + floor * norm + 1e-7), k, err, base, norm) for k, err, base, norm in rows), reverse=True)
    print("%s: %d tensors, bound err <= %.1f x base + %.0e x |g|; worst three (fraction of the bound, key, err/|g|, base/|g|):"
          % (what, len(scored), bound, floor))
    for frac, k, err, base, norm in scored[:3]:
        print("    %.3f  %-40s %.2e  %.2e" % (frac, k, err / (norm + 1e-30), base / (norm + 1e-30)))
    bad = [(k, frac) for frac, k, _, _, _ in scored if frac > 1.0]
    if bad:                                # localise: every tensor in registration order (a backward bug shows from its layer towards the input)
        for k, err, base, norm in rows:
            print("    %-44s err/|g| %.2e  base/|g| %.2e  |g| %.3e" % (k, err / (norm + 1e-30), base / (norm + 1e-30), norm))
    assert not bad, (what, bad[:10])


def _rec_err(t, rec):
    """(estimated L2 error, norm) against a fixture record {shape, step, sample, norm} -- tests/test_step_gpu.py::_rec_err."""
    import math
    f = t.detach().cpu().double().reshape(-1)
    assert list(t.shape) == rec["shape"], (tuple(t.shape), rec["shape"])
    s = f[::rec["step"]][:rec["sample"].numel()]
    return float((s - rec["sample"]).norm()) * math.sqrt(f.numel() / s.numel()), float(f.norm())


@pytest.mark.parametrize("d_stream", ["0", "1"])
def test_netmodel_ho_step_two_ranks_vs_sharded_oracle(d_stream):
    """Pi + Pa + Ho, two ranks, real HIP kernels: SyncABN in every student BN, the student's AND the discriminator's
    bucketed gradient all-reduce (the latter issued from hooks that fire inside the D stream when SKD_D_STREAM=1),
    D's requires_grad toggle around the G step's critic forward, local D BatchNorm, spectral-norm u / v."""
    from oracle import step_torch as O
    outs = _variant("d" + d_stream)
    gen = _generator()
    fx = torch.load(os.path.join(GOLDEN_DIR, "gpu_suite_oracle.pt"), weights_only=False)["sharded2"]
    PS, PT, PD = gen.init_nets("sharded2")
    for name, P in (("student", PS), ("teacher", PT), ("D", PD)):
        for k, v in gen.checksum(P).items():
            assert abs(v - fx["checksums"][name][k]) <= 1e-9 * max(1.0, abs(v)), ("weight RNG drifted", name, k)
    for k in outs[0]["built"]:
        assert torch.equal(outs[0]["built"][k], outs[1]["built"][k]), "construction must broadcast rank 0's weights: %s" % k
    x, y, alpha, shards = gen.sharded2_inputs()
    cfg = O.StepConfig(weight_decay=fx["cfg"]["weight_decay"], lambda_pa=fx["cfg"]["lambda_pa"], dropout_p=0.0)
    for r in range(2):
        for k, ref in fx["shard_losses"][r].items():
            got = outs[r]["losses"][k]
            print("D_STREAM=%s rank %d %-10s hip %.8g  sharded oracle %.8g  rel %.2e" % (d_stream, r, k, got, ref, abs(got - ref) / abs(ref)))
            assert abs(got - ref) <= 1e-4 * abs(ref), (r, k, got, ref)
    assert outs[0]["losses"] != outs[1]["losses"]
    # averaged gradients: bit-identical on both ranks, student within the ONE bound of the fp64 sharded oracle
    rows = []
    for k, rec in fx["grads_S"].items():
        g0, g1 = outs[0]["grads"][k], outs[1]["grads"][k]
        assert torch.equal(g0, g1), "averaged student gradients must be identical on every rank: %s" % k
        rows.append((k, _rec_err(g0, rec)[0], rec["base"], rec["norm"]))
    _bound_report("D_STREAM=%s student gradients (2 ranks, averaged)" % d_stream, rows, GRAD_BOUND, GRAD_FLOOR)
    rows = []
    for k, rec in fx["grads_D"].items():
        g0, g1 = outs[0]["d_grads"][k], outs[1]["d_grads"][k]
        assert torch.equal(g0, g1), "averaged discriminator gradients must be identical on every rank: %s" % k
        if rec["norm"] > 1e-12:
            rows.append((k, _rec_err(g0, rec)[0], rec["base"], rec["norm"]))
    # end to end the critic amplifies the student's / teacher's logit differences (tests/test_step_gpu.py): informative bound
    _bound_report("D_STREAM=%s discriminator gradients end to end" % d_stream, rows, 10.0, 2e-2)
    # ... and the D step itself on the very logits each rank produced, per shard, averaged: the ONE bound.  (The critic alone
    # is 0.12 GMAC per image: this is the only oracle arithmetic left in the test, about a second.)
    ref = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        acc = {}
        for r, sl in enumerate(shards):
            P = {k: (v.to(dt, copy=True) if v.is_floating_point() else v.clone()) for k, v in PD.items()}
            pS, pT = outs[r]["logits"]
            loss, grads = O.discriminator_step(P, pS.to(dt), pT.to(dt), cfg, alpha[sl].to(dt))
            if name == "f64":
                # (north_star's 1e-4: the critic loss is a cancelling sum on real logits, tests/test_step_gpu.py has the numbers)
                assert abs(outs[r]["losses"]["D_loss"] - loss) <= 1e-4 * abs(loss), (r, outs[r]["losses"]["D_loss"], loss)
            for k, g in grads.items():
                if g is not None:
                    acc[k] = acc.get(k, 0.0) + g / 2
        ref[name] = acc
    _bound_report("D_STREAM=%s discriminator step on the ranks' own logits (averaged over the 2 shards)" % d_stream,
                  [(k, float((outs[0]["d_grads"][k].double() - g).norm()), float((ref["f32"][k].double() - g).norm()), float(g.norm()))
                   for k, g in ref["f64"].items() if float(g.norm()) > 1e-12], GRAD_BOUND, GRAD_FLOOR)
    # replicas after the step
    for k in outs[0]["after"]:
        assert torch.equal(outs[0]["after"][k], outs[1]["after"][k]), "student replicas diverged: %s" % k
    for k, rec in fx["running"].items():
        err, _ = _rec_err(outs[0]["after"][k], rec)
        assert err <= 1e-4 * rec["norm"] + 1e-9, k
    local_bn = 0
    for k in outs[0]["d_after"]:
        a, b = outs[0]["d_after"][k], outs[1]["d_after"][k]
        if k.startswith("preprocess_additional.running"):
            local_bn += int(not torch.equal(a, b))            # sagan_models.py:148: plain BatchNorm2d, not synchronised
            for r in range(2):
                assert rel(outs[r]["d_after"][k], fx["d_bn_running"][r][k]) < 1e-4, (r, k)
            continue
        assert torch.equal(a, b), "discriminator replicas diverged: %s" % k     # incl. weight_u / weight_v, bit for bit
        if k.endswith(("weight_u", "weight_v")):
            assert rel(a, fx["d_uv"][k]) < 1e-4, k
    assert local_bn == 2, "the discriminator's BatchNorm statistics must stay local to the replica"


def test_netmodel_ho_step_two_ranks_mailbox_equals_collectives_bit_for_bit():
    """The whole configs[3] step (Pi + Pa + Ho, SyncABN in every student BN, both gradient all-reduces) on two ranks, once with
    the statistics travelling through the IPC mailboxes and once through torch.distributed, under SKD_DETERMINISTIC=1 (no
    atomics anywhere): every loss, every averaged gradient, every parameter and running statistic after the step must have
    the SAME BITS in both runs, on both ranks."""
    runs = {"1": _variant("det_ipc1"), "0": _variant("det_ipc0")}
    for r in range(2):
        a, b = runs["1"][r], runs["0"][r]
        assert a["ipc"] and not b["ipc"], "the two runs must differ in the transport of the SyncABN statistics"
        assert a["forms"] == (0, 0), "deterministic mode: NCHW activations, the exchange is driven from Python (%s)" % (a["forms"],)
        assert a["losses"] == b["losses"], (r, a["losses"], b["losses"])
        for name in ("grads", "d_grads", "after", "d_after"):
            diff = [k for k in a[name] if not torch.equal(a[name][k], b[name][k])]
            assert not diff, "rank %d: %s differ between the mailbox and the collective exchange: %s" % (r, name, diff[:8])


def test_netmodel_ho_step_two_ranks_one_launch_syncabn_matches_three_launch_form():
    """The default-mode two-rank step with the student's SyncABN layers in the ONE-launch form (exchange inside the
    register-resident kernels: what N > 1 runs by default) against the same step in the three-launch form: the forms really
    differ (library counters), losses agree to 1e-5 (MIOpen's convolutions are not bit-reproducible run to run, so the
    comparison cannot be tighter than two runs of either form), replicas stay identical."""
    one, three = _variant("d1"), _variant("d1_three")
    for r in range(2):
        # 29 training ABN layers, forward + backward; the (B, C, 1, 1) pyramid stage is NCHW-contiguous and exchanges from Python,
        # and with the compute units split between the two ranks the 256 x 256 stem tensors do not fit half a register file: those
        # few calls take the three-launch form beside peers -- the protocol does not care (observed: 50 one-launch, 4 three-launch).
        # Round 6: the stem's bn3 is part of the fused normalise-rectify-pool passes (statistics -> skd_abn_sync_stats -> ...,
        # exchanged by the mailbox kernels directly, not through the one-call *_sync entries these counters count): 28 layers = 56 - 2
        assert one[r]["forms"][0] >= 40 and one[r]["forms"][0] + one[r]["forms"][1] == 54, one[r]["forms"]
        assert three[r]["forms"] == (0, 54), three[r]["forms"]
        for k, v in one[r]["losses"].items():
            assert abs(v - three[r]["losses"][k]) <= 1e-5 * abs(v), (r, k, v, three[r]["losses"][k])
    for k in one[0]["after"]:
        assert torch.equal(one[0]["after"][k], one[1]["after"][k]), "student replicas diverged: %s" % k
        if "running" in k:
            assert rel(one[0]["after"][k], three[0]["after"][k]) < 1e-4, k


def _netmodel_step_world8(rank, world, dev=None):
    """BASELINE configs[3]'s world size -- EIGHT ranks, two images each -- on the real kernels (the eight processes share the one
    MI355X; SyncMailbox caps every rank's grid-barrier launches at its share of the compute units).  Two steps from the same
    weights and inputs: Pi + Pa + Ho (with the gradient that reaches the student's logits recorded by a tensor hook), then
    Pi + Pa on a second NetModel (smooth criteria only: the step whose student gradients a recorded oracle can bound tightly)."""
    import importlib
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    from structure_knowledge_distillation_amd import _lib as L
    from kinks import LeakyRecorder
    PC = importlib.import_module("structure_knowledge_distillation_amd.networks.pspnet_combine")
    torch.set_num_threads(2)
    gen = _generator()
    dev = torch.device("cuda", 0) if dev is None else dev      # (tests/diagnostics/diag_world8_cpu_fixture.py runs this on the CPU double)
    on_gpu = dev.type == "cuda"
    x, y, alpha, shards = gen.sharded8_inputs()
    sl = shards[rank]
    PS, PT, PD = gen.init_nets("sharded8")
    keep = lambda d: {k: v for k, v in d.items()} if rank == 0 else {k: v for k, v in list(d.items())[:6] + list(d.items())[-6:]}
    got = {}
    plain_forward = PC.ResNet.forward

    def recording_forward(self, inp):
        outs = plain_forward(self, inp)
        if self.training and torch.is_grad_enabled():            # the student's training forward: d G_loss / d logits
            outs[0].register_hook(lambda g: got.__setitem__("d_logits", g.detach().clone()))
        return outs

    out = {}
    for name, ho in (("ho", True), ("pa", False)):
        torch.manual_seed(30 + rank)
        model = NetModel(default_args(batch_size=_B * world, ho=ho, device=dev, weight_decay=5e-4, lambda_pa=0.5))
        assert not model._teacher_graph_on          # N > 1: eager teacher
        for m in model.student.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        model.student.load_state_dict(PS); model.teacher.load_state_dict(PT)
        if ho:
            model.D_model.load_state_dict(PD)
            model.gp_alpha = alpha[sl].to(dev)
        forms0 = L.sync_form_counts()
        model.set_input((x[sl], y[sl], None, None))
        PC.ResNet.forward = recording_forward if ho else plain_forward
        try:
            with LeakyRecorder(model.D_model) as rec:      # the critic's own LeakyReLU decisions: G step, D(T), D(S), gradient penalty
                model.optimize_parameters()
        finally:
            PC.ResNet.forward = plain_forward
        if on_gpu:
            torch.cuda.synchronize()
            L.raise_on_device_errors()
        forms1 = L.sync_form_counts()
        o = {"losses": {k: getattr(model, k) for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss") + (("D_loss",) if ho else ())},
             "forms": (forms1[0] - forms0[0], forms1[1] - forms0[1]),
             "grads": keep({k: p.grad.detach().cpu() for k, p in model.student.named_parameters()}),
             "running": keep({k: v.detach().cpu() for k, v in model.student.state_dict().items() if "running" in k})}
        if ho:
            o["logits"] = (model.preds_S[0].detach().cpu(), model.preds_T[0].detach().cpu())
            o["d_logits"] = got["d_logits"].contiguous().cpu()
            o["masks"] = rec.masks
            assert len(o["masks"]) == 16, len(o["masks"])
            o["d_grads"] = {k: p.grad.detach().cpu() for k, p in model.D_model.named_parameters() if p.grad is not None}
            o["d_uv"] = {k: v.detach().cpu() for k, v in model.D_model.state_dict().items() if k.endswith(("weight_u", "weight_v"))}
        out[name] = o
        del model
        if on_gpu:
            torch.cuda.empty_cache()
        dist.barrier()
    return out


def test_netmodel_ho_step_eight_ranks_vs_sharded_oracle():
    """configs[3]'s world size on the real kernels, against the recorded 8-shard fp64 oracle (tests/golden/gpu_suite_oracle.pt
    ["sharded8"]).

    What a recorded oracle can and cannot bound (round 4, profiles/r04q_world8_discontinuities.md; the first version of this test
    sat 10 % from its record with every kernel and every exchange right):
      * the pyramid's 1 x 1 stage normalises B nearly identical pooled vectors (variance ~ eps): ONE of its 16 x 128 units sat at
        y = -8.9e-6 in front of its leaky ReLU with the first batch seed, the GPU's y was +4.5e-6, the slope flipped 0.01 -> 1 and the
        gradient of a whole image channel changed by a factor 100.  The fixture generator now refuses batches that put a pyramid unit
        that close to the kink (make_golden_gpu_suite.pyramid_margins);
      * the adversarial term makes the student's gradients a discontinuous function of its logits in the same way (the critic's
        LeakyReLU slopes): on every rank the product's d G_loss / d logits equals CE + KL part + the fp64 critic's gradient on the
        rank's OWN logits to 2e-5, while the fp64 critic itself moves by 4e-3 .. 1.3e-2 between the product's and the oracle's logits
        (which agree to 1.4e-5) on five of the eight shards.
    So:  (1) losses per shard, replicas, running statistics, u / v against the record;  (2) d G_loss / d logits per rank
    against record (smooth part) + fp64 critic on the rank's own logits ON THE LINEAR PIECE THE RANK TOOK (tests/kinks.py: the
    critic's recorded LeakyReLU decisions replace the oracle's own signs, and only units within rounding of zero may differ) to
    2e-4 -- the tight statement about the Ho gradient (round 4 had widened it to 3e-2 to leave room for flipped units);  (3) the D
    step on the ranks' own logits, same linear piece, under the ONE bound;  (4) every averaged student gradient of the Pi + Pa step
    (smooth criteria) under the ONE bound;  (5) the Ho step's end-to-end gradients to a 10 % bound (a rank missing from the average
    is 12 %).  Size note: BASELINE configs[3] is 8 ranks x 8 images; the eight ranks here share ONE MI355X (and one 45 GB CPU
    fixture), so each holds 2 images (global batch 16) -- the world size, the exchanges and the averaging are configs[3]'s, the
    per-rank batch is not."""
    # In its OWN process group with a hard limit: nine processes (this one's child + eight ranks) that a hang anywhere -- a rank stuck in
    # a collective after a peer died in a way mp.spawn does not see -- must not turn into the suite's time limit.  It takes ~70 s.
    import signal
    import subprocess
    node = os.path.join(ROOT, "tests", "test_distributed_gpu.py") + "::case_netmodel_ho_step_eight_ranks_vs_sharded_oracle"
    cmd = [sys.executable, "-m", "pytest", node, "-q", "-x", "-m", "gpu", "-s", "-p", "no:cacheprovider", "-o", "python_functions=case_*"]
    proc = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=400)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)            # the whole session: pytest child and the eight ranks
        out, _ = proc.communicate()
        pytest.fail("the eight-rank step did not finish within 400 s (process group killed); output tail:\n" + (out or "")[-3000:])
    print("\n".join(l for l in (out or "").splitlines() if "Gloo" not in l and "amdgpu.ids" not in l and "socket.cpp" not in l)[-6000:])
    assert proc.returncode == 0, "see the output above"


def case_netmodel_ho_step_eight_ranks_vs_sharded_oracle():
    """The body of test_netmodel_ho_step_eight_ranks_vs_sharded_oracle (collected only with -o python_functions=case_*)."""
    _check_world8(_run("_netmodel_step_world8", 8), on_gpu=True)


def _check_world8(both, on_gpu=True):
    """The assertions of test_netmodel_ho_step_eight_ranks_vs_sharded_oracle on the eight ranks' outputs (also driven from
    tests/diagnostics/diag_world8_cpu_fixture.py with the C-ABI double on the CPU: same fixture, same checks, no HIP kernels)."""
    world = 8
    outs, outs_pa = [o["ho"] for o in both], [o["pa"] for o in both]
    fx = torch.load(os.path.join(GOLDEN_DIR, "gpu_suite_oracle.pt"), weights_only=False)["sharded8"]
    gen = _generator()
    PS, PT, PD = gen.init_nets("sharded8")
    for name, P in (("student", PS), ("teacher", PT), ("D", PD)):
        for k, v in gen.checksum(P).items():
            assert abs(v - fx["checksums"][name][k]) <= 1e-9 * max(1.0, abs(v)), ("weight RNG drifted", name, k)
    from oracle import step_torch as O
    x, y, alpha, shards = gen.sharded8_inputs()
    cfg = O.StepConfig(weight_decay=fx["cfg"]["weight_decay"], lambda_pa=fx["cfg"]["lambda_pa"], dropout_p=0.0)
    ipc = on_gpu and os.environ.get("SKD_SYNC_IPC", "1") == "1"
    # (1) + (2): per shard
    worst_dl = 0.0
    for r in range(world):
        for k, ref in fx["shard_losses"][r].items():
            if k != "D_loss":              # the critic loss: in (3), where the magnitudes of its cancelling summands are at hand
                assert abs(outs[r]["losses"][k] - ref) <= 1e-4 * abs(ref), (r, k, outs[r]["losses"][k], ref)
        for k, ref in fx["pa"]["shard_losses"][r].items():
            assert abs(outs_pa[r]["losses"][k] - ref) <= 1e-4 * abs(ref) + 1e-12, ("Pi + Pa step", r, k, outs_pa[r]["losses"][k], ref)
        if ipc:
            for o in (outs[r], outs_pa[r]):
                assert o["forms"][0] + o["forms"][1] == 54, o["forms"]      # every channels-last layer through the one-call entries
                                                                             # (28: the stem's bn3 exchanges inside the fused stem passes)
        logits = outs[r]["logits"][0].double().requires_grad_(True)
        P = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in PD.items()}
        lm = O.LeakyMasks(outs[r]["masks"][:4])          # the decisions of the rank's own G-step critic forward
        adv = cfg.lambda_d * O.criterion_adv_for_g(O.discriminator_forward(P, logits, masks=lm), cfg.adv_loss_type)     # kd_model.py:141-142
        want = torch.autograd.grad(adv, logits)[0].reshape(-1)
        assert_only_rounding_flips(lm.done(), "rank %d, critic forward of the G step" % r)
        rec = fx["dlogits_smooth"][r]
        assert list(outs[r]["d_logits"].shape) == rec["shape"]
        idx = torch.arange(rec["sample"].numel()) * rec["step"]
        want_s = want[idx] + rec["sample"]
        got_s = outs[r]["d_logits"].double().reshape(-1)[idx]
        err = float((got_s - want_s).norm() / want_s.norm())
        worst_dl = max(worst_dl, err)
        # On the same linear piece the product's gradient equals the fp64 critic's to ~2e-5 (round 4 measured 2e-5 on the ranks without a
        # flipped unit and 4e-3 = one first-layer unit of 65536 on the one with); a wrong lambda_d, a missing term or a sign is O(1).
        assert err <= 2e-4, "rank %d: d G_loss / d logits differs from CE + KL (record) + critic on its own logits (fp64, same linear piece): %.2e" % (r, err)
    print("world 8: synchronised ABN calls per rank (one launch, three launches):", outs[0]["forms"], "Pi + Pa step:", outs_pa[0]["forms"])
    print("world 8: d G_loss / d logits vs record + fp64 critic on the rank's own logits, worst rank: %.2e" % worst_dl)
    # replicas: identical averaged gradients, running statistics, u / v on all eight ranks -- in both steps
    for r in range(1, world):
        for o0, o in ((outs[0], outs[r]), (outs_pa[0], outs_pa[r])):
            for name in ("grads", "d_grads", "running", "d_uv"):
                for k, v in o.get(name, {}).items():
                    assert torch.equal(v, o0[name][k]), "rank %d differs from rank 0 in %s[%s]" % (r, name, k)
    # (3) the D step on the very logits each rank produced, per shard, averaged over the 8 shards: the ONE bound
    ref, moved = {}, 0.0
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        acc = {}
        for r, sl in enumerate(shards):
            P = {k: (v.to(dt, copy=True) if v.is_floating_point() else v.clone()) for k, v in PD.items()}
            pS, pT = outs[r]["logits"]
            lm, terms = O.LeakyMasks(outs[r]["masks"]), {}
            loss, grads = O.discriminator_step(P, pS.to(dt), pT.to(dt), cfg, alpha[sl].to(dt), masks=lm, terms=terms)
            if name == "f64":
                assert_only_rounding_flips(lm, "rank %d, the four critic forwards of the step" % r)
                got = outs[r]["losses"]["D_loss"]
                # north_star's 1e-4 against the oracle's D step on the rank's OWN logits ...
                assert abs(got - loss) <= 1e-4 * abs(loss), (r, got, loss)
                # ... and end to end against the record (the fp64 oracle on ITS OWN logits, which agree with the rank's to ~1.4e-5).  By the
                # triangle inequality |got - record| <= |got - oracle(own logits)| + |oracle(own logits) - record|: the first term is the
                # statement about the product (1e-4, just asserted), the second is the fp64 critic's own movement between two sets of logits
                # 1.4e-5 apart -- the gradient penalty (|grad D| - 1)^2 makes D_loss about ten times as sensitive to its input as a linear
                # functional (CPU double, product not involved: 1.05e-4 on one shard) -- so 1e-4 cannot hold end to end for ANY correct fp32
                # forward; the end-to-end figure is a sanity bound on that movement (printed), not the parity statement.
                ref_rec = fx["shard_losses"][r]["D_loss"]
                moved = max(moved, abs(loss - ref_rec) / abs(ref_rec))
                assert abs(got - ref_rec) <= 1e-4 * abs(loss) + abs(loss - ref_rec) + 1e-12, (r, got, loss, ref_rec)
                assert abs(got - ref_rec) <= 1e-3 * abs(ref_rec), (r, got, ref_rec, terms)
            for k, g in grads.items():
                if g is not None:
                    acc[k] = acc.get(k, 0.0) + g / world
        ref[name] = acc
    print("world 8: the fp64 critic's D_loss on the ranks' own logits vs on the oracle's logits (the function's movement, product not involved), worst shard: %.2e" % moved)
    _bound_report("world 8 discriminator step on the ranks' own logits (averaged over the 8 shards)",
                  [(k, float((outs[0]["d_grads"][k].double() - g).norm()), float((ref["f32"][k].double() - g).norm()), float(g.norm()))
                   for k, g in ref["f64"].items() if float(g.norm()) > 1e-12], GRAD_BOUND, GRAD_FLOOR)
    # (4) Pi + Pa: every averaged student gradient under the ONE bound, pooled running statistics
    rows = [(k, _rec_err(outs_pa[0]["grads"][k], rec)[0], rec["base"], rec["norm"]) for k, rec in fx["pa"]["grads_S"].items()]
    _bound_report("world 8 student gradients of the Pi + Pa step (averaged over 8 ranks)", rows, GRAD_BOUND, GRAD_FLOOR)
    for k, rec in fx["pa"]["running"].items():
        err, _ = _rec_err(outs_pa[0]["running"][k], rec)
        assert err <= 1e-4 * rec["norm"] + 1e-9, ("Pi + Pa step", k)
    # (5) Ho end to end: sanity only (see the docstring); running statistics and u / v do not depend on the critic's slopes
    for what, key, recs in (("student", "grads", fx["grads_S"]), ("discriminator", "d_grads", fx["grads_D"])):
        rows = sorted(((_rec_err(outs[0][key][k], rec)[0] / rec["norm"], k) for k, rec in recs.items() if rec["norm"] > 1e-12), reverse=True)
        print("world 8 Ho step, %s gradients end to end (informative): worst three err/|g| %s" % (what, [(k, "%.2e" % e) for e, k in rows[:3]]))
        assert rows[0][0] <= 0.10, (what, rows[:5])
    for k, rec in fx["running"].items():
        err, _ = _rec_err(outs[0]["running"][k], rec)
        assert err <= 1e-4 * rec["norm"] + 1e-9, k
    for k, v in fx["d_uv"].items():
        assert rel(outs[0]["d_uv"][k], v) < 1e-4, k


def test_bench_under_torchrun_two_ranks_over_gloo():
    """The exact command line the driver uses for its multi-GPU runs (python -m torch.distributed.run ... bench.py
    --gpus N), with two ranks sharing cuda:0 over gloo (SKD_DIST_BACKEND): rendezvous, replica broadcast, SyncABN
    collectives in every training BN, bucketed gradient all-reduce from the backward hooks, barrier + max-over-ranks
    timing and the single JSON line from rank 0 are all exercised end to end."""
    import json
    import subprocess
    env = dict(os.environ, SKD_DIST_BACKEND="gloo", MIOPEN_LOG_LEVEL="3")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 4 * 1e3 / out["ms_per_step"]) < 1e-2 * out["value"]
    for k, v in out["config"]["losses_last_step"].items():
        assert v == v and abs(v) < 1e6, (k, v)          # finite
    # the N > 1 line explains its own efficiency: roofline stays, comm breakdown added
    assert out["roofline"]["bound"] in ("mfma", "hbm") and out["roofline"]["achieved"] > 0      # mfma: the fused bottleneck-tail GEMM (default)
    comm = out["comm"]
    assert comm["syncabn_collectives"] == 58                                      # 29 training ABN layers, forward + backward
    # channels-last layers exchange inside the ABN calls; the (B, C, 1, 1) pyramid stage is NCHW-contiguous: from Python
    assert comm["syncabn_in_abn_calls"] >= 50 and comm["abn_sync_call_ms"] > 0 and comm["syncabn_one_launch_calls"] >= 40
    assert comm["buckets"] >= 2 and comm["allreduce_wait_ms"] >= 0 and comm["backend"] == "gloo"
    # the warm-up's safety net (bench.warm_up_with_fallback) ran and had nothing to do: first form, no fallback; over gloo on a
    # shared device the configured form is the in-kernel exchange (over RCCL the default is the three-launch form)
    assert comm["forms_tried"] == 1 and comm["fallback_reason"] is None and "exchange inside" in comm["form"], comm


# ---------------------------------------------------------------------------------------------------
# Real multi-GPU: backend "nccl" (= RCCL over xGMI), one rank per device, hipIpcOpenMemHandle ACROSS devices.  Runs only
# where the box has at least two GPUs (the build container and the 1-GPU test boxes skip it); the first multi-GPU box that
# runs `pytest -m gpu` exercises RCCL, the cross-device mailboxes and the in-kernel SyncABN exchange immediately
# (VERDICT r03 item 3b).
def _ngpus():
    try:
        return torch.cuda.device_count()
    except Exception:
        return 0


def _worker_nccl(rank, world, port, fn_name, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world), MIOPEN_LOG_LEVEL="3")
    os.environ.setdefault("SKD_SYNC_TIMEOUT_S", "60")
    os.environ.pop("SKD_DIST_BACKEND", None)
    if fn_name == "_multi_gpu_plumbing":
        # the plumbing test is the one that opts into the in-kernel exchange over RCCL (default over RCCL since round 5: the
        # three-launch form, utils/parallel.init_distributed) -- with the compute-unit reserve that opting in applies
        os.environ["SKD_ABN_SYNC_FUSED"] = "1"
    torch.set_num_threads(4)
    from structure_knowledge_distillation_amd.utils import parallel as P
    P.init_distributed()                    # picks "nccl", one device per rank, exactly like bench.py under torchrun
    assert dist.get_backend() == "nccl" and torch.cuda.current_device() == rank
    try:
        out = globals()[fn_name](rank, world)
        torch.save(out, os.path.join(outdir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _run_nccl(fn_name, world):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_nccl, args=(world, _free_port(), fn_name, d), nprocs=world, join=True)
        return [torch.load(os.path.join(d, "r%d.pt" % r)) for r in range(world)]


def _multi_gpu_plumbing(rank, world, dev=None):
    import importlib
    IA = importlib.import_module("structure_knowledge_distillation_amd.libs.inplace_abn")
    from structure_knowledge_distillation_amd import libs, _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    dev = torch.device("cuda", rank) if dev is None else dev      # (tests/diagnostics/diag_multi_gpu_cpu_rehearsal.py: the CPU double)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    lib, group = _lib.get(), dist.group.WORLD
    st = _lib.stream_of(torch.empty(1, device=dev))
    out = {}
    # (1) the mailboxes across devices, against RCCL's own collectives on the same data
    res = {}
    for mode in ("1", "0"):
        os.environ["SKD_SYNC_IPC"] = mode
        P.SyncMailbox.reset()
        mb = P.SyncMailbox.get(group, dev)
        out["mailbox_" + mode] = mb is not None
        C = 512
        stat = (torch.randn(2, C, generator=torch.Generator().manual_seed(100 + rank)).abs() + 0.1).to(dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        mean, var = IA._sync_stats(stat.clone(), C, 7, group, rm, rv, 0.1, lib, st)
        gstat = torch.randn(2, C, generator=torch.Generator().manual_seed(7000 + rank)).to(dev)
        IA._sync_grad_stats(gstat, group)
        sync()
        res[mode] = [t.cpu() for t in (mean, var, rm, rv, gstat)]
    out["stats"] = res
    os.environ["SKD_SYNC_IPC"] = "1"
    P.SyncMailbox.reset()
    # (2) channels-last InPlaceABNSync through the one-call entries: the exchange inside the register-resident kernels over xGMI
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    g = torch.Generator().manual_seed(1)
    n = 2 * world
    x = torch.randn(n, 64, 33, 33, generator=g) * 2 + 1
    gz = torch.randn(n, 64, 33, 33, generator=g)
    sl = slice(2 * rank, 2 * rank + 2)
    mod = libs.InPlaceABNSync(64, activation="leaky_relu").to(dev).train()
    xs = x[sl].to(dev).requires_grad_(True)
    z = mod(cl(xs * 1.0))
    (z * gz[sl].to(dev)).sum().backward()
    sync()
    out["abn"] = {"z": z.detach().contiguous().cpu(), "dx": xs.grad.cpu(), "rm": mod.running_mean.cpu(), "rv": mod.running_var.cpu()}
    out["cap"] = lib.skd_abn_set_fused_max_workgroups(-1)
    # (3) bucketed gradient averaging over RCCL
    params = [torch.nn.Parameter(torch.full((n_,), float(rank + 1), device=dev)) for n_ in (5, 3000000, 7, 1000)]
    red = P.GradientAllReducer(params, bucket_bytes=1 << 20)
    red.arm()
    sum((i + 1) * (rank + 1) * p.sum() for i, p in enumerate(params)).backward()
    red.finish()
    sync()
    out["avg"] = [float(p.grad[0]) for p in params]
    out["status"] = _lib.device_status()
    return out


@pytest.mark.skipif(_ngpus() < 2, reason="needs at least two GPUs (RCCL + cross-device HIP IPC)")
def test_multi_gpu_rccl_and_cross_device_mailboxes():
    world = min(_ngpus(), 8)
    from structure_knowledge_distillation_amd.utils.parallel import reserved_fused_cap
    # one rank per device over RCCL: no sharing, but a reserve of compute units for the collectives' kernels (utils/parallel.py)
    _check_multi_gpu_plumbing(_run_nccl("_multi_gpu_plumbing", world), world, reserved_fused_cap(torch.cuda.get_device_properties(0).multi_processor_count))


def _check_multi_gpu_plumbing(outs, world, cap):
    from oracle import abn_torch
    for o in outs:
        assert o["mailbox_1"] and not o["mailbox_0"], "cross-device IPC mailboxes could not be set up"
        assert not any(o["status"]) and o["cap"] == cap, (o["status"], o["cap"], cap)
        for a, b in zip(o["stats"]["1"], o["stats"]["0"]):
            assert rel(a, b) < 1e-6                      # RCCL's ring order vs the mailboxes' rank order: rounding only
        for a, b in zip(o["stats"]["1"], outs[0]["stats"]["1"]):
            assert torch.equal(a, b), "replicas must hold identical pooled statistics"
        mean_rank = (world + 1) / 2.0
        assert all(abs(v - (i + 1) * mean_rank) < 1e-5 * mean_rank * (i + 1) for i, v in enumerate(o["avg"])), o["avg"]
    g = torch.Generator().manual_seed(1)
    n = 2 * world
    x = torch.randn(n, 64, 33, 33, generator=g) * 2 + 1
    gz = torch.randn(n, 64, 33, 33, generator=g)
    xo = x.double().requires_grad_(True)
    rm, rv = torch.zeros(64, dtype=torch.float64), torch.ones(64, dtype=torch.float64)
    zo = abn_torch.abn_autograd(xo, torch.ones(64, dtype=torch.float64), torch.zeros(64, dtype=torch.float64), rm, rv, True, 0.1, 1e-5,
                                "leaky_relu", 0.01)
    (zo * gz.double()).sum().backward()
    for r, o in enumerate(outs):
        sl = slice(2 * r, 2 * r + 2)
        assert rel(o["abn"]["z"], zo[sl]) < 1e-5 and rel(o["abn"]["dx"], xo.grad[sl]) < 1e-4
        assert rel(o["abn"]["rm"], rm) < 1e-5 and rel(o["abn"]["rv"], rv) < 1e-5
        assert torch.equal(o["abn"]["rm"], outs[0]["abn"]["rm"])


def _multi_gpu_step(rank, world, dev=None):
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    from oracle import step_torch as O
    dev = torch.device("cuda", rank) if dev is None else dev
    torch.manual_seed(10 + rank)
    model = NetModel(default_args(batch_size=_B * world, ho=True, device=dev, weight_decay=5e-4, lambda_pa=0.5))
    for m in model.student.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    if world == 2:                          # the recorded sharded oracle (tests/golden/gpu_suite_oracle.pt["sharded2"]) applies
        gen = _generator()
        PS, PT, PD = gen.init_nets("sharded2")
        model.student.load_state_dict(PS); model.teacher.load_state_dict(PT); model.D_model.load_state_dict(PD)
        x, y, alpha, shards = gen.sharded2_inputs()
        sl = shards[rank]
    else:
        x, y = O.synthetic_batch(_B * world, 512, 512, seed=3)
        alpha = torch.rand(_B * world, 1, 1, 1, generator=torch.Generator().manual_seed(17))
        sl = slice(rank * _B, (rank + 1) * _B)
    losses = []
    for step in range(2):
        model.gp_alpha = alpha[sl].to(dev)
        model.set_input((x[sl], y[sl], None, None))
        model.optimize_parameters()
        losses.append({k: getattr(model, k) for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss", "D_loss")})
    if dev.type == "cuda":
        torch.cuda.synchronize()
    from structure_knowledge_distillation_amd import _lib
    return {"losses": losses, "after": _snap(model.student), "d_after": _snap(model.D_model), "status": _lib.device_status(),
            "grads": {k: p.grad.detach().cpu() for k, p in list(model.student.named_parameters())[:8]}}


@pytest.mark.skipif(_ngpus() < 2, reason="needs at least two GPUs (RCCL + cross-device HIP IPC)")
def test_multi_gpu_netmodel_steps_over_rccl():
    """Two full Pi + Pa + Ho steps, one rank per GPU over RCCL: replicas stay bit-identical (student, discriminator incl. u / v),
    no in-kernel wait times out; with exactly two GPUs the first step's per-shard losses meet the recorded sharded oracle."""
    world = min(_ngpus(), 8)
    _check_multi_gpu_step(_run_nccl("_multi_gpu_step", world), world)


def _check_multi_gpu_step(outs, world):
    for o in outs:
        assert not any(o["status"])
        assert all(v == v and abs(v) < 1e6 for step in o["losses"] for v in step.values())
    for k in outs[0]["after"]:
        for o in outs[1:]:
            assert torch.equal(outs[0]["after"][k], o["after"][k]), "student replicas diverged over RCCL: %s" % k
    for k in outs[0]["d_after"]:
        if not k.startswith("preprocess_additional.running"):
            for o in outs[1:]:
                assert torch.equal(outs[0]["d_after"][k], o["d_after"][k]), "discriminator replicas diverged over RCCL: %s" % k
    if world == 2:
        fx = torch.load(os.path.join(GOLDEN_DIR, "gpu_suite_oracle.pt"), weights_only=False)["sharded2"]
        for r in range(2):
            for k, ref in fx["shard_losses"][r].items():
                got = outs[r]["losses"][0][k]
                assert abs(got - ref) <= 1e-4 * abs(ref), (r, k, got, ref)


# ---------------------------------------------------------------------------------------------------
# RCCL on ONE GPU: a communicator of one rank under SKD_DIST_SOLO=1 (utils.parallel.solo_rehearsal) runs the N > 1 FORM of the step
# -- replica broadcast, gradient hooks + bucketed asynchronous all-reduce, synchronised InPlace-ABN over torch.distributed
# collectives, eager teacher -- on backend "nccl".  No data crosses a link, but ncclCommInitRank, ProcessGroupNCCL's stream / event
# plumbing and every device-side verdict tensor of that control flow do run on hardware; the result must be the single-rank step's.
def _worker_solo(rank, world, port, outdir, solo):
    sys.path.insert(0, ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "SKD_DIST_BACKEND"):
        os.environ.pop(k, None)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MIOPEN_LOG_LEVEL="3", SKD_DIST_SOLO="1" if solo else "0")
    torch.set_num_threads(4)
    from structure_knowledge_distillation_amd.utils import parallel as P
    P.init_distributed()
    if solo:
        assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1 and P.replicated()
    else:
        assert not dist.is_initialized() and not P.replicated()
    try:
        out = _multi_gpu_step(0, 1)
        out["form"] = P.comm_form()
        torch.save(out, os.path.join(outdir, "solo%d.pt" % int(solo)))
    finally:
        if solo:
            dist.destroy_process_group()


def test_solo_rccl_group_of_one_netmodel_steps_equal_the_single_rank_steps():
    with tempfile.TemporaryDirectory() as d:
        for solo in (True, False):
            mp.spawn(_worker_solo, args=(1, _free_port(), d, solo), nprocs=1, join=True)
        a = torch.load(os.path.join(d, "solo1.pt"))
        b = torch.load(os.path.join(d, "solo0.pt"))
    assert a["form"].startswith("ipc mailboxes, three launches") and b["form"] == "single rank"     # the default form over RCCL
    assert not any(a["status"]) and not any(b["status"])
    for step in range(2):
        # (the second step runs on weights updated from default-mode gradients -- MIOpen's atomic split-K sums -- which two runs of
        # the SAME form do not reproduce bit for bit either: the tiny pair-wise term moves by ~1e-4 of itself there)
        tol = 1e-4 if step == 0 else 2e-3
        for k in ("mc_G_loss", "pi_G_loss", "pa_G_loss", "G_loss"):
            assert abs(a["losses"][step][k] - b["losses"][step][k]) <= tol * abs(b["losses"][step][k]), (step, k, a["losses"], b["losses"])
        assert abs(a["losses"][step]["D_loss"] - b["losses"][step]["D_loss"]) <= 1e-3 * max(1.0, abs(b["losses"][step]["D_loss"]))
    diffs = sorted(((rel(a["after"][k], b["after"][k]), k) for k in a["after"]
                    if a["after"][k].dtype.is_floating_point and a["after"][k].numel() > 1), reverse=True)
    print("solo RCCL vs single rank, student state after two steps: worst relative differences %s"
          % [("%.2e" % d, k) for d, k in diffs[:4]])
    # NOT a rounding-level comparison, and it cannot be one: the step's gradients are conditioned at the 1e-3 level in fp32 (the
    # gradient tests' `base`: the fp32 CPU oracle itself sits 3.6e-3 from the fp64 one on the stem's tensors), and the synchronised
    # ABN passes sum in another order than the single-rank ones -- tests/diagnostics/diag_solo_vs_plain.py: every form of the
    # exchange lands 3.6e-3 ... 5.3e-3 from the single-rank gradients in step 0 while two single-rank runs agree to 3e-6.  The
    # accuracy of the N > 1 form is established against the fp64 oracle by the two- and eight-rank tests above; this test is about
    # the RCCL plumbing producing THE SAME STEP.  The BN shifts start at zero in front of another training-mode BN (exact gradient
    # ~ 0): what they accumulate is that noise itself.
    for d, k in diffs:
        noise = a["after"][k].dim() == 1 and not k.split(".")[-1].startswith("running")
        assert d < (0.25 if noise else 1e-2), (k, d)


def _worker_solo_d_graph(rank, world, port, outdir, flag):
    sys.path.insert(0, ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "SKD_DIST_BACKEND"):
        os.environ.pop(k, None)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MIOPEN_LOG_LEVEL="3", SKD_DIST_SOLO="1", SKD_DETERMINISTIC="1",
                      SKD_TEACHER_GRAPH="0", SKD_D_GRAPH=flag)
    torch.set_num_threads(4)
    from structure_knowledge_distillation_amd.utils import parallel as P
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    from oracle import step_torch as O
    P.init_distributed()
    assert dist.get_backend() == "nccl" and P.replicated()
    try:
        dev = torch.device("cuda", 0)
        torch.manual_seed(99)
        args = default_args(batch_size=2, device=dev, ho=True, weight_decay=5e-4, lambda_pa=0.5)
        model = NetModel(args)
        assert model.deterministic and model._d_graph_on == (flag == "1") and model._d_reducer.active
        losses = []
        for step in range(5):
            images, labels = O.synthetic_batch(2, 512, 512, seed=step)
            model.gp_alpha = torch.rand(2, 1, 1, 1, generator=torch.Generator().manual_seed(70 + step)).to(dev)
            torch.manual_seed(500 + step)
            model.adjust_learning_rate(args.lr_d, model.D_solver, step * 1000)
            model.set_input((images, labels, None, None))
            model.optimize_parameters()
            losses.append([model.G_loss, model.mc_G_loss, model.pi_G_loss, model.pa_G_loss, model.D_loss])
        torch.cuda.synchronize()
        assert len(model._d_graphs) == (1 if flag == "1" else 0)
        torch.save({"losses": losses, "student": _snap(model.student), "D": _snap(model.D_model)}, os.path.join(outdir, "dg%s.pt" % flag))
    finally:
        dist.destroy_process_group()
        torch.backends.cudnn.enabled = True
        torch.use_deterministic_algorithms(False)


def test_solo_rccl_d_step_hipgraph_with_the_all_reduce_behind_it_equals_eager_bit_for_bit():
    """SKD_D_GRAPH=1 in the N > 1 form (one-rank RCCL communicator): the capture holds no collective, the critic's gradients are
    averaged after the replay -- deterministic mode, two eager + capture + two replay steps: the same bits as the eager D step with its
    hook-driven all-reduce, in every loss and every student / discriminator tensor."""
    with tempfile.TemporaryDirectory() as d:
        for flag in ("0", "1"):
            mp.spawn(_worker_solo_d_graph, args=(1, _free_port(), d, flag), nprocs=1, join=True)
        eager = torch.load(os.path.join(d, "dg0.pt"))
        graph = torch.load(os.path.join(d, "dg1.pt"))
    assert eager["losses"] == graph["losses"], (eager["losses"], graph["losses"])
    for what in ("student", "D"):
        diff = [k for k, v in eager[what].items() if not torch.equal(v, graph[what][k])]
        assert not diff, "%s state differs with the D step replayed from a hipGraph in the N > 1 form: %s" % (what, diff[:8])
