#!/usr/bin/env python
"""bench.py -- distillation-step images/sec @512x512 (Pi+Pa+Ho) on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

``--gpus N`` (N > 1) WITHOUT torchrun starts its own ranks (self_launch: the same command line re-run under torch.distributed.run
on 127.0.0.1 and a free port; fewer than N devices -> one JSON line with "error", exit code 2); under torchrun it joins the
launcher's group.  Either way rank 0 prints the ONE line.

A step = one NetModel.optimize_parameters() (kd_model.py:167-173): frozen ResNet101-PSPNet teacher
forward + ResNet18-PSPNet student forward/backward (InPlace-ABN at every BN) + CE/Pi/Pa/Ho losses +
SGD + discriminator step (adv + WGAN-GP) + SGD, fp32, batch 8 per GPU, synthetic 512x512 19-class
tensors already resident in HBM, Dropout on, logged scalars read back every step like
train_and_eval.py:26.  N > 1: one process per GPU, weak scaling (8 images per rank), RCCL gradient
all-reduce + cross-GPU InPlaceABNSync.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant hand-written kernel of the step, timed live with HIP events on the launch stream over the
                timed steps: since round 3 the fused bottleneck-tail GEMM of the frozen teacher (conv1x1_abn_kernel:
                bn2 + ReLU prologue, 1x1 convolution on fp32 MFMA, bn3 + residual + ReLU epilogue; 2*M*K*N algorithmic
                flops per launch, bound "mfma")
  kernels       the same measurement for the other hand-written kernels / kernel chains
  cpu_baseline  the CPU oracle (oracle/step_torch.py, a port of the reference step pinned to the
                reference's own Python) timed on this host's cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, exact fp32
STEP_TFLOP_PER_IMAGE = 0.959   # teacher fwd 579.3 GF + student fwd 126.7 + bwd 253.3
# the PSP bottleneck fold (csrc/ppm.hip, networks/pspnet_combine.PSP_FOLD) evaluates the priors' half of both bottleneck convolutions by
# a small GEMM + fold kernel: 2*65*65*9*(512*2048 [teacher fwd] + 3 * 128*512 [student fwd + 2 bwd]) flop per image less
FOLD_TFLOP_PER_IMAGE = 2 * 65 * 65 * 9 * (512 * 2048 + 3 * 128 * 512) / 1e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE: 8)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-pairwise-sweep", action="store_true")
    ap.add_argument("--losses", default="pi,pa,ho", help="distillation terms of the step (BASELINE configs[2]: all three); the CPU "
                                                       "rehearsal of the launcher runs 'pi,pa' at 256x256 (no 65x65 logits for D there)")
    ap.add_argument("--device", default="cuda", choices=("cuda", "cpu"),
                    help="'cpu' is the launcher REHEARSAL only (tests/test_distributed_cpu.py): it needs a C-ABI test double installed "
                         "by the test's sitecustomize and refuses to run otherwise; nothing is measured in that mode")
    ap.add_argument("--dsn-ab", action="store_true", help="also re-time the step without the teacher's dead DSN head (informative; opt-in: "
                                                         "the default run stays exactly the measured configuration and nothing else)")
    return ap.parse_args()


def cpu_baseline(seconds, size):
    """Reference step on the host cores (the oracle port; /root/reference does not exist on the GPU box).  Two bounded
    samples: BASELINE configs[0] -- the reference's own CPU-runnable case, Pi only, batch 2, 256x256 -- as the median
    of three steps after a warm-up, and the benchmarked Pi+Pa+Ho step at batch 2, 512x512
    (the 'value': same metric as the GPU line)."""
    import statistics
    import torch
    from oracle import step_torch as O
    B = 2
    cores = torch.get_num_threads()
    PS, PT, PD = O.pspnet_init(O.STUDENT, 19, 1), O.pspnet_init(O.TEACHER, 19, 2), O.discriminator_init(seed=3)
    c1 = O.StepConfig(pi=True, pa=False, ho=False, weight_decay=5e-4)
    x1, y1 = O.synthetic_batch(B, 256, 256)
    st1 = {"G": {}, "D": {}}
    O.distillation_step(PS, PT, None, x1, y1, c1, st1)            # warm-up
    t1 = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.distillation_step(PS, PT, None, x1, y1, c1, st1)
        t1.append(time.perf_counter() - t0)
    cfg = O.StepConfig(weight_decay=5e-4, lambda_pa=0.5)
    state = {"G": {}, "D": {}}
    x, y = O.synthetic_batch(B, size, size)
    O.distillation_step(PS, PT, PD, x, y, cfg, state)            # warm-up
    times, t_all = [], time.perf_counter()
    while True:                                                    # >= 3 steps, more while the budget lasts (<= 5): MEDIAN step
        t0 = time.perf_counter()
        O.distillation_step(PS, PT, PD, x, y, cfg, state)
        times.append(time.perf_counter() - t0)
        if len(times) >= 5 or (len(times) >= 3 and time.perf_counter() - t_all >= seconds):
            break
    n, el = len(times), statistics.median(times) * len(times)    # value = B / median step time
    try:
        with open("/proc/cpuinfo") as fh:
            model = [l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": round(B * n / el, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "median of %d timed steps (+1 warm-up; step times %s s) of the full Pi+Pa+Ho step at batch %d, %dx%d, fp32, "
                      "torch CPU (oracle/step_torch.py); host CPU: %s" % (n, "/".join("%.1f" % t for t in times), B, size, size, model),
            "config1": {"value": round(B / statistics.median(t1), 4), "unit": "images/sec",
                        "sample": "median of 3 steps (+1 warm-up), BASELINE configs[0]: Pi only, batch 2, 256x256"}}


def pairwise_sweep(dev):
    """Pa Gram kernel (gram_loss_kernel, fp32 MFMA) at B=8, C_S=128, C_T=512 over the pool scales of
    SURVEY.md 8d: M = 9 (reference default) ... 4225 (pool-scale -> 1/65), timed with HIP events around back-to-back
    launches on the launch stream.  Two conventions side by side: TFLOPs = 2*B*M^2*(Cs+Ct) (the full M x M matrix the
    reference computes) and executed_TFLOPs = what the matrix pipe actually ran (upper-triangle tiles of
    the zero-padded problem); the executed figure is the one the SQ_VALU_MFMA_BUSY_CYCLES counter confirms (profiles/)."""
    import torch
    from structure_knowledge_distillation_amd import _lib
    lib = _lib.load()
    B, Cs, Ct = 8, 128, 512
    out = {}
    for M in (9, 81, 289, 1089, 4225):
        ldm = lib.skd_pairwise_ldm(M)
        ps, pt = torch.randn(B, Cs, M, device=dev), torch.randn(B, Ct, M, device=dev)
        fs, ft = torch.empty(B, Cs, ldm, device=dev), torch.empty(B, Ct, ldm, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        lib.skd_channel_l2_normalise(B, Cs, M, ps.data_ptr(), fs.data_ptr(), ldm, None, 0, None, st)
        lib.skd_channel_l2_normalise(B, Ct, M, pt.data_ptr(), ft.data_ptr(), ldm, None, 0, None, st)
        G = torch.empty(B, ldm, ldm, device=dev)
        loss = torch.empty(1, device=dev)
        ws = torch.empty(max(1, lib.skd_pairwise_workspace_floats(B, M)), device=dev)
        call = lambda: lib.skd_pairwise_gram_loss(B, Cs, Ct, M, ldm, fs.data_ptr(), ft.data_ptr(), G.data_ptr(),
                                                  loss.data_ptr(), ws.data_ptr(), st)
        nt = ldm // 128                       # G is symmetric: only the nt (nt + 1) / 2 upper-triangle 128 x 128 tiles run
        flop_full = 2.0 * B * M * M * (Cs + Ct)
        flop_exec = 2.0 * B * (nt * (nt + 1) // 2) * 128 * 128 * (Cs + Ct)
        flop_useful = 2.0 * B * (M * (M + 1) // 2) * (Cs + Ct)       # unique entries of the symmetric matrices

        def warm_up(fn, ms=30.0):
            """>= 30 ms of back-to-back launches before anything is timed (round 6): the sweep follows host-side allocation and
            random-number generation, i.e. a GPU that has dropped its clocks, and three warm-up launches read 5-10 % slow
            (profiles/r06_gemm_lab_notes.md)."""
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            spent = 0.0
            for _ in range(400):
                w0.record()
                for _ in range(4):
                    fn()
                w1.record()
                w1.synchronize()
                spent += w0.elapsed_time(w1)
                if spent >= ms:
                    break

        def timed(fn, reps=20):
            """every launch bracketed on its own: median AND minimum of `reps` (five back-to-back launches
            hid a 13 % spread between sessions); launches below ~60 us are timed back to back (event overhead)."""
            warm_up(fn)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for e0, e1 in ev:
                e0.record()
                fn()
                e1.record()
            torch.cuda.synchronize()
            t = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
            return t[len(t) // 2], t[0], t[-1]

        if M <= 289:
            warm_up(call, 10.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = mn = mx = e0.elapsed_time(e1) / 20
        else:
            ms, mn, mx = timed(call)
        # the only FRACTION printed is of executed matrix work (incl. the zero padding to ldm); the full-matrix figure is the
        # reference-algorithm rate (what einsum would have had to sustain), a convention, not a utilisation: it may exceed the peak
        rate = lambda f, t: f / (t * 1e-3) / 1e12
        out["M=%d" % M] = {"us": round(ms * 1e3, 1), "us_min": round(mn * 1e3, 1), "us_max": round(mx * 1e3, 1), "reps": 20,
                           "TFLOPs_full_matrix_convention": round(rate(flop_full, ms), 2),
                           "executed_TFLOPs": round(rate(flop_exec, ms), 2),
                           "frac_fp32_mfma_executed": round(rate(flop_exec, ms) / MFMA_F32_PEAK_TFLOPS, 4),
                           "frac_fp32_mfma_executed_best_launch": round(rate(flop_exec, mn) / MFMA_F32_PEAK_TFLOPS, 4),
                           "unique_useful_TFLOPs": round(rate(flop_useful, ms), 2),
                           "frac_fp32_mfma_unique_useful": round(rate(flop_useful, ms) / MFMA_F32_PEAK_TFLOPS, 4)}
        if M >= 1089:
            # the backward GEMM dP = Fhat_S G (2 B ldm^2 Cs executed flop on the zero-padded problem; round 5: both operands k-major;
            # round 6: stream-K over (output tile, K-tile) units, one round of the chip, fix-up in workgroup order)
            nrm = torch.rand(B, M, device=dev) + 0.5
            gl = torch.ones(1, device=dev)
            dp = torch.empty(B, Cs, ldm, device=dev)
            bws = torch.empty(max(1, lib.skd_pairwise_backward_workspace_floats(B, Cs, M)), device=dev)
            bcall = lambda: lib.skd_pairwise_backward(B, Cs, M, ldm, fs.data_ptr(), G.data_ptr(), nrm.data_ptr(), gl.data_ptr(),
                                                      dp.data_ptr(), bws.data_ptr(), st)
            bms, bmn, bmx = timed(bcall)
            fb = 2.0 * B * ldm * ldm * Cs
            out["M=%d" % M]["backward"] = {"us": round(bms * 1e3, 1), "us_min": round(bmn * 1e3, 1), "us_max": round(bmx * 1e3, 1),
                                           "executed_TFLOPs": round(rate(fb, bms), 2),
                                           "frac_fp32_mfma_executed": round(rate(fb, bms) / MFMA_F32_PEAK_TFLOPS, 4),
                                           "note": "whole skd_pairwise_backward call: node-major copy + stream-K GEMM + fix-up of the shared tiles"}
    return out


def abn_pmc_ratio(case_prefix):
    """HBM traffic / algorithmic bytes of an ABN entry from the committed rocprofv3 counter passes (profiles/*_pmc.json,
    written by tools/summarise_pmc.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of tools/kernel_microbench.py;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950), byte-weighted over the
    microbench cases whose name starts with ``case_prefix``.  PMC counters cannot be collected from inside this process,
    so the live line carries the profiled ratio and says where it is from."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            cases = [c for c in json.load(open(path)).get("cases", []) if c["name"].startswith(case_prefix)
                     and c.get("hbm_over_algorithmic")]
            if cases:
                algo = sum(c["algo_bytes"] for c in cases)
                hbm = sum(c["hbm_over_algorithmic"] * c["algo_bytes"] for c in cases)
                return {"ratio": round(hbm / algo, 4),
                        "source": os.path.relpath(path, ROOT) + " ((2*FETCH_SIZE + WRITE_SIZE) / algorithmic bytes over the %d "
                                  "'%s*' cases of tools/kernel_microbench.py)" % (len(cases), case_prefix)}
        except Exception:
            continue
    return None


def summarise_gemm(recs):
    """[(ms, (M, K, N, x))] of the fused 1x1-convolution GEMM -> TFLOP/s over all launches (sum of 2*M*K*N / sum of time)
    and per problem shape."""
    if not recs:
        return None
    tot_ms = sum(ms for ms, _ in recs)
    tot_f = sum(2.0 * d[0] * d[1] * d[2] for _, d in recs)
    shapes = {}
    for ms, d in recs:
        e = shapes.setdefault((d[0], d[1], d[2]), [0, 0.0])
        e[0] += 1
        e[1] += ms
    return {"launches": len(recs), "total_ms": round(tot_ms, 3), "avg_us": round(1e3 * tot_ms / len(recs), 2),
            "achieved_TFLOPs": round(tot_f / (tot_ms * 1e-3) / 1e12, 2),
            # per shape both roofs: the GEMM reads X (M x K) + the residual (M x N) and writes Y (M x N); at K <= 128 the
            # arithmetic intensity (2 K N / (4 K + 8 N) flop per byte) is below the chip's ridge (157.3 TF / 6.3 TB/s = 25), i.e.
            # those launches are HBM-bound and their MFMA fraction says nothing
            "per_shape": {"M=%d K=%d N=%d" % k: {"launches": n, "avg_us": round(1e3 * ms / n, 2),
                                                  "TFLOPs": round(2.0 * k[0] * k[1] * k[2] / (ms / n * 1e-3) / 1e12, 2),
                                                  "algorithmic_GBs": round(4.0 * k[0] * (k[1] + 2 * k[2]) / (ms / n * 1e-3) / 1e9, 1),
                                                  "flop_per_byte": round(2.0 * k[1] * k[2] / (4.0 * (k[1] + 2 * k[2])), 1),
                                                  "bound": "hbm" if 2.0 * k[1] * k[2] / (4.0 * (k[1] + 2 * k[2])) < 25.0 else "mfma"}
                          for k, (n, ms) in shapes.items()}}


def gemm_pmc_traffic():
    """HBM bytes per launch of conv1x1_abn_kernel on the layer-3 problem (M = 33800, K = 256, N = 1024) from the committed counter
    passes over tools/gemm_lab (profiles/*_gemm_lab_pmc.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, summarised by
    tools/summarise_lab_pmc.py).  The problem is recognised by what the launch WRITES (M * N * 4 = 138.4 MB: the grid size depends on
    the device's compute-unit count since the half-height tiles of round 6); the prologue + residual form when it was profiled."""
    import glob
    want_write = 33800 * 1024 * 4 / 1e6
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gemm_lab_pmc.json")), reverse=True):
        try:
            rows = [r for r in json.load(open(path)) if r["kernel"].startswith("conv1x1_abn_kernel<3, true") and "hbm_read_MB" in r
                    and abs(r.get("hbm_write_MB", 0.0) - want_write) <= 0.01 * want_write and 150.0 <= r["hbm_read_MB"] <= 500.0]
            rows.sort(key=lambda r: r["kernel"].startswith("conv1x1_abn_kernel<3, true, true"), reverse=True)
            if rows:
                r = rows[0]
                return {"MB": round(r["hbm_read_MB"] + r["hbm_write_MB"], 2),
                        "source": os.path.relpath(path, ROOT) + " (" + r["kernel"] + ": 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes over "
                                  "tools/gemm_lab; algorithmic bytes of that launch: 312.5 MB)"}
        except Exception:
            continue
    return None


def summarise(recs, bytes_per_elem, nhwc=False):
    """[(ms, first four C-ABI args)] -> achieved GB/s over all launches (sum of algorithmic bytes / sum of time).
    NCHW entries: args = (N, C, S, ...); skd_abn_apply_nhwc: args = (rows, C, x, residual) with 8 B/element, 12 with
    a residual."""
    if not recs:
        return None
    if nhwc == "apply":
        recs = [(ms, (d[0], d[1], 1.5 if d[3] else 1.0)) for ms, d in recs]      # third factor scales 8 -> 12 B
    elif nhwc:
        recs = [(ms, (d[0], d[1], 1.0)) for ms, d in recs]
    else:
        recs = [(ms, (d[0], d[1], d[2])) for ms, d in recs]
    tot_ms = sum(ms for ms, _ in recs)
    tot_b = sum(bytes_per_elem * d[0] * d[1] * d[2] for _, d in recs)
    big = [(ms, d) for ms, d in recs if d[0] * d[1] * d[2] >= (1 << 22)]
    out = {"launches": len(recs), "avg_us": round(1e3 * tot_ms / len(recs), 2),
           "avg_elems": round(tot_b / bytes_per_elem / len(recs)),
           "achieved_GBs": round(tot_b / (tot_ms * 1e-3) / 1e9, 1)}
    if big:
        bms = sum(ms for ms, _ in big)
        bb = sum(bytes_per_elem * d[0] * d[1] * d[2] for _, d in big)
        out["achieved_GBs_large"] = round(bb / (bms * 1e-3) / 1e9, 1)   # launches >= 4M elements
    return out


# ---- N > 1: the first multi-GPU run must produce a number, whatever happens -------------------------------
# The cross-replica InPlace-ABN exchange has three forms, fastest first; every one computes the reference's combine rule
# (libs/functions.py:183-218, 257-294) bit for bit.  The warm-up steps run under a SHORT in-kernel wait limit; after each of them
# the ranks agree (one all-reduce) whether anybody saw a device status word / a non-finite loss, and if so ALL of them drop the
# model and the mailboxes and start again in the next, safer form.  The JSON line says which form ran and why.
# (form: name, sync_fused library state to set -- None = leave as configured --, environment read by the PYTHON side only)
COMM_FORMS = (
    ("as configured", None, {}),
    ("three launches per pass over the ipc mailboxes", False, {}),
    ("torch.distributed collectives", False, {"SKD_SYNC_IPC": "0"}),
)


def _effective_form():
    from structure_knowledge_distillation_amd.utils import parallel as P
    return (os.environ.get("SKD_SYNC_IPC", "1") == "1", P.sync_fused())


def warm_up_with_fallback(build, warmup, world, dev, warm_timeout_s=30.0, run_timeout_s=120.0, forms=COMM_FORMS):
    """``build()`` -> (model, step) under the CURRENT environment; runs ``warmup`` untimed steps and returns
    (model, step, info) with info = {"form", "attempts", "fallback_reason"}.  world == 1: a plain warm-up."""
    import math
    import torch
    import torch.distributed as dist
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    if not P.replicated():
        model, step = build()
        for i in range(warmup):
            step(i)
        return model, step, None
    on_gpu = torch.device(dev).type == "cuda"
    host_agree = dist.get_backend() != "nccl"
    reasons, tried = [], set()
    for name, fused, env in forms:
        os.environ.update(env)
        if fused is not None:
            P.set_sync_fused(fused)                    # library state (include/skd.h section 13), not the process environment
        if _effective_form() in tried:
            continue                                   # e.g. the configured form already is the three-launch one
        tried.add(_effective_form())
        os.environ["SKD_SYNC_TIMEOUT_S"] = str(warm_timeout_s)      # read when the mailboxes are set up (first synchronised layer)
        model, step = build()
        failed = None
        for i in range(warmup):
            err = None
            try:
                losses = step(i)
                if not all(math.isfinite(float(v)) for v in losses):
                    err = "non-finite loss in warm-up step %d: %s" % (i, [float(v) for v in losses])
            except _lib.SkdDeviceError as e:           # raised where a logged scalar is read: every collective of the step has been issued
                err = str(e)[:300]
            if on_gpu:
                torch.cuda.synchronize()
            words = _lib.device_status()
            if words and any(words):
                _lib.get().skd_status_clear()
                err = err or "device status words %s in warm-up step %d" % (["0x%08x" % w for w in words], i)
            flag = torch.tensor([1.0 if err else 0.0], device="cpu" if host_agree else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)             # same program point on every rank: nobody is left inside a collective
            if float(flag) > 0:
                failed = err or "a peer rank reported a failure in warm-up step %d" % i
                break
        if failed is None:
            P.SyncMailbox.set_timeout_all(run_timeout_s)
            os.environ["SKD_SYNC_TIMEOUT_S"] = str(run_timeout_s)
            return model, step, {"form": P.comm_form(), "attempts": len(tried),
                                 "fallback_reason": "; then ".join(reasons) if reasons else None}
        reasons.append("'%s' (%s): %s" % (name, P.comm_form(), failed))
        del model, step
        if on_gpu:
            torch.cuda.synchronize()
        P.SyncMailbox.reset()
        _lib.get().skd_status_clear()
        if on_gpu:
            torch.cuda.empty_cache()
        dist.barrier()
    raise SystemExit("bench.py: every form of the cross-replica exchange failed its warm-up: " + "; then ".join(reasons))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(a):
    """``python bench.py --gpus N`` (N > 1) started WITHOUT torchrun -- the N = 1 command with the number changed, which is how a
    driver would start it: re-run this very command line under ``python -m torch.distributed.run`` (one process per GPU, 127.0.0.1
    rendezvous on a free port), let rank 0's ONE JSON line through on the inherited stdout and return the launcher's exit code.
    With fewer than N devices (and no SKD_DIST_BACKEND=gloo device sharing) nothing is started: a JSON line with "error" and rc 2."""
    import subprocess
    import torch
    ndev = torch.cuda.device_count() if (a.device == "cuda" and torch.cuda.is_available()) else 0
    shared = os.environ.get("SKD_DIST_BACKEND") == "gloo"      # N ranks on fewer devices: tests / 1-GPU boxes only (init_distributed)
    if a.device == "cuda" and (ndev == 0 or (ndev < a.gpus and not shared)):
        print(json.dumps({"metric": "distillation-step images/sec @512x512 (Pi+Pa+Ho)", "value": None, "unit": "images/sec",
                          "n_gpus": a.gpus, "error": "--gpus %d asked for, %d GPU(s) visible to this process (torch.cuda.device_count()); "
                                                     "one process per GPU is the only multi-GPU form" % (a.gpus, ndev)}), flush=True)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.environ.get("SKD_BENCH_ENTRY") or os.path.abspath(__file__)] + sys.argv[1:]     # (the tests' wrapper names itself there)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on these hosts (RCCL, the SyncABN mailboxes)
    env.setdefault("OMP_NUM_THREADS", "8")                      # torchrun would set 1 and say so on stderr
    sys.stderr.write("bench.py: --gpus %d without torchrun: starting %s\n" % (a.gpus, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    import torch
    import torch.distributed as dist
    from structure_knowledge_distillation_amd import _lib
    from structure_knowledge_distillation_amd.utils import parallel as P
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args

    # a benchmark run must fail fast: an exchange that waits for a peer longer than this poisons its outputs and raises
    # SkdDeviceError at the next step (training keeps the library's 600 s, what torch.distributed would have waited); during the
    # warm-up the limit is shorter still (warm_up_with_fallback)
    run_timeout = float(os.environ.get("SKD_SYNC_TIMEOUT_S", "120"))
    os.environ.setdefault("SKD_DIST_TIMEOUT_S", "300")       # torch.distributed collectives: same idea (utils/parallel.init_distributed)
    rank, world, local = P.init_distributed()
    if world != a.gpus:
        a.gpus = world                                       # torchrun's world is authoritative
    # the N > 1 FORM of the step: more than one rank, or SKD_DIST_SOLO=1's one-rank communicator (hardware rehearsal of the RCCL
    # plumbing on a 1-GPU box: its line is marked "rehearsal" and measures nothing about scaling)
    multi = P.replicated()
    if a.device == "cpu":
        # launcher rehearsal (tests/test_distributed_cpu.py): the test's sitecustomize has installed a C-ABI double; without one
        # there is nothing to run -- this package has no CPU path
        if not _lib.test_backend_active():
            raise SystemExit("bench.py --device cpu: no C-ABI test double installed (rehearsal mode of the tests only)")
        dev = torch.device("cpu")
        a.no_kernel_timing = a.no_pairwise_sweep = a.no_cpu_baseline = True
        torch.cuda.synchronize = lambda *x, **k: None
    else:
        assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
        _lib.load()
        dev = torch.device("cuda", torch.cuda.current_device())
    # args.batch_size is the reference's GLOBAL batch (nn.DataParallel scatters it); every rank holds a.batch images
    terms = set(a.losses.split(","))
    assert terms <= {"pi", "pa", "ho"}, a.losses
    args = default_args(batch_size=a.batch * world, device=dev, weight_decay=5e-4, lambda_pa=0.5, num_steps=40000,
                        pi="pi" in terms, pa="pa" in terms, ho="ho" in terms)
    gen = torch.Generator().manual_seed(100 + rank)
    images = (torch.randn(a.batch, 3, a.size, a.size, generator=gen) * 57.0).to(dev)
    labels = torch.randint(0, 19, (a.batch, a.size, a.size), generator=gen)
    labels[0, : a.size // 16] = 255
    labels = labels.to(dev)
    data = (images, labels, None, None)

    def build():
        torch.manual_seed(1234)
        model = NetModel(args)

        def step(i):
            model.adjust_learning_rate(args.lr_g, model.G_solver, i)
            model.adjust_learning_rate(args.lr_d, model.D_solver, i)
            model.set_input(data)
            model.optimize_parameters()
            return (model.G_loss, model.mc_G_loss, model.pi_G_loss, model.pa_G_loss, model.D_loss)  # print_info's reads
        return model, step

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # W untimed warm-up steps; N > 1: under a short in-kernel wait limit, with a collective verdict after every step and a
    # re-initialisation in the next safer form of the SyncABN exchange if any rank saw a device error (the line says which form ran)
    model, step, comm_setup = warm_up_with_fallback(build, a.warmup, world, dev, run_timeout_s=run_timeout)
    timed = ["skd_abn_apply_nhwc", "skd_abn_apply", "skd_abn_apply_residual", "skd_abn_forward_train", "skd_abn_backward",
             "skd_abn_forward_train_to", "skd_abn_relu_backward_reduce", "skd_abn_relu_backward_dx",
             "skd_abn_forward_train_nhwc", "skd_abn_backward_reduce_nhwc", "skd_abn_backward_dx_nhwc",
             "skd_abn_relu_backward_reduce_nhwc", "skd_abn_relu_backward_dx_nhwc",
             "skd_abn_relu_backward_reduce_nhwc_x", "skd_abn_relu_backward_dx_nhwc_x", "skd_abn_backward_nhwc",
             "skd_abn_relu_backward_nhwc",
             # round 5: the kernels rewritten this round, as they run inside the step
             "skd_ce_dsn_forward", "skd_maxpool_argmax_nhwc", "skd_maxpool3x3s2_backward_nhwc",
             # round 6: the training stem fused (bn3 -> relu3 -> max-pool without the normalised tensor)
             "skd_abn_relu_maxpool3x3s2_nhwc", "skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc", "skd_abn_relu_maxpool3x3s2_backward_dx_nhwc"]
    # Inside the timed region only the ROOFLINE entry is bracketed with HIP events (111 calls per step; bracketing all
    # ~700 hand-written calls costs 1.4 ms = 1.8 % of the step -- measured, profiles/r02 notes); the table of the other
    # kernels is collected in three extra, untimed steps afterwards (single-rank runs only: every rank must step).
    roofline_entry = "skd_abn_apply_nhwc"
    gemm_entry = "skd_conv1x1_abn_pro_nhwc"
    # (with the teacher on its own stream -- the default at N = 1 -- its kernels share the chip with the student's forward in the timed
    # steps: like the graph replay, they are timed in the extra serial-teacher steps below instead)
    teacher_beside = getattr(model, "_teacher_stream", None) is not None
    if not a.no_kernel_timing and rank == 0 and not teacher_beside:
        _lib.enable_kernel_timing([roofline_entry, gemm_entry])
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        losses = step(a.warmup + i)
    fence()
    el = time.perf_counter() - t0
    recs = _lib.disable_kernel_timing() if (not a.no_kernel_timing and rank == 0 and not teacher_beside) else {}
    roofline_steps = None
    if not a.no_kernel_timing and (getattr(model, "_teacher_graph_on", False) or teacher_beside):
        # The roofline kernel lives in the frozen teacher, and in the timed steps the teacher runs on its own stream beside the
        # student's forward (SKD_TEACHER_STREAM, default at N = 1: its kernels share the chip) or is ONE hipGraph replay
        # (SKD_TEACHER_STREAM=0: no host-side launch to put HIP events around).  So the same kernels are timed in extra steps
        # right after the timed region with the teacher issued eagerly on the main stream -- same process, shapes, weights, D step
        # on its own stream as in the timed region; rocprofv3's per-kernel averages of the same command (profiles/) cover both
        # kinds of step.  `value` / `ms_per_step` come from the timed region only.
        graph_was_on, teacher_stream = model._teacher_graph_on, model._teacher_stream
        model._teacher_graph_on, model._teacher_stream = False, None      # eager teacher on the main stream
        step(a.warmup + a.steps)                         # one untimed eager step (the eager path was last run during capture)
        if rank == 0:
            _lib.enable_kernel_timing([roofline_entry, gemm_entry])
        fence()
        roofline_steps = min(a.steps, 5)
        for i in range(roofline_steps):
            step(a.warmup + a.steps + 1 + i)
        fence()
        if rank == 0:
            recs = _lib.disable_kernel_timing()
        model._teacher_graph_on, model._teacher_stream = graph_was_on, teacher_stream
    comm = None
    if not a.no_kernel_timing:
        # kernel rates are a statement about the kernel, so these three steps run the D step serially (in the timed steps
        # it shares the chip with the student's backward passes on a second stream, which stretches both).  Every rank
        # steps (the collectives need all of them); rank 0 brackets its hand-written kernels, and -- N > 1 -- the spans its
        # compute stream spends blocked on collectives (P.comm_timer: SyncABN exchanges, gradient all-reduce waits).
        d_stream, model._d_stream = model._d_stream, None
        t_stream, model._teacher_stream = model._teacher_stream, None
        if rank == 0:
            _lib.enable_kernel_timing([n for n in timed if n not in (roofline_entry, gemm_entry)])
            if multi:
                P.comm_timer.enable()
                forms0 = _lib.sync_form_counts()
        for i in range(3):
            step(a.warmup + a.steps + i)
        if rank == 0:
            recs.update(_lib.disable_kernel_timing())
            if multi:
                spans = P.comm_timer.disable()
                sa, wa = spans.get("syncabn", (0.0, 0)), spans.get("allreduce_wait", (0.0, 0))
                sf = spans.get("syncabn_fused", (0.0, 0))
                nb = len(model._s_reducer.buckets) + len(model._d_reducer.buckets)
                mb = sum(b.flat.numel() * 4 for r in (model._s_reducer, model._d_reducer) for b in r.buckets) / 1e6
                comm = {"form": comm_setup["form"], "fallback_reason": comm_setup["fallback_reason"], "forms_tried": comm_setup["attempts"],
                        "syncabn_ms": round(sa[0] / 3, 3), "syncabn_collectives": (sa[1] + sf[1]) // 3,
                        # exchanges performed INSIDE the library's synchronised ABN calls (one register-resident launch when
                        # the tensor fits); the span is the whole pass (statistics + exchange + normalise), not the exchange alone
                        "syncabn_in_abn_calls": sf[1] // 3, "abn_sync_call_ms": round(sf[0] / 3, 3),
                        "syncabn_one_launch_calls": (_lib.sync_form_counts()[0] - forms0[0]) // 3,   # of those: exchange INSIDE the one launch
                        "allreduce_wait_ms": round(wa[0] / 3, 3), "buckets": nb, "gradient_MB": round(mb, 1),
                        "backend": dist.get_backend(),
                        "syncabn_transport": ("ipc mailboxes, one kernel per exchange (csrc/sync.hip)"
                                              if P.SyncMailbox.active() else "torch.distributed all_gather / all_reduce"),
                        "note": "per step, from 3 extra untimed steps with the D step serial: time rank 0's compute stream was "
                                "blocked in the SyncABN exchanges (incl. waiting for the slowest rank) and in GradientAllReducer.finish() waits"}
        model._d_stream, model._teacher_stream = d_stream, t_stream
    if multi:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t)
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return
    ms = 1e3 * el / a.steps
    value = a.batch * world * a.steps / el
    line = {
        "metric": "distillation-step images/sec @512x512 (Pi+Pa+Ho)", "value": round(value, 3), "unit": "images/sec",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]%s: ResNet18-PSPNet student + frozen ResNet101-PSPNet teacher, "
                               "%s, batch %d per GPU, %dx%d, 19 classes"
                               % (" x %d ranks (configs[3])" % world if world > 1 else "",
                                  "Pi+Pa+Ho (SAGAN D, spectral norm, WGAN-GP)" if terms == {"pi", "pa", "ho"} else
                                  "+".join(t.capitalize() for t in ("pi", "pa", "ho") if t in terms) + " ONLY (not the BASELINE configuration)",
                                  a.batch, a.size, a.size),
                   "global_batch": a.batch * world, "parallelism": "dp%d" % world,
                   "losses_last_step": {k: round(float(v), 6) for k, v in
                                        zip(("G", "mc", "pi", "pa", "D"), losses)}},
    }
    from structure_knowledge_distillation_amd.networks import pspnet_combine as _PC
    executed = STEP_TFLOP_PER_IMAGE - (FOLD_TFLOP_PER_IMAGE if _PC.PSP_FOLD else 0.0)
    line["step_tflop_per_image"] = {"reference_algorithm": STEP_TFLOP_PER_IMAGE, "executed": round(executed, 4)}
    line["step_fp32_mfma_frac"] = round(value / world * executed / MFMA_F32_PEAK_TFLOPS, 4)   # executed flops only
    ap = summarise(recs.get(roofline_entry, []), 8, nhwc="apply")
    gm = summarise_gemm(recs.get(gemm_entry, []))
    if gm and (not ap or gm["total_ms"] > ap["launches"] * ap["avg_us"] * 1e-3):
        # the dominant hand-written kernel since round 3: the frozen teacher's bottleneck tail as ONE fp32-MFMA GEMM
        line["roofline"] = {"kernel": "conv1x1_abn_kernel<relu, residual, prologue> (skd_conv1x1_abn_pro_nhwc: relu(bn2(x)) applied to the A "
                                      "operand on the way into LDS, 1x1 convolution on v_mfma_f32_32x32x2_f32, bn3 + residual + ReLU in the "
                                      "epilogue; algorithmic flops 2*M*K*N per launch, M = B*H*W)",
                            "bound": "mfma", "achieved": gm["achieved_TFLOPs"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(gm["achieved_TFLOPs"] / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None, "detail": gm}
        line["roofline"]["measured_in"] = ("the %d timed steps" % a.steps if roofline_steps is None else
                                           "%d extra steps right after the timed region with the teacher issued eagerly on the main stream (in the "
                                           "timed steps the teacher forward shares the chip with the student's forward on a stream of its own -- or, "
                                           "SKD_TEACHER_STREAM=0, is one hipGraph replay without host-side launches to bracket with HIP events); same "
                                           "kernels, shapes, weights" % roofline_steps)
        pmc = gemm_pmc_traffic()
        if pmc is not None:
            line["roofline"]["traffic"] = pmc["MB"]
            line["roofline"]["traffic_unit"] = "MB per launch of the layer-3 problem (M=33800, K=256, N=1024): HBM read + write"
            line["roofline"]["traffic_source"] = pmc["source"]
        if ap:
            line["roofline"]["abn_apply_nhwc (previous dominant kernel, hbm-bound)"] = {
                "achieved_GBs": ap["achieved_GBs"], "frac_hbm": round(ap["achieved_GBs"] / HBM_PEAK_GBS, 4), "detail": ap}
    elif ap:
        # the dominant hand-written kernel of the step (3.9 of ~7 ms of InPlace-ABN time): ALL its launches of the timed
        # region, algorithmic bytes = 8 per element (12 with the residual read) x elements of the launch
        line["roofline"] = {"kernel": "abn_apply_nhwc_kernel (skd_abn_apply_nhwc: the frozen teacher's eval-mode InPlace-ABN + ReLU "
                                      "[+ residual], channels-last, in place; 8 algorithmic bytes per element, 12 with the residual read)",
                            "bound": "hbm", "achieved": ap["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ap["achieved_GBs"] / HBM_PEAK_GBS, 4), "traffic": None,
                            "algorithmic_bytes_per_element": "8 (12 with residual)", "detail": ap}
        pmc = abn_pmc_ratio("apply_nhwc")
        if pmc is not None:
            line["roofline"]["traffic"] = round(pmc["ratio"] * 8 * ap["avg_elems"] / 1e6, 2)   # avg_elems is byte-weighted
            line["roofline"]["traffic_unit"] = "MB per average launch"
            line["roofline"]["traffic_source"] = pmc["source"]
    if ap or gm:
        line["kernels"] = {
            "skd_abn_backward_nhwc (leaky ABN, reduce + dx in one call: 20 B/elem two-pass algorithmic, 12 B/elem when register-resident)": summarise(recs.get("skd_abn_backward_nhwc", []), 20, nhwc="train"),
            "skd_abn_relu_backward_nhwc (BN+ReLU[+res], reduce + dx in one call: >=20 B/elem two-pass algorithmic)": summarise(recs.get("skd_abn_relu_backward_nhwc", []), 20, nhwc="train"),
            "skd_abn_apply_residual (teacher block tails, 12 B/elem)": summarise(recs.get("skd_abn_apply_residual", []), 12),
            "skd_abn_forward_train (leaky-ReLU ABN: stats+finalize+apply, 12 B/elem)": summarise(recs.get("skd_abn_forward_train", []), 12),
            "skd_abn_backward (leaky-ReLU ABN: reduce+finalize+dx, 20 B/elem)": summarise(recs.get("skd_abn_backward", []), 20),
            "skd_abn_forward_train_to (BN+ReLU[+res] fused: stats+finalize+apply, >=12 B/elem)": summarise(recs.get("skd_abn_forward_train_to", []), 12),
            "skd_abn_relu_backward_reduce (12 B/elem)": summarise(recs.get("skd_abn_relu_backward_reduce", []), 12),
            "skd_abn_relu_backward_dx (>=16 B/elem)": summarise(recs.get("skd_abn_relu_backward_dx", []), 16),
            "skd_abn_forward_train_nhwc (channels-last student: stats+finalize+apply, >=12 B/elem)": summarise(recs.get("skd_abn_forward_train_nhwc", []), 12, nhwc="train"),
            "skd_abn_relu_backward_reduce_nhwc (12 B/elem)": summarise(recs.get("skd_abn_relu_backward_reduce_nhwc", []), 12, nhwc="train"),
            "skd_abn_relu_backward_dx_nhwc (>=16 B/elem)": summarise(recs.get("skd_abn_relu_backward_dx_nhwc", []), 16, nhwc="train"),
            "skd_abn_relu_backward_reduce_nhwc_x (no residual: mask from x, 8 B/elem)": summarise(recs.get("skd_abn_relu_backward_reduce_nhwc_x", []), 8, nhwc="train"),
            "skd_abn_relu_backward_dx_nhwc_x (no residual: mask from x, 12 B/elem)": summarise(recs.get("skd_abn_relu_backward_dx_nhwc_x", []), 12, nhwc="train"),
            "skd_abn_backward_reduce_nhwc (leaky ABN, 8 B/elem)": summarise(recs.get("skd_abn_backward_reduce_nhwc", []), 8, nhwc="train"),
            "skd_abn_backward_dx_nhwc (leaky ABN, 12 B/elem)": summarise(recs.get("skd_abn_backward_dx_nhwc", []), 12, nhwc="train"),
        }
        def plain(name, nbytes, what, bound):
            r = recs.get(name, [])
            if not r:
                return None
            tot_ms = sum(ms for ms, _ in r)
            tot_b = sum(nbytes(d) for _, d in r)
            # avg_elems = 0 keeps a kernel that is not HBM-bound out of the "worst HBM fraction" pick below
            return {"launches": len(r), "avg_us": round(1e3 * tot_ms / len(r), 2), "avg_elems": round(tot_b / 4 / len(r)) if bound == "hbm" else 0,
                    "achieved_GBs": round(tot_b / (tot_ms * 1e-3) / 1e9, 1), "bound": bound, "algorithmic_bytes": what}
        line["kernels"].update({
            "skd_ce_dsn_forward (fused upsample + CE of both heads + gradients, csrc/ce_dsn.hip)": plain(
                "skd_ce_dsn_forward", lambda d: 8.0 * d[0] * a.size * a.size + 16.0 * d[0] * d[1] * d[2] * d[3],
                "int64 target + 2 x (logits in + gradients out): 27.05 MB at batch 8", "valu (19-class softmax per up-sampled pixel, exp / log)"),
            "skd_maxpool_argmax_nhwc (pair-wise pooling of the channels-last PSP features)": plain(
                "skd_maxpool_argmax_nhwc", lambda d: 4.0 * d[0] * d[1] * d[2] * d[3], "one read of the feature map", "hbm"),
            "skd_maxpool3x3s2_backward_nhwc (stem max-pool backward)": plain(
                "skd_maxpool3x3s2_backward_nhwc", lambda d: 4.0 * d[0] * d[1] * d[2] * d[3] + 5.0 * d[0] * d[1] * (d[2] // 2 + 1) * (d[3] // 2 + 1),
                "4 B per input element out + 5 B per output element in", "hbm"),
            "skd_abn_relu_maxpool3x3s2_nhwc (student stem: normalise + ReLU + max-pool in one pass, round 6)": plain(
                "skd_abn_relu_maxpool3x3s2_nhwc", lambda d: 4.0 * d[0] * d[1] * d[2] * d[3] + 5.0 * d[0] * d[1] * (d[2] // 2 + 1) * (d[3] // 2 + 1),
                "4 B per input element in + 5 B per pooled element out (the normalised tensor is never written)", "hbm"),
            "skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc (student stem: edz / eydz through the argmax bytes)": plain(
                "skd_abn_relu_maxpool3x3s2_backward_reduce_nhwc",
                lambda d: 4.0 * d[0] * d[1] * d[2] * d[3] + 5.0 * d[0] * d[1] * (d[2] // 2 + 1) * (d[3] // 2 + 1),
                "4 B per input element in + 5 B per pooled element in", "hbm"),
            "skd_abn_relu_maxpool3x3s2_backward_dx_nhwc (student stem: dx, the un-pooled gradient never exists)": plain(
                "skd_abn_relu_maxpool3x3s2_backward_dx_nhwc",
                lambda d: 8.0 * d[0] * d[1] * d[2] * d[3] + 5.0 * d[0] * d[1] * (d[2] // 2 + 1) * (d[3] // 2 + 1),
                "4 B per input element in + 4 out + 5 B per pooled element in", "hbm"),
        })
        line["kernels"] = {k: v for k, v in line["kernels"].items() if v}
        line["kernels_note"] = "HIP-event rates from three extra untimed steps run with the D step serial (no co-running stream)"
        if line["kernels"]:
            worst = min(line["kernels"].items(), key=lambda kv: kv[1]["achieved_GBs"] if kv[1]["avg_elems"] >= (1 << 20) else 1e9)
            line["roofline"]["worst_other_kernel"] = {"entry": worst[0], "achieved_GBs": worst[1]["achieved_GBs"],
                                                      "frac": round(worst[1]["achieved_GBs"] / HBM_PEAK_GBS, 4)}
    if comm is None and comm_setup is not None:          # --no-kernel-timing: still say which form of the exchange ran
        comm = {"form": comm_setup["form"], "fallback_reason": comm_setup["fallback_reason"], "forms_tried": comm_setup["attempts"]}
    if comm is not None:
        comm["backend"] = dist.get_backend()
        comm["ranks"] = dist.get_world_size()            # what the process group (RCCL on a GPU node) reports, not the flag
        line["comm"] = comm
    if a.device == "cpu":
        line["rehearsal"] = "launcher rehearsal on the C-ABI double (tests only): NOT a measurement"
    elif multi and world == 1:
        line["rehearsal"] = ("SKD_DIST_SOLO=1: the N > 1 form of the step (hooks, buckets, synchronised ABN over torch.distributed, eager "
                             "teacher) on a communicator of ONE rank -- plumbing evidence, NOT a scaling measurement")
    if world == 1 and not a.no_pairwise_sweep:
        line["pairwise_gram_mfma"] = pairwise_sweep(dev)
    if not a.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(a.cpu_baseline_seconds, a.size)
    if world == 1 and a.dsn_ab:
        # INFORMATIVE, never `value`: the same step with the teacher's deep-supervision head not computed.  Its only consumer is the
        # teacher's own CE, which the reference computes and discards (kd_model.py:129) and this package does not compute; the head
        # itself IS part of the benched configuration above because the reference's forward runs it.
        keep = getattr(model.teacher, "skip_dsn", False)
        try:
            model.teacher.skip_dsn = True
            model._teacher_graphs.clear()
            for i in range(3):
                step(i)
            n2 = min(a.steps, 10)
            fence()
            t1 = time.perf_counter()
            for i in range(n2):
                step(3 + i)
            fence()
            el2 = time.perf_counter() - t1
            line["informative_teacher_dsn_head_skipped"] = {
                "ms_per_step": round(1e3 * el2 / n2, 3), "images_per_sec": round(a.batch * n2 / el2, 3), "steps": n2,
                "note": "model.teacher.skip_dsn = True: 0.32 TFLOP per step of dead work less (the head's output feeds only the teacher CE the reference "
                        "discards); losses, gradients and updates are unchanged; NOT the benched configuration"}
        except Exception as e:                       # an extra must never cost the line
            line["informative_teacher_dsn_head_skipped"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        finally:
            model.teacher.skip_dsn = keep
            model._teacher_graphs.clear()
    print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
