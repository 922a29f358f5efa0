"""Data parallelism for the distillation step on one 8x MI355X node.

The reference (utils/parallel.py:66-214) is single-process ``nn.DataParallel``: every forward
re-broadcasts all parameters (282 MB teacher + 52 MB student), scatters the batch,
runs one Python thread per GPU and reduces gradients / losses to GPU 0.  Here the layout is one
process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests):

  * replicas are persistent -- nothing is broadcast per step (C1, C3 removed);
  * each rank runs the step on its own shard of the minibatch; losses stay local (C8 removed);
  * gradients are averaged with bucketed asynchronous all-reduces launched from
    post-accumulate-grad hooks while backward is still running (C2), which reproduces the
    reference's ``Reduce(...)/len(outputs)`` mean over shards (parallel.py:155);
  * InPlaceABNSync exchanges its statistics through the same process group (libs/inplace_abn.py).

``DataParallelModel`` / ``my_DataParallelCriterion`` keep the reference's call signatures
(``model(inputs, parallel=...)``, ``criterion(inputs, *targets, is_target_scattered=...)``) as thin
shims so networks/kd_model.py reads like the original.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

__all__ = ["DataParallelModel", "my_DataParallelCriterion", "DataParallelCriterion", "GradientAllReducer",
           "init_distributed", "world_size", "rank", "replicated", "solo_rehearsal", "ranks_on_this_device", "broadcast_module", "per_rank_batch", "set_replica_batch",
           "clear_replica_batch", "replica_weights", "SyncMailbox", "device_identity", "reserve_for_collectives", "reserved_fused_cap", "sync_fused_over_rccl", "comm_form"]


def init_distributed(backend=None):
    """Join the process group described by torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world_size, local_rank); a no-op for WORLD_SIZE <= 1 (unless SKD_DIST_SOLO=1:
    ``solo_rehearsal``)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        ndev = max(1, torch.cuda.device_count())
        torch.cuda.set_device(local % ndev)   # more ranks than GPUs: ranks share devices
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", ws))
        share_device(-(-local_world // ndev))
    if (ws > 1 or solo_rehearsal()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # SKD_DIST_BACKEND=gloo lets N ranks share ONE GPU (tests, bench.py --gpus N on a 1-GPU box): same code
            # path -- hooks, buckets, SyncABN collectives -- over gloo instead of RCCL
            backend = os.environ.get("SKD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
            # RCCL's kernels run BESIDE the step.  The in-kernel SyncABN exchange (one grid-barrier launch per pass that waits for
            # the peer rank inside the barrier) can close a cross-rank cycle with them (reserve_for_collectives has the argument);
            # the reserve makes that cycle unlikely, not impossible (other kernels may take the reserved units), and no multi-GPU
            # box has ever run it.  Same rule as for the teacher hipGraph at N > 1: until a hardware A/B exists the DEFAULT over
            # RCCL is the three-launch form (statistics / one-workgroup exchange / normalise: no grid barrier, nothing to cycle
            # with, + ~1 ms per step); SKD_ABN_SYNC_FUSED=1 opts into the in-kernel form, and only then are the compute-unit
            # reserve and the RCCL channel cap applied (they cost every collective bandwidth and every ABN pass a
            # quarter of its parallelism, so they must not be process-wide defaults).
            # (library state, not the process environment: ADVICE r05 -- a user's explicit SKD_ABN_SYNC_FUSED is honoured)
            if "SKD_ABN_SYNC_FUSED" not in os.environ:
                set_sync_fused(False)
            if sync_fused_over_rccl():
                os.environ.setdefault("NCCL_MAX_NCHANNELS", "32")
        if os.environ.get("SKD_DIST_TIMEOUT_S"):
            # how long a collective waits for a peer before the job is torn down (torch's defaults: 10 min RCCL, 30 min gloo);
            # bench.py asks for 5 min: a rank that died must fail the measurement, not park seven GPUs until somebody notices
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(os.environ["SKD_DIST_TIMEOUT_S"]))
        dist.init_process_group(backend=backend, rank=rk, world_size=ws, **kw)
        if (backend == "nccl" and torch.cuda.is_available() and sync_fused_over_rccl()
                and int(os.environ.get("LOCAL_WORLD_SIZE", ws)) <= max(1, torch.cuda.device_count())):
            cap = reserve_for_collectives()    # one rank per device over RCCL: its kernels must always find a compute unit
            if rk == 0:
                import logging
                logging.getLogger(__name__).warning(
                    "SKD_ABN_SYNC_FUSED=1 over RCCL: one-launch InPlace-ABN passes capped at %s workgroups (SKD_ABN_RCCL_RESERVE_CUS), "
                    "NCCL_MAX_NCHANNELS=%s", cap, os.environ.get("NCCL_MAX_NCHANNELS"))
    return rk, ws, local


def sync_fused_over_rccl():
    """True when the synchronised InPlace-ABN layers may take the one-launch form (exchange inside the grid-barrier kernel) in a
    job whose collectives are RCCL kernels: opt-in (SKD_ABN_SYNC_FUSED=1), see init_distributed."""
    return sync_fused()


def sync_fused():
    """The library's effective state of the in-kernel exchange switch (include/skd.h section 13: environment default read once,
    overridden by set_sync_fused)."""
    from .. import _lib
    return bool(_lib.get().skd_abn_get_sync_fused())


def set_sync_fused(on):
    """True / False: the synchronised one-launch InPlace-ABN passes exchange inside the kernel / run as three launches; None: back to
    the environment (SKD_ABN_SYNC_FUSED, default on).  Explicit library state -- nothing here touches os.environ, which the native
    library no longer reads on the call path."""
    from .. import _lib
    _lib.get().skd_abn_set_sync_fused(-1 if on is None else int(bool(on)))


def comm_form(group=None):
    """How the cross-replica InPlace-ABN statistics travel right now -- what bench.py prints as ``comm.form``."""
    if not replicated(group):
        return "single rank"
    if not SyncMailbox.active():
        return "torch.distributed all_gather / all_reduce per exchange (SKD_SYNC_IPC=0 or the mailbox self-test failed)"
    from .. import _lib
    if sync_fused() and _lib.get().skd_abn_get_fused():
        return "ipc mailboxes, exchange inside the one-launch ABN kernels where the tensor fits (csrc/abn.hip, sync_dev.hpp)"
    return "ipc mailboxes, three launches per pass: statistics / one-workgroup exchange kernel / normalise (csrc/sync.hip)"


_SHARED = {"ranks": 1}


def ranks_on_this_device():
    """How many ranks of this job run on this process's GPU (1 in production; > 1 only in the two-process tests and
    ``bench.py --gpus N`` on a box with fewer devices): what ``share_device`` was last told."""
    return _SHARED["ranks"]


def share_device(ranks_per_device):
    """Ranks that share ONE device (tests, bench.py --gpus N on a 1-GPU box) must share its compute units: the one-launch
    InPlace-ABN passes hold a grid barrier (and, synchronised, wait for the PEER's launch inside it), so every rank's grid
    is capped at its share -- include/skd.h section 13.  One rank per device (production): no cap beyond the device's own."""
    _SHARED["ranks"] = max(_SHARED["ranks"], int(ranks_per_device))
    if ranks_per_device <= 1 or not torch.cuda.is_available():
        return None
    from .. import _lib
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    # a little under the even share: the other streams of a rank (D step) and the exchange kernels need slots too
    return _lib.get().skd_abn_set_fused_max_workgroups(max(4, (cus - 16) // ranks_per_device))


def reserved_fused_cap(cus, reserve=None):
    """Workgroups a grid-barrier launch may use when RCCL's kernels run beside the step: the device's compute units minus a reserve
    (SKD_ABN_RCCL_RESERVE_CUS, default 64)."""
    if reserve is None:
        reserve = int(os.environ.get("SKD_ABN_RCCL_RESERVE_CUS", "64"))
    return max(4, min(256, int(cus)) - max(0, reserve))


def reserve_for_collectives(device=None):
    """Keep compute units OUT of the one-launch InPlace-ABN passes when the process group's collectives are GPU kernels (RCCL).

    The gradient buckets are all-reduced WHILE backward continues, so an RCCL kernel and a one-launch ABN pass meet on the chip.
    The ABN pass holds a grid barrier -- every workgroup resident until the last one has arrived, one 1024-thread workgroup with
    up to the whole register file of its compute unit (`-Rpass-analysis=kernel-resource-usage`: 128 VGPRs x 4 waves per SIMD) --
    and, synchronised, waits INSIDE that barrier for the peer rank's pass.  With the grid at one workgroup per compute unit that
    closes a cycle across two ranks whose hardware queues picked the two kernels in opposite order: A's ABN pass waits for B's; B's
    cannot become resident because B's RCCL kernel holds compute units; B's RCCL kernel waits for A's RCCL kernel; A's RCCL
    kernel finds no compute unit because A's ABN pass spins on all of them.  (Nothing on a one-GPU box shows this: gloo's
    all-reduce runs on the host.)  With a reserve the collective can always start on every rank, finishes, and frees what the ABN
    pass is waiting for: the cycle cannot close; the price is a quarter of the ABN passes' parallelism (2.4 -> ~3 ms per step)."""
    if not torch.cuda.is_available():
        return None
    from .. import _lib
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    return _lib.get().skd_abn_set_fused_max_workgroups(reserved_fused_cap(cus))


def device_identity(props, index):
    """A string that is equal for two ranks of one host exactly when they sit on the SAME physical GPU: the PCI address (domain :
    bus : device) and the runtime's UUID when it reports a non-zero one; the device index only when neither is available (ranks
    that mask devices with HIP_VISIBLE_DEVICES all call theirs 0).  A wrong "shared" verdict is not an error but costs throughput:
    every rank's grid-barrier launches would be capped at 1 / N of the compute units (share_device)."""
    parts = []
    pci = [getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    if all(isinstance(v, int) for v in pci) and (pci[1] or pci[2] or pci[0]):
        parts.append("pci %04x:%02x:%02x" % tuple(pci))
    uuid = str(getattr(props, "uuid", "") or "")
    if uuid.strip("0-").strip():                       # an all-zero UUID (older runtimes) identifies nothing
        parts.append("uuid " + uuid)
    return " ".join(parts) if parts else "index %d" % (index or 0)


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def solo_rehearsal():
    """SKD_DIST_SOLO=1 (hardware rehearsal, never a training configuration): a process group of ONE rank runs the N > 1 FORM of the
    step -- replica broadcast, gradient hooks + bucketed asynchronous all-reduce, synchronised InPlace-ABN over torch.distributed
    collectives, eager teacher, sharded evaluation -- so that a 1-GPU box exercises that control flow on backend "nccl" (RCCL
    communicator, ProcessGroupNCCL's streams and events, device-side verdict tensors).  A communicator of one rank moves no data
    between devices: the numbers of such a run say nothing about scaling (bench.py marks its line "rehearsal")."""
    return os.environ.get("SKD_DIST_SOLO", "0") == "1"


def replicated(group=None):
    """True when the step must take its multi-replica form: more than one rank in the group, or an initialised group of one rank
    under ``solo_rehearsal``."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or solo_rehearsal()


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def per_rank_batch(global_batch, group=None):
    """The reference's ``--batch-size`` is the GLOBAL minibatch: nn.DataParallel scatters it over the GPUs
    (utils/parallel.py:54-64) and lr_g / lr_d are tuned for it.  With one process per GPU every rank's loader must
    therefore deliver ``batch_size // world_size`` samples (``DistributedSampler(..., drop_last=True)``); use this
    helper so a world size that does not divide the batch is an error instead of a silently different recipe."""
    w = world_size(group)
    if global_batch % w:
        raise ValueError("global batch %d is not divisible by the %d replicas" % (global_batch, w))
    return global_batch // w


_REPLICA = {"weights": None, "key": None}
_REPLICA_BUF = {}


def set_replica_batch(local_batch, device, group=None):
    """Tell InPlaceABNSync how many samples THIS rank holds in the current step.  The reference's combine rule
    (libs/functions.py:196-197) silently assumes equal shards; here the per-rank counts are all-gathered (one tiny
    collective per step, no host sync) and the BN STATISTICS are pooled with weights n_g / sum(n) -- identical to the
    reference when the shards are equal, exact batch statistics when a loader hands out a short last batch.  Only the
    statistics: the LOSS stays the reference's plain mean over replicas of the per-replica losses (Reduce / len(outputs),
    utils/parallel.py:155), so GradientAllReducer averages with 1 / G whatever the shard sizes -- for ragged shards that
    is the reference's own weighting of the samples, not a sample-uniform one.  Valid until ``clear_replica_batch``."""
    if not replicated(group):
        _REPLICA["weights"] = None
        return None
    # No host-to-device copy (a pageable one synchronises the stream: measured on the one-rank rehearsal as ~0.7 ms of main-stream
    # gaps at every step start, profiles/r08f_timeline_solo.md): a persistent one-float buffer filled by a launch
    w = world_size(group)
    key = (str(torch.device(device)), w)
    mine = _REPLICA_BUF.get(key)
    if mine is None:
        mine = _REPLICA_BUF[key] = torch.empty(1, device=device)
    mine.fill_(float(local_batch))
    counts = torch.empty(w, device=device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    _REPLICA["weights"] = counts / counts.sum()
    return _REPLICA["weights"]


def clear_replica_batch():
    """Forget the per-rank sample weights: ``NetModel.optimize_parameters`` calls this when the step is over, so a
    synchronised forward that was NOT announced by ``set_replica_batch`` (another model, a direct forward) pools with the
    reference's equal-shard rule instead of with a previous step's stale weights."""
    _REPLICA["weights"] = None


def replica_weights():
    """(G,) device tensor of n_g / sum(n) set by ``set_replica_batch`` for this step, or None (equal shards)."""
    return _REPLICA["weights"]


class SyncMailbox:
    """Per process group: the IPC mailboxes of csrc/sync.hip (include/skd.h section 12) that carry the cross-replica
    InPlace-ABN statistics in one small kernel per exchange instead of a torch.distributed collective.

    ``SyncMailbox.get(group, device)`` is COLLECTIVE the first time it is called for a group: every rank creates its
    mailbox, the IPC handles (and host names) travel with ``dist.all_gather_object``, every rank opens the others' and the
    ranks run one self-test exchange whose result they check on the host; the verdicts are combined with an all-reduce
    (MIN) so that EITHER every rank uses the mailboxes OR every rank keeps the torch.distributed path -- a rank on another
    host, a failed hipIpcOpenMemHandle or a wrong self-test answer anywhere switches the whole group back.
    SKD_SYNC_IPC=0 keeps torch.distributed (the selectable fallback)."""
    _by_group = {}

    def __init__(self, ctx, lib, world, rank):
        self.ctx, self.lib, self.world, self.rank = ctx, lib, world, rank
        self.max_channels = lib.skd_sync_max_floats() // 2

    @classmethod
    def get(cls, group, device):
        # Keyed by the group's geometry AND validated against the group OBJECT through a weak reference: after
        # destroy_process_group + init_process_group a new group can reuse the id() of the old one with the same world / rank /
        # device -- its predecessor's context (IPC handles into dead mailboxes, a sequence counter the restarted peers do not
        # share) must not be inherited.  The default group (None) is validated against dist.group.WORLD.
        import weakref
        g = group if group is not None else (dist.group.WORLD if dist.is_available() and dist.is_initialized() else None)
        key = (world_size(group), rank(group), str(torch.device(device)), id(g))
        hit = cls._by_group.get(key)
        if hit is not None:
            ref, mb = hit
            if (ref is None and g is None) or (ref is not None and ref() is g):
                return mb
            if mb:                                  # same id, different object: the old group is gone
                mb.lib.skd_sync_destroy(mb.ctx)
        try:
            ref = weakref.ref(g) if g is not None else None
        except TypeError:                           # not weak-referenceable: fall back to a strong reference (keeps the id unique)
            ref = (lambda obj: (lambda: obj))(g)
        mb = cls._create(group, device)
        cls._by_group[key] = (ref, mb)
        return mb

    @classmethod
    def active(cls):
        return any(mb for _, mb in cls._by_group.values())

    @classmethod
    def reset(cls):
        """COLLECTIVE in effect: every rank must drop its mailboxes at the same point of the program (the next ``get`` sets them
        up again, from sequence number zero, under the SKD_SYNC_IPC / SKD_ABN_SYNC_FUSED settings of that moment)."""
        for _, mb in cls._by_group.values():
            if mb:
                mb.lib.skd_sync_destroy(mb.ctx)
        cls._by_group.clear()

    @classmethod
    def set_timeout_all(cls, seconds):
        """The in-kernel wait limit of every live mailbox context (SKD_SYNC_TIMEOUT_S applies to contexts created later)."""
        for _, mb in cls._by_group.values():
            if mb:
                mb.lib.skd_sync_set_timeout(mb.ctx, float(seconds))

    @classmethod
    def _create(cls, group, device):
        import ctypes
        import socket
        from .. import _lib
        import contextlib
        w = world_size(group)
        on_gpu = torch.device(device).type == "cuda"
        # (CPU tensors only with the tests' C-ABI double installed: its mailboxes are POSIX shared memory, oracle/sync_ref.c)
        # (a group of ONE rank under solo_rehearsal gets a mailbox too: it has no peer to open, and its exchanges run the same kernels)
        if (not replicated(group) or w > 16 or os.environ.get("SKD_SYNC_IPC", "1") != "1"
                or not (on_gpu or _lib.test_backend_active())):
            return None
        lib = _lib.get()
        rk = dist.get_rank(group)
        nb = lib.skd_sync_handle_bytes()
        buf = ctypes.create_string_buffer(nb)
        on_device = (lambda: torch.cuda.device(device)) if on_gpu else contextlib.nullcontext
        with on_device():
            ctx = lib.skd_sync_create(w, rk, ctypes.cast(buf, ctypes.c_void_p))
        dev_id = None
        if on_gpu:
            dev_id = device_identity(torch.cuda.get_device_properties(torch.device(device)), torch.device(device).index)
        mine = (bytes(buf.raw) if ctx else None, socket.gethostname(), dev_id)
        everyone = [None] * w
        dist.all_gather_object(everyone, mine, group=group)
        good = bool(ctx) and all(h is not None and host == mine[1] for h, host, _ in everyone)
        if on_gpu:
            # ranks of this group that sit on MY device: their grid-barrier launches (which, synchronised, wait for each other
            # inside the kernel) must fit the device together
            here = sum(1 for _, host, d in everyone if host == mine[1] and d == dev_id)
            share_device(here)
            if here <= 1 and dist.get_backend(group) == "nccl" and sync_fused_over_rccl():
                reserve_for_collectives(device)         # (also for groups that were not set up by init_distributed)
        if good:
            blob = ctypes.create_string_buffer(b"".join(h for h, _, _ in everyone), nb * w)
            with on_device():
                good = bool(lib.skd_sync_connect(ctx, ctypes.cast(blob, ctypes.c_void_p)))
        def agree(ok):                               # every rank or nobody: all-reduce(MIN) of the local verdicts
            v = torch.tensor([1.0 if ok else 0.0], device=device if dist.get_backend(group) == "nccl" else "cpu")   # gloo: host
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=group)
            return float(v) >= 1.0

        good = agree(good)                           # a rank that could not connect must not leave the others spinning in the self-test
        if good:                                     # self-test: four all-gathers (both slot parities, growing payloads) of a known pattern
            for n in (2, 64, 1024, lib.skd_sync_max_floats()):
                src = (torch.arange(n, dtype=torch.float32) * 0.5 + rk * 4096.0).to(device)
                out = torch.full((w, n), -1.0, device=device)
                good = good and bool(lib.skd_sync_all_gather(ctx, n, src.data_ptr(), out.data_ptr(), _lib.stream_of(src)))
                want = torch.stack([torch.arange(n, dtype=torch.float32) * 0.5 + r * 4096.0 for r in range(w)])
                good = good and torch.equal(out.cpu(), want)
            good = agree(good)
        if not good:
            if ctx:
                lib.skd_sync_destroy(ctx)
            return None
        # The group works: from here on an exchange waits for a late peer as long as torch.distributed would have (a loader
        # stall, a MIOpen JIT on one rank, a slow checkpoint filesystem -- ADVICE r03); a wait that still runs out poisons the
        # statistics AND raises the device status word that NetModel checks every step (_lib.raise_on_device_errors).
        lib.skd_sync_set_timeout(ctx, float(os.environ.get("SKD_SYNC_TIMEOUT_S", "600")))
        return cls(ctx, lib, w, rk)


def broadcast_module(module, src=0, group=None):
    """Make every replica bit-identical to rank ``src`` once, at construction."""
    if not replicated(group):
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=group)


class CommTimer:
    """Optional accounting of the time the COMPUTE stream spends blocked on collectives (bench.py --gpus N: the ``comm``
    object of the JSON line).  A span is bracketed by two events on the current stream -- one recorded before the
    collective (or the wait for an asynchronous one) is issued, one after -- so its duration is what the collective
    exposed to the critical path, not its own length.  Off unless ``enable()`` was called; CPU tensors (gloo tests) use
    the host clock."""

    def __init__(self):
        self.on = False
        self.spans = []

    def enable(self):
        self.on, self.spans = True, []

    def begin(self, tag, ref):
        if not self.on:
            return None
        if ref.is_cuda:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(ref.device))
            return (tag, e0, ref.device)
        import time
        return (tag, time.perf_counter(), None)

    def end(self, tok):
        if tok is None:
            return
        tag, t0, dev = tok
        if dev is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(dev))
            self.spans.append((tag, t0, e1))
        else:
            import time
            self.spans.append((tag, None, time.perf_counter() - t0))

    def disable(self):
        """-> {tag: (total milliseconds, spans)}; synchronises the device once."""
        self.on = False
        if any(t0 is not None for _, t0, _ in self.spans) and torch.cuda.is_available():
            torch.cuda.synchronize()
        out = {}
        for tag, t0, t1 in self.spans:
            ms = t0.elapsed_time(t1) if t0 is not None else 1e3 * t1
            tot, n = out.get(tag, (0.0, 0))
            out[tag] = (tot + ms, n + 1)
        self.spans = []
        return out


comm_timer = CommTimer()


class _Bucket:
    __slots__ = ("params", "flat", "pending", "work", "offsets", "views", "avg")


class GradientAllReducer:
    """Bucketed, backward-overlapped gradient averaging (replaces ReduceAddCoalesced).

    Parameters are bucketed in reverse registration order (roughly the order backward produces their
    gradients).  When the last gradient of a bucket has been accumulated, the bucket is packed into
    its persistent flat buffer and an asynchronous all-reduce is issued; ``finish()`` (called before
    the optimizer step) waits for all of them and unpacks the averages.  Bucket size: one xGMI ring
    step moves bucket/world bytes per link, so >= 16 MiB keeps every link in its bandwidth regime
    (7 links x ~153 GB/s per GPU) while still giving backward several buckets to overlap with.
    """

    def __init__(self, params, group=None, bucket_bytes=16 << 20):
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self.world = world_size(group)
        self.active = replicated(group)        # world > 1 (or the one-rank rehearsal: same hooks, same buckets, same waits)
        self.armed = False
        self.buckets = []
        self._bucket_of = {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self._handles = []
        if self.active:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _close(self, plist):
        b = _Bucket()
        b.params = list(plist)
        n = sum(p.numel() for p in plist)
        b.flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        b.offsets = []
        b.views = []
        o = 0
        for p in plist:
            b.offsets.append(o)
            # The segment is laid out in the PARAMETER's memory order (a channels-last weight stays channels-last): the gradient
            # autograd hands over has those strides, so the pack is ONE multi-tensor copy per bucket (torch._foreach_copy_'s fast
            # route needs equal strides), the averaged segment can BE p.grad afterwards (no unpack), and the fused SGD update sees
            # a gradient laid out like its parameter.  Every replica builds the same model, hence the same element order.
            if p.dim() > 1 and _dense_like(p):
                b.views.append(b.flat[o:o + p.numel()].as_strided(p.shape, p.stride()))
            else:
                b.views.append(b.flat[o:o + p.numel()].view(p.shape))
            o += p.numel()
        b.pending = len(plist)
        b.work = None
        for p in plist:
            self._bucket_of[p] = b
        self.buckets.append(b)

    def arm(self):
        """Call right before the backward whose gradients must be averaged."""
        self.armed = True
        for b in self.buckets:
            b.pending = len(b.params)
            b.work = None

    def _on_grad(self, p):
        if not self.armed:
            return
        b = self._bucket_of[p]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        with torch.no_grad():
            dsts, srcs = [], []
            for p, v in zip(b.params, b.views):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():      # (a gradient that already IS its segment: zero_grad(set_to_none=False))
                    dsts.append(v)
                    srcs.append(p.grad)
            if dsts:
                torch._foreach_copy_(dsts, srcs)         # one multi-tensor launch per bucket where the strides agree
        # RCCL averages in the collective (ncclAvg); gloo has no AVG: sum, then one scaling pass per bucket in finish()
        b.avg = _backend_has_avg(self.group) and b.flat.is_cuda
        if b.avg:
            try:
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                return
            except (RuntimeError, ValueError, NotImplementedError):      # a build whose RCCL has no ncclAvg: refused at the call, synchronously
                _AVG["broken"] = True
                b.avg = False
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Wait for the in-flight buckets, launch any bucket whose parameters did not all receive a
        gradient (same set on every rank), and make the averages the parameters' ``.grad``: each ``p.grad`` becomes the
        parameter's segment of the bucket (a view with the parameter's own strides -- no copy back; the next
        ``zero_grad()`` drops it, the next backward produces a fresh gradient that is packed again)."""
        if not self.active or not self.armed:
            self.armed = False
            return
        for b in self.buckets:
            if b.work is None:
                self._launch(b)
        inv = 1.0 / self.world
        with torch.no_grad():
            for b in self.buckets:
                tok = comm_timer.begin("allreduce_wait", b.flat)
                b.work.wait()
                comm_timer.end(tok)
                if not b.avg and self.world > 1:
                    b.flat.mul_(inv)
                for p, v in zip(b.params, b.views):
                    p.grad = v
                b.work = None
        self.armed = False


def _dense_like(p):
    """True when ``p`` is non-overlapping and dense (some permutation of a contiguous tensor): its strides can be copied onto a flat
    segment of numel() elements."""
    if p.is_contiguous():
        return True
    sizes_strides = sorted(((st, sz) for sz, st in zip(p.shape, p.stride()) if sz > 1))
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True


_AVG = {"broken": False}


def _backend_has_avg(group=None):
    if _AVG["broken"]:
        return False
    try:
        return dist.get_backend(group) == "nccl"
    except Exception:
        return False


class DataParallelModel(nn.Module):
    """Signature shim of utils/parallel.py:66-111: ``model(inputs, parallel=...)`` returns the
    module's own (un-gathered) output for this rank's shard."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids is not None else [0]
        self.dim = dim

    def forward(self, inputs, **kwargs):
        kwargs.pop("parallel", None)          # "this key is unexpected" (parallel.py:104)
        return self.module(inputs, **kwargs)


class my_DataParallelCriterion(nn.Module):
    """Signature shim of utils/parallel.py:114-155: ``criterion(inputs, *targets, is_target_scattered=...)``."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids is not None else [0]

    def forward(self, inputs, *targets, **kwargs):
        kwargs.pop("is_target_scattered", None)
        return self.module(inputs, *targets, **kwargs)


DataParallelCriterion = my_DataParallelCriterion
