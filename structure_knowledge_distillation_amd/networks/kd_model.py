"""NetModel -- the distillation-step orchestrator, same surface as the reference's
networks/kd_model.py:27-193 so ``train_and_eval.py:19-29`` runs unchanged:

    model = NetModel(args)
    model.adjust_learning_rate(args.lr_g, model.G_solver, step)
    model.set_input((images, labels, _, _)); model.optimize_parameters(); model.print_info(epoch, step)

One step = frozen teacher forward (no grad, eval) -> student forward (train) -> G loss =
CE(DSN) + lambda_pi*Pi + lambda_pa*Pa + lambda_d*Ho -> backward -> SGD -> (Ho) discriminator step
(adv + WGAN-GP) -> SGD  (kd_model.py:167-173).

What differs from the reference, none of it changing a number the step produces:
  * one process per GPU (utils/parallel.py here); gradients averaged by bucketed RCCL all-reduce;
  * the four logged scalars are kept as device tensors and read back lazily (one host sync when
    ``print_info`` / the attributes are read, instead of four ``.item()`` stalls inside the step);
  * the teacher's cross-entropy, which the reference computes and discards (kd_model.py:129), is not computed
    (``model.log_teacher_ce = True`` computes it, forward only, and keeps it as ``mc_T_loss``);
  * while the student loss is back-propagated through D (kd_model.py:148-150) D's parameters do not
    require grad: the reference computes those weight gradients and then zeroes them (kd_model.py:154).
"""
import logging
import os
import os.path as osp

import torch
import torch.nn as nn  # noqa: F401
import torch.optim as optim

from .. import _lib
from ..utils import parallel as parallel_old
from ..utils.criterion import (CriterionAdditionalGP, CriterionAdv, CriterionAdvForG, CriterionDSN,
                               CriterionPairWiseforWholeFeatAfterPool, CriterionPixelWise)
from ..utils.utils import print_model_parm_nums, to_tuple_str  # noqa: F401
from .pspnet_combine import BasicBlock, Bottleneck, Res_pspnet
from .sagan_models import Discriminator


def load_T_model(model, ckpt_path):
    """utils/utils.py:73-91: teacher checkpoints store the PSP module under head.0 / the classifier under head.1."""
    if not ckpt_path or not osp.isfile(ckpt_path):
        logging.info("teacher checkpoint %r not found: random initialisation", ckpt_path)
        return False
    saved = torch.load(ckpt_path, map_location="cpu")
    new = {}
    for k, v in saved.items():
        if k.startswith("head.0."):
            k = "pspmodule." + k[len("head.0."):]
        elif k.startswith("head.1."):
            k = "head." + k[len("head.1."):]
        new[k] = v
    model.load_state_dict(new, strict=False)
    return True


def _strip_module(state_dict, with_module):
    """Checkpoints written from an nn.DataParallel wrapper carry a 'module.' prefix (utils/utils.py:119-122, 141-145)."""
    if with_module:
        return state_dict
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def load_S_model(args, model, with_module=False):
    """utils/utils.py:93-128.  Either the ImageNet ResNet18 weights by key intersection (``is_student_load_imgnet``), or --
    ``S_resume`` -- the run's own ``<S_ckpt_path>/model_best.pth.tar``: a dict with 'state_dict' (+ optional 'step',
    'epoch', 'best_mean_IU', 'IU_array') from which ``args.last_step / start_epoch / best_mean_IU`` are restored so
    that train_and_eval.py:20-21 continues where it stopped.  Returns what was loaded: 'imagenet', 'resume' or False."""
    ckpt_dir = getattr(args, "S_ckpt_path", None)
    if ckpt_dir and not osp.exists(ckpt_dir) and not str(ckpt_dir).endswith((".pth", ".tar")):
        os.makedirs(ckpt_dir, exist_ok=True)                                      # utils.py:95-96
    if getattr(args, "is_student_load_imgnet", False):
        path = getattr(args, "student_pretrain_model_imgnet", None)
        if path and osp.isfile(str(path)):
            saved = torch.load(path, map_location="cpu")
            own = model.state_dict()
            own.update({k: v for k, v in saved.items() if k in own and own[k].shape == v.shape})
            model.load_state_dict(own)
            logging.info("=> load %s", path)
            return "imagenet"
        logging.info("=> the pretrain model on imgnet %r does not exist", path)
        return False
    if getattr(args, "S_resume", False) and ckpt_dir:
        file = osp.join(str(ckpt_dir), "model_best.pth.tar")
        if osp.isfile(file):
            ckpt = torch.load(file, map_location="cpu")
            args.last_step = ckpt.get("step")
            args.start_epoch = ckpt.get("epoch")
            args.best_mean_IU = ckpt.get("best_mean_IU")
            model.load_state_dict(_strip_module(ckpt["state_dict"], with_module))
            logging.info("=> loaded checkpoint %r (epoch:%s step:%s best_mean_IU:%s IU_array:%s)", file, args.start_epoch,
                         args.last_step, args.best_mean_IU, ckpt.get("IU_array"))
            return "resume"
        logging.info("=> checkpoint %r does not exist: random initialisation", file)
    return False


def load_D_model(args, model, with_module=False):
    """utils/utils.py:130-151: ``D_resume`` restores the discriminator (and start_epoch / best_mean_IU) from
    ``<D_ckpt_path>/model_best.pth.tar``."""
    ckpt_dir = getattr(args, "D_ckpt_path", None)
    if not getattr(args, "D_resume", False) or not ckpt_dir:
        return False
    os.makedirs(ckpt_dir, exist_ok=True)
    file = osp.join(str(ckpt_dir), "model_best.pth.tar")
    if not osp.isfile(file):
        logging.info("=> checkpoint %r does not exist: random initialisation", file)
        return False
    ckpt = torch.load(file, map_location="cpu")
    args.start_epoch = ckpt["epoch"]
    args.best_mean_IU = ckpt["best_mean_IU"]
    model.load_state_dict(_strip_module(ckpt["state_dict"], with_module))
    logging.info("=> loaded checkpoint %r (epoch %s)", file, ckpt["epoch"])
    return True


# With the teacher on its own stream: its deep-supervision branch (a 2.4 ms convolution whose output only the teacher's own CE would
# read -- computed like the reference's forward does, kd_model.py:121, consumed by nothing) is issued LAST, behind the event the step's
# criteria wait for, and works off beside the criteria and the start of the backbone backward: -0.46 ms per step, three of three
# interleaved pairs (profiles/r09y_teacher_dsn_last_ab.txt).  False: in the reference's order (between layer3 and layer4).
TEACHER_DSN_LAST = True


class NetModel():
    def name(self):
        return "kd_seg"

    def DataParallelModelProcess(self, model, ParallelModelType=1, is_eval="train", device="cuda"):
        if ParallelModelType not in (1, 2):
            raise ValueError("ParallelModelType should be 1 or 2")
        parallel_model = parallel_old.DataParallelModel(model)
        if is_eval == "eval":
            parallel_model.eval()
        elif is_eval == "train":
            parallel_model.train()
        else:
            raise ValueError("is_eval should be eval or train")
        parallel_model.float()
        parallel_model.to(device)
        return parallel_model

    def DataParallelCriterionProcess(self, criterion, device="cuda"):
        criterion = parallel_old.my_DataParallelCriterion(criterion)
        criterion.to(self.args.device)
        return criterion

    def __init__(self, args):
        self.args = args
        device = args.device
        from .. import configure_miopen
        configure_miopen()                           # tuned find-db + kernel cache, Winograd off (process-wide, explicit)
        torch.backends.cudnn.enabled = True          # MIOpen
        # SKD_DETERMINISTIC=1: run-to-run bit-reproducible steps.  Every hand-written kernel already is (fixed-order
        # reductions, no float atomics); what is not by default are MIOpen's fastest fp32 weight-gradient / backward-data
        # solvers (split-K with atomic adds).  MIOpen's own DETERMINISTIC attribute (torch.backends.cudnn.deterministic)
        # was measured first (gpurun_out r03a, tools/determinism_probe.py): with this find-db it falls back to solvers
        # that take 9.2 s per step AND still differ between runs, so the mode routes every convolution through PyTorch's
        # own im2col + rocBLAS path instead (cudnn disabled) with rocBLAS in atomics-not-allowed mode
        # (torch.use_deterministic_algorithms; warn_only because a few stock backward ops have no deterministic variant
        # and are not on this path).  Cost: DESIGN.md section 9 (profiles/r03*_determinism.json).
        self.deterministic = os.environ.get("SKD_DETERMINISTIC", "0") == "1"
        if self.deterministic:
            torch.backends.cudnn.enabled = False
            torch.use_deterministic_algorithms(True, warn_only=True)
        student = Res_pspnet(BasicBlock, [2, 2, 2, 2], num_classes=args.classes_num)
        load_S_model(args, student, False)
        print_model_parm_nums(student, "student_model")
        self.parallel_student = self.DataParallelModelProcess(student, 2, "train", device)
        self.student = student
        parallel_old.broadcast_module(student)      # every replica starts from rank 0's weights
        # Channels-last student: every convolution NHWC-native in MIOpen, InPlace-ABN through the skd_abn_*_nhwc training kernels
        # (the shipped find-db is tuned for the NHWC problems, tools/miopen_tune.py; A/B of round 1: -1.9 ms per step).
        self.student_nhwc = torch.device(device).type == "cuda"
        if self.student_nhwc:
            os.environ["PYTORCH_MIOPEN_SUGGEST_NHWC"] = "1"
            student.to(memory_format=torch.channels_last)

        teacher = Res_pspnet(Bottleneck, [3, 4, 23, 3], num_classes=args.classes_num)
        load_T_model(teacher, getattr(args, "T_ckpt_path", None))
        print_model_parm_nums(teacher, "teacher_model")
        for p in teacher.parameters():
            p.requires_grad_(False)
        self.parallel_teacher = self.DataParallelModelProcess(teacher, 2, "eval", device)
        self.teacher = teacher
        parallel_old.broadcast_module(teacher)
        # The frozen teacher runs channels-last: MIOpen's fastest fp32 kernels on gfx950 are NHWC igemm kernels, and
        # with NCHW tensors MIOpen wraps each of them in NCHW<->NHWC transposes (5.6 ms per step, profiles/).  Its
        # eval-mode BN (+ReLU, +residual) has an NHWC kernel (skd_abn_apply_nhwc).
        self.teacher_nhwc = torch.device(device).type == "cuda"
        if self.teacher_nhwc:
            os.environ["PYTORCH_MIOPEN_SUGGEST_NHWC"] = "1"
            teacher.to(memory_format=torch.channels_last)

        D_model = Discriminator(args.preprocess_GAN_mode, args.classes_num, args.batch_size,
                                args.imsize_for_adv, args.adv_conv_dim)
        load_D_model(args, D_model, False)
        print_model_parm_nums(D_model, "D_model")
        self.parallel_D = self.DataParallelModelProcess(D_model, 2, "train", device)
        self.D_model = D_model

        parallel_old.broadcast_module(D_model)      # (the reference re-broadcasts all three every forward)

        self._s_params = [p for p in self.student.parameters() if p.requires_grad]
        self._d_params = [p for p in D_model.parameters() if p.requires_grad]
        # kd_model.py:74-75.  On the GPU the update runs as torch's FUSED multi-tensor SGD (same formula: g += wd * p;
        # buf = mu * buf + g; p -= lr * buf -- one or two launches per optimizer instead of seven; profiles/r05f_timeline.md had the
        # student's update at 0.17 ms of main-stream time per step)
        fused = {"fused": True} if torch.device(device).type == "cuda" else {}
        self.G_solver = optim.SGD([{"params": self._s_params, "initial_lr": args.lr_g}], args.lr_g,
                                  momentum=args.momentum, weight_decay=args.weight_decay, **fused)
        self.D_solver = optim.SGD([{"params": self._d_params, "initial_lr": args.lr_d}], args.lr_d,
                                  momentum=args.momentum, weight_decay=args.weight_decay, **fused)
        self._s_reducer = parallel_old.GradientAllReducer(self._s_params)
        self._d_reducer = parallel_old.GradientAllReducer(self._d_params)

        self.best_mean_IU = args.best_mean_IU

        self.criterion = self.DataParallelCriterionProcess(CriterionDSN())
        self.criterion_pixel_wise = self.DataParallelCriterionProcess(CriterionPixelWise())
        self.criterion_pair_wise_for_interfeat = self.DataParallelCriterionProcess(
            CriterionPairWiseforWholeFeatAfterPool(scale=args.pool_scale, feat_ind=-5))
        self.criterion_adv = self.DataParallelCriterionProcess(CriterionAdv(args.adv_loss_type))
        if args.adv_loss_type == "wgan-gp":
            self.criterion_AdditionalGP = self.DataParallelCriterionProcess(
                CriterionAdditionalGP(self.parallel_D, args.lambda_gp))
        self.criterion_adv_for_G = self.DataParallelCriterionProcess(CriterionAdvForG(args.adv_loss_type))

        # N > 1: off unless forced.  Measured (profiles/r04h_two_ranks_one_gpu_bisect.txt): with two ranks SHARING one MI355X the
        # replayed graph turns a 94 ms step into 7.8 s -- graph launches and the peer process's in-kernel exchange waits do not
        # get co-scheduled.  One rank per GPU should not meet that, but no multi-GPU box was available to show it, the gain is
        # 1.2 %, and the failure mode is two orders of magnitude: the default for N > 1 stays the eager teacher.
        graph_env = os.environ.get("SKD_TEACHER_GRAPH", "1")
        # SKD_TEACHER_STREAM (round 6; default ON at N = 1): the frozen teacher's forward is issued EAGERLY on a second HIP stream
        # beside the student's forward (kd_model.py:121-123 runs them back to back) and joined before the criteria.  Two independent
        # kernel sequences fill each other's launch tails and partial last rounds.  One box, interleaved
        # (profiles/r09k_teacher_stream_one_box_cool_long_warm.txt): 61.77 (teacher = one hipGraph replay on the main stream) ->
        # 61.01 ms per step in 20-step runs on the fresh box, 61.87 -> 61.10 over 500 steps, 61.85 -> 61.01 in 20-step runs right
        # after those; on a slower box of the pool (63.6-63.9 ms in every run) it was neutral (profiles/r09h_teacher_stream_ab.txt).
        # NOT combined with the graph: a graph replayed beside another stream's kernels does not co-schedule on this runtime (89 ms
        # per step; the same pathology as two ranks sharing a device and as graph branches, ROUND6_NOTES.md) -- SKD_TEACHER_GRAPH=
        # force selects the graph and switches the stream off; SKD_TEACHER_STREAM=0 gives the round-4/5 default (graph, one stream).
        # Never in the deterministic mode: its three-stream configuration of im2col + rocBLAS kernels with atomics off
        # intermittently never finished inside the vendor stack in round 4 -- the reason the rounds 2-4
        # switch of this name was removed; the default mode never hung then and has not in this round's 4000 steps and two full
        # test-suite runs.  N > 1: opt-in only (SKD_TEACHER_STREAM=1), like every other form no multi-GPU box has run.
        stream_env = os.environ.get("SKD_TEACHER_STREAM", "0" if parallel_old.replicated() else "1")
        self._teacher_stream = (torch.cuda.Stream(device=device)
                                if (stream_env == "1" and graph_env != "force"
                                    and torch.device(device).type == "cuda" and not self.deterministic) else None)
        self._teacher_graph_on = (graph_env in ("1", "force") and torch.device(device).type == "cuda"
                                  and self._teacher_stream is None
                                  and (not parallel_old.replicated() or graph_env == "force")
                                  and not self.deterministic_no_graph())
        self._teacher_graphs = {}
        self._teacher_tensors = list(teacher.parameters()) + list(teacher.buffers())
        self._teacher_stamp = None
        # The D step (kd_model.py:153-165) only needs the two logit tensors and D's own state: it runs on a second HIP
        # stream next to the student's backbone backward (its ~700 launches are 3-8 us kernels on (8, 19..512, <=65, 65)
        # tensors that leave the chip idle when serialised).  SKD_D_STREAM=0 restores the serial order.
        self._d_stream = (torch.cuda.Stream(device=device, priority=-1)
                          if (os.environ.get("SKD_D_STREAM", "1") == "1" and torch.device(device).type == "cuda") else None)
        # SKD_D_GRAPH=1 (round 6, opt-in): the whole D step but its SGD update -- D(T), D(S), the WGAN-GP double backward,
        # d_loss.backward() -- captured ONCE per logit shape into a hipGraph and replayed: ~1100 host launches per step become one.
        # At N = 1 the host is ahead of the device either way (the D stream is hidden): nothing to win there.
        # N > 1: the capture holds NO collective -- the reducer's hooks are not armed inside it; the critic's gradients (12.8 MB) are
        # packed and all-reduced in one go after the replay, on the D stream, before the update (the eager form overlaps them with
        # D's backward; both sit beside the student's backbone backward).  Not with ranks SHARING a device (the two-process tests:
        # graph launches and a co-tenant's in-kernel waits do not co-schedule, profiles/r04h_two_ranks_one_gpu_bisect.txt).
        self._d_graph_on = os.environ.get("SKD_D_GRAPH", "0") == "1" and torch.device(device).type == "cuda"
        self._d_graphs = {}
        self._d_eager_steps = 0
        self._scalars = {"mc_G_loss": 0.0, "pi_G_loss": 0.0, "pa_G_loss": 0.0, "G_loss": 0.0, "D_loss": 0.0, "mc_T_loss": 0.0}
        self.gp_alpha = None     # tests pin the WGAN-GP interpolation coefficients through this
        self.log_teacher_ce = False      # True: also compute the teacher's own CE (kd_model.py:129: computed and discarded) as mc_T_loss
        # preds_T[1] (the teacher's deep-supervision logits) is read by nothing but that CE; it is computed anyway, like the
        # reference's forward (``model.teacher.skip_dsn = True`` skips it: bench.py --dsn-ab's informative figure, never the default)
        self.teacher.skip_dsn = False

        # MIOpen find mode is opt-in: this ROCm image ships no gfx950 find/kernel database, so "find"
        # JIT-compiles every candidate solver for every convolution shape on a fresh machine.
        torch.backends.cudnn.benchmark = os.environ.get("SKD_MIOPEN_FIND", "0") == "1"
        if not torch.backends.cudnn.benchmark:
            from .. import check_miopen_db
            check_miopen_db()        # warns when the shipped find-db does not belong to the running MIOpen build
        snap = getattr(args, "snapshot_dir", None)
        if snap and not os.path.exists(snap):
            os.makedirs(snap)

    def deterministic_no_graph(self):
        """PyTorch's im2col convolutions (the deterministic mode's path) allocate and free column buffers per call; they capture
        fine, but the mode exists to compare orders of execution bit for bit, so it keeps the eager teacher unless asked
        (SKD_TEACHER_GRAPH=force)."""
        return self.deterministic and os.environ.get("SKD_TEACHER_GRAPH", "1") != "force"

    # ---- logged scalars: device tensors until somebody reads them -------------------------------
    def _get_scalar(self, key):
        v = self._scalars[key]
        if torch.is_tensor(v):
            pend = getattr(self, "_scalar_pending", None)
            if pend is not None and pend[0].get(key) is v:
                # all logged scalars of the step came back in ONE device-to-host copy (_publish_scalars): wait for it once
                tensors, host, event = pend
                event.synchronize()
                vals = host.tolist()
                for i, (k, t) in enumerate(tensors.items()):
                    if self._scalars.get(k) is t:
                        self._scalars[k] = vals[i]
                self._scalar_pending = None
                v = self._scalars[key]
            else:
                v = v.item()
                self._scalars[key] = v
            _lib.raise_on_device_errors()      # the read above synchronised: a timed-out in-kernel wait of this step is visible
        return v

    def _publish_scalars(self, stream=None):
        """End of a step: the logged scalars (device tensors) are packed by one small kernel and sent to a pinned host buffer with
        ONE asynchronous copy; ``print_info`` / the attribute reads then cost one wait instead of one blocking ``.item()`` each.
        Measured (profiles/r05a_timeline.md): the five reads of train_and_eval.py:26 were five serial D2H copies ~27 us apart with the
        GPU idle at every step boundary (the reference stalls four times INSIDE the step, kd_model.py:127-165).
        ``stream``: where the LAST scalar of the step is produced (the D stream when the D step runs on its own): packing there keeps
        the cross-stream hand-off (150-280 us under the profiler, profiles/r05f_timeline.md) out of the read-back's path."""
        tensors = {k: v for k, v in self._scalars.items() if torch.is_tensor(v)}
        if not tensors or not all(t.is_cuda and t.dtype == torch.float32 for t in tensors.values()):
            self._scalar_pending = None
            return
        dev = next(iter(tensors.values())).device
        stream = torch.cuda.current_stream(dev) if stream is None else stream
        with torch.cuda.stream(stream):
            packed = torch.stack([t.reshape(()) for t in tensors.values()])
            host = getattr(self, "_scalar_host", None)
            if host is None or host.numel() != len(tensors):
                host = self._scalar_host = torch.empty(len(tensors), dtype=torch.float32, pin_memory=True)
            host.copy_(packed, non_blocking=True)
            event = torch.cuda.Event()
            event.record(stream)
        self._scalar_pending = (tensors, host, event)

    mc_G_loss = property(lambda self: self._get_scalar("mc_G_loss"))
    pi_G_loss = property(lambda self: self._get_scalar("pi_G_loss"))
    pa_G_loss = property(lambda self: self._get_scalar("pa_G_loss"))
    G_loss = property(lambda self: self._get_scalar("G_loss"))
    D_loss = property(lambda self: self._get_scalar("D_loss"))
    mc_T_loss = property(lambda self: self._get_scalar("mc_T_loss"))   # the teacher's own CE (reference: computed, discarded)

    def set_input(self, data):
        images, labels, _, _ = data
        dev = self.args.device
        if hasattr(images, "to_device") and hasattr(images, "params"):
            # dataset.CSDataSet.collate: decoded uint8 batch + random draws from a loader worker; the reference's whole
            # per-sample transform (dataset/datasets.py:176-210) runs here, in the process that owns the GPU, as one kernel
            images, labels = images.to_device(dev)
        self.images = images.to(dev, non_blocking=True)
        self.labels = labels.long().to(dev, non_blocking=True)
        parallel_old.set_replica_batch(self.images.shape[0], self.images.device)   # InPlaceABNSync pools by sample count

    def lr_poly(self, base_lr, iter, max_iter, power):
        return base_lr * ((1 - float(iter) / max_iter) ** (power))

    def adjust_learning_rate(self, base_lr, optimizer, i_iter):
        args = self.args
        lr = self.lr_poly(base_lr, i_iter, args.num_steps, args.power)
        optimizer.param_groups[0]["lr"] = lr
        return lr

    def _teacher_forward_eager(self, images):
        args = self.args
        with torch.no_grad():
            images_T = images.contiguous(memory_format=torch.channels_last) if self.teacher_nhwc else images
            preds_T = self.parallel_teacher.eval()(images_T, parallel=args.parallel)
            # the two logit tensors the criteria / D read are handed on in the reference's NCHW layout (19 channels: no channel
            # quads); the PSP feature (preds[2], 69 MB at batch 8) stays as it is -- the pair-wise pooling reads channels-last
            return [None if t is None else t.contiguous() for t in preds_T[:2]] + list(preds_T[2:])

    def _teacher_forward(self):
        """kd_model.py:121-122.  The frozen teacher is 100 % static -- same weights, same shapes, no autograd, ~330 launches
        issued op by op from Python -- so (SKD_TEACHER_GRAPH, default on) its forward is captured ONCE per input shape into a
        hipGraph and replayed: one host call per step instead of ~330 launches plus their Python dispatch (VERDICT r03 item 4;
        DESIGN.md Appendix A.3 has the A/B).  The graph owns its input / activation / output buffers (2.6 GB at batch 8,
        resident in HBM between steps -- 288 GB per GPU is what makes that free); ``preds_T`` are the graph's output tensors:
        STATIC buffers that the next replay overwrites.  Inside a step that is safe by stream order (main.wait_stream(D stream)
        at the end of a step keeps the next replay behind every reader); a consumer that wants to keep teacher outputs ACROSS
        steps must take ``model.teacher_outputs()`` (clones) instead of holding ``model.preds_T``.  The replayed kernels are the SAME kernels in the same order: results are bit-identical to the
        eager forward under SKD_DETERMINISTIC=1 (tests/test_step_gpu.py)."""
        images = self.images
        if not self._teacher_graph_on or not images.is_cuda:
            return self._teacher_forward_eager(images)
        # the captured kernels read the teacher's tensors -- and the packed / folded copies functional.py derives from them
        # (keyed on the tensors' autograd versions) -- in place: a written tensor (load_state_dict after construction) drops the graphs
        stamp = sum(t._version for t in self._teacher_tensors)
        if stamp != self._teacher_stamp:
            self._teacher_graphs.clear()
            self._teacher_stamp = stamp
        key = (tuple(images.shape), images.dtype, images.device.index)
        entry = self._teacher_graphs.get(key)
        if entry is None:
            try:
                entry = self._capture_teacher(images)
            except Exception as e:           # an op that cannot be captured on this ROCm / MIOpen build: say so, run eagerly
                import warnings
                warnings.warn("teacher hipGraph capture failed (%s: %s): the frozen teacher runs eagerly (SKD_TEACHER_GRAPH=0 "
                              "silences this)" % (type(e).__name__, str(e)[:300]), RuntimeWarning)
                self._teacher_graph_on = False
                torch.cuda.synchronize(images.device)
                return self._teacher_forward_eager(images)
            self._teacher_graphs[key] = entry
        static_in, graph, outs = entry
        static_in.copy_(images)
        graph.replay()
        return list(outs)

    def teacher_outputs(self):
        """The teacher's outputs of the last forward as tensors the caller may keep: with the hipGraph on, ``preds_T`` are the
        graph's static output buffers (overwritten by the next step's replay) and are cloned here; eager outputs are returned as is."""
        if not getattr(self, "_teacher_graph_on", False):
            return list(self.preds_T)
        return [None if t is None else t.clone() for t in self.preds_T]

    def _capture_teacher(self, images):
        static_in = torch.empty_like(images)
        static_in.copy_(images)
        # warm-up on a side stream (lazy MIOpen / rocBLAS handles, kernel-module loads, the library's per-device pools and the
        # packed-parameter caches must exist BEFORE capture: none of that is capturable), then capture
        side = torch.cuda.Stream(device=images.device)
        side.wait_stream(torch.cuda.current_stream(images.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._teacher_forward_eager(static_in)
        torch.cuda.current_stream(images.device).wait_stream(side)
        torch.cuda.synchronize(images.device)
        graph = torch.cuda.CUDAGraph()
        # thread_local: the capture is lazy (first step), i.e. a DataLoader's pin-memory thread (dataset/datasets.py's documented
        # loader: pin_memory=True -> hipHostMalloc / hipEventQuery from another thread) may be alive; in the default "global" mode
        # such a call would invalidate the capture or raise hipErrorStreamCaptureUnsafe IN THAT THREAD, where the try / except
        # around this function cannot see it.  Only this thread issues work into the capture.
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            outs = self._teacher_forward_eager(static_in)
        return static_in, graph, outs

    def _student_forward(self):
        args = self.args
        if not self.student_nhwc:
            return self.parallel_student.train()(self.images, parallel=args.parallel)
        preds = self.parallel_student.train()(self.images.contiguous(memory_format=torch.channels_last),
                                              parallel=args.parallel)
        # logits / DSN logits go to the criteria and to D in the reference's NCHW layout; the PSP feature (17 MB at batch 8) is
        # pooled as it is (functional._pool_feature) and its gradient comes back channels-last: no layout copy either way
        return [t.contiguous() for t in preds[:2]] + list(preds[2:])

    def forward(self):
        # kd_model.py:121-123: teacher forward, student forward -- independent of each other, so (SKD_TEACHER_STREAM, __init__) the
        # teacher runs on its own stream beside the student's forward and is joined before the criteria read its outputs.
        side = self._teacher_stream
        if side is None or not self.images.is_cuda:
            self.preds_T = self._teacher_forward()
            self.preds_S = self._student_forward()
            return
        main = torch.cuda.current_stream(self.images.device)
        side.wait_stream(main)                 # the images (and, across steps, every reader of the previous teacher outputs: the
        outputs_ready = []                     # step ends with main.wait_stream(D stream) and main.wait_stream(teacher stream))

        def _mark(logits, feature):
            # (the criteria read the logits and the PSP feature: both exist here; NetModel hands the logits on as they are when
            # they are already contiguous NCHW -- the head kernel's output -- so nothing of theirs is issued behind this point)
            if logits.is_contiguous():
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(logits.device))
                outputs_ready.append(ev)

        self.teacher.dsn_last = _mark if TEACHER_DSN_LAST else None
        try:
            with torch.cuda.stream(side):
                self.preds_T = self._teacher_forward()
        finally:
            self.teacher.dsn_last = None
        self.preds_S = self._student_forward()
        # The join is LATE: the first read of ``self.preds_T`` (a property) makes the main stream wait for the teacher.  What does not
        # read the teacher -- the student's CE and the critic's forward on the student's logits (student_backward) -- is issued
        # before that and runs while the teacher's last layers still do: -0.4 ms per step (profiles/r09q_late_join_ab.txt).
        self._teacher_pending = (main, side, outputs_ready[0] if outputs_ready else None)

    def _join_teacher(self, whole=True):
        """Make the main stream (and the current one, if different) wait for the teacher.  ``whole=False`` (the step's own criteria,
        which read the logits and the PSP feature only): wait for the event recorded before the deep-supervision branch, which the
        teacher stream works off behind it -- 2.5 ms of convolution that nothing on the critical path reads; ``_finish_teacher`` (end of
        the step) or any read of the ``preds_T`` property waits for the rest."""
        pend = self._teacher_pending
        if pend is not None:
            self._teacher_pending = None
            main, side, ev = pend
            cur = torch.cuda.current_stream(main.device)
            for st in ([main] if cur == main else [main, cur]):     # (first read under another stream: the D step when neither Pi nor Pa is on)
                if ev is not None:
                    st.wait_event(ev)
                else:
                    st.wait_stream(side)
            for t in self._preds_T:            # allocated on the side stream, read on the main and the D stream from here on
                if t is not None:
                    t.record_stream(main)
                    if self._d_stream is not None:
                        t.record_stream(self._d_stream)
            self._teacher_tail = (main, side) if ev is not None else None
        if whole:
            self._finish_teacher()

    def _finish_teacher(self):
        tail = self._teacher_tail
        if tail is not None:
            self._teacher_tail = None
            main, side = tail
            main.wait_stream(side)
            cur = torch.cuda.current_stream(main.device)
            if cur != main:
                cur.wait_stream(side)

    _teacher_tail = None

    def _teacher_main_outputs(self):
        """The teacher's outputs for the step's own criteria and the D step (logits, PSP feature): joined up to the event in front of
        the deep-supervision branch."""
        self._join_teacher(whole=False)
        return self._preds_T

    _teacher_pending = None
    _preds_T = None

    @property
    def preds_T(self):
        """The teacher's outputs of this step.  With the teacher on its own stream (SKD_TEACHER_STREAM) a read joins ALL of it."""
        self._join_teacher(whole=True)
        return self._preds_T

    @preds_T.setter
    def preds_T(self, value):
        self._preds_T = value

    def _adv_for_G(self, logits=None):
        """The critic on the student's logits with its own parameters frozen (kd_model.py:141-143): lambda_d x the adversarial term."""
        args = self.args
        for p in self._d_params:
            p.requires_grad_(False)
        try:
            d_out_S = self.parallel_D(self.preds_S[0] if logits is None else logits, parallel=args.parallel)
        finally:
            for p in self._d_params:
                p.requires_grad_(True)
        return args.lambda_d * self.criterion_adv_for_G(d_out_S, d_out_S, is_target_scattered=True)

    def student_backward(self, on_logits_ready=None):
        """kd_model.py:125-150.  ``on_logits_ready`` (the two-stream step): called once the student loss has finished back-propagating
        through D and the teacher's outputs are joined -- the point from which the D step may run; only the split form below calls
        it (the serial form's caller hooks the logits' gradient instead)."""
        if self._teacher_pending is not None and os.environ.get("SKD_SPLIT_BACKWARD", "1") == "1":
            return self._student_backward_split(on_logits_ready)
        args = self.args
        temp = self.criterion(self.preds_S, self.labels, is_target_scattered=False)
        self._scalars["mc_G_loss"] = temp.detach()
        if self.log_teacher_ce and self.preds_T[1] is not None:     # kd_model.py:129 computes the teacher's CE and throws it away; off unless asked for
            self._scalars["mc_T_loss"] = self.criterion(self.preds_T, self.labels, is_target_scattered=False).detach()
        G_loss = temp
        if args.pi == True:  # noqa: E712  (flags may arrive as 0/1)
            temp = args.lambda_pi * self.criterion_pixel_wise(self.preds_S, self.preds_T, is_target_scattered=True)
            self._scalars["pi_G_loss"] = temp.detach()
            G_loss = G_loss + temp
        if args.pa == True:  # noqa: E712
            temp1 = self.criterion_pair_wise_for_interfeat(self.preds_S, self.preds_T, is_target_scattered=True)
            self._scalars["pa_G_loss"] = temp1.detach()
            G_loss = G_loss + args.lambda_pa * temp1
        if args.ho == True:  # noqa: E712
            G_loss = G_loss + self._adv_for_G()
        self._s_reducer.arm()
        G_loss.backward()
        self._s_reducer.finish()
        self._scalars["G_loss"] = G_loss.detach()

    def _student_backward_split(self, on_logits_ready=None):
        """The same loss and (by linearity of the backward pass) the same gradients with the teacher still running on its own stream:
        the terms that do not read the teacher -- the student's CE and the adversarial term -- are evaluated AND differentiated down
        to the student's outputs first (criteria kernels and the critic's forward + data-gradient backward: launch-bound work that
        runs under the teacher's MFMA-bound last layers); the first read of ``self.preds_T`` then joins the teacher, Pi and Pa are
        evaluated and differentiated down to the student's outputs, and ONE backward pass through the backbone starts from the summed
        output gradients.  G_loss is summed from the same term values in the reference's order (kd_model.py:125-146)."""
        args = self.args
        # what the criteria read of the student: the logits (CE, Pi, the critic), the DSN logits (CE), the PSP feature (Pa).  The criteria
        # see them as LEAVES (detached copies of the same storage): the logits are computed FROM the feature, and a partial backward
        # pass must stop at the student's outputs instead of running on through the heads
        feat = getattr(getattr(self.criterion_pair_wise_for_interfeat, "module", None), "feat_ind", -5) % len(self.preds_S)
        idx = [i for i in sorted({0, 1, feat}) if torch.is_tensor(self.preds_S[i]) and self.preds_S[i].requires_grad]
        S = list(self.preds_S)
        for i in idx:
            S[i] = self.preds_S[i].detach().requires_grad_(True)
        outs = [S[i] for i in idx]
        ce = self.criterion(S, self.labels, is_target_scattered=False)
        self._scalars["mc_G_loss"] = ce.detach()
        early = ce
        adv = None
        if args.ho == True:  # noqa: E712
            adv = self._adv_for_G(S[0])
            early = ce + adv
        grads = list(torch.autograd.grad(early, outs, allow_unused=True))
        if self.log_teacher_ce and self.preds_T[1] is not None:
            self._scalars["mc_T_loss"] = self.criterion(self.preds_T, self.labels, is_target_scattered=False).detach()
        G_loss = ce.detach()
        late = None
        if args.pi == True:  # noqa: E712
            temp = args.lambda_pi * self.criterion_pixel_wise(S, self._teacher_main_outputs(), is_target_scattered=True)   # (joins)
            self._scalars["pi_G_loss"] = temp.detach()
            G_loss = G_loss + temp.detach()
            late = temp
        if args.pa == True:  # noqa: E712
            temp1 = self.criterion_pair_wise_for_interfeat(S, self._teacher_main_outputs(), is_target_scattered=True)
            self._scalars["pa_G_loss"] = temp1.detach()
            G_loss = G_loss + args.lambda_pa * temp1.detach()
            late = args.lambda_pa * temp1 if late is None else late + args.lambda_pa * temp1
        if adv is not None:
            G_loss = G_loss + adv.detach()
        self._join_teacher(whole=False)           # (no Pi and no Pa: nobody has read the teacher yet)
        if on_logits_ready is not None:
            on_logits_ready()
        if late is not None:
            for k, g in enumerate(torch.autograd.grad(late, outs, allow_unused=True)):
                if g is not None:
                    grads[k] = g if grads[k] is None else grads[k] + g
        roots = [(self.preds_S[i], g) for i, g in zip(idx, grads) if g is not None]
        self._s_reducer.arm()
        torch.autograd.backward([t for t, _ in roots], [g for _, g in roots])
        self._s_reducer.finish()
        self._scalars["G_loss"] = G_loss

    def _d_loss(self, logits_S, logits_T, alpha):
        """kd_model.py:153-163 on two DETACHED logit tensors: the critic on teacher and student logits, the adversarial loss and
        the gradient penalty (criterion.py:92-120, 146-166)."""
        args = self.args
        d_out_T = self.parallel_D(logits_T, parallel=True)
        d_out_S = self.parallel_D(logits_S, parallel=True)
        d_loss = args.lambda_d * self.criterion_adv(d_out_S, d_out_T, is_target_scattered=True)
        if args.adv_loss_type == "wgan-gp":
            gp = self.criterion_AdditionalGP([logits_S], [logits_T], alpha=alpha, is_target_scattered=True)
            d_loss = d_loss + args.lambda_d * gp
        return d_loss

    def discriminator_backward(self):
        if self._d_graph_on and parallel_old.ranks_on_this_device() <= 1 and self._discriminator_backward_graphed():
            return
        self.D_solver.zero_grad()
        d_loss = self._d_loss(self.preds_S[0].detach(), self._teacher_main_outputs()[0].detach(), self.gp_alpha)
        self._d_reducer.arm()
        d_loss.backward()
        self._d_reducer.finish()
        self._scalars["D_loss"] = d_loss.detach()
        self.D_solver.step()
        self._d_eager_steps += 1

    def _discriminator_backward_graphed(self):
        """SKD_D_GRAPH=1: replay (after the first two eager steps: lazy handles, kernel modules and the spectral-norm caches must exist
        before a capture) the captured D step; False = run eagerly.  The graph owns the two logit inputs, the interpolation
        coefficients when the caller pins them (``gp_alpha``; otherwise torch.rand is part of the capture and advances the generator
        per replay), every activation and -- because the parameters' ``.grad`` are None when the capture starts -- the gradient
        buffers the optimizer then reads: nobody may ``zero_grad(set_to_none=True)`` the critic between steps (checked: the graph is
        dropped and the step runs eagerly if a gradient went missing)."""
        if self._d_eager_steps < 2:
            return False
        logits_S, logits_T = self.preds_S[0].detach(), self._teacher_main_outputs()[0].detach()
        key = (tuple(logits_S.shape), logits_S.device.index, self.gp_alpha is not None)
        entry = self._d_graphs.get(key)
        if entry is None:
            try:
                entry = self._capture_d_step(logits_S, logits_T)
            except Exception as e:
                import warnings
                warnings.warn("D-step hipGraph capture failed (%s: %s): the D step runs eagerly" % (type(e).__name__, str(e)[:300]),
                              RuntimeWarning)
                self._d_graph_on = False
                torch.cuda.synchronize(logits_S.device)
                return False
            self._d_graphs[key] = entry
        static_S, static_T, static_alpha, graph, d_loss, grads = entry
        if any(p.grad is not g for p, g in zip(self._d_params, grads)):
            self._d_graphs.clear()                  # somebody replaced / dropped the critic's gradient tensors
            self._d_eager_steps = 0
            return False
        static_S.copy_(logits_S)
        static_T.copy_(logits_T)
        if static_alpha is not None:
            static_alpha.copy_(self.gp_alpha)
        graph.replay()
        self._scalars["D_loss"] = d_loss            # static: packed for the read-back right after (same stream), overwritten next step
        if self._d_reducer.active:
            # N > 1: average the graph's gradient buffers over the replicas (one pack + all-reduce per bucket, nothing overlapped: the
            # D stream is off the critical path); finish() leaves p.grad pointing at the averaged bucket segments for the update
            self._d_reducer.arm()
            self._d_reducer.finish()
        self.D_solver.step()
        if self._d_reducer.active:
            for p, g in zip(self._d_params, grads):
                p.grad = g                          # the graph's own buffers again (their identity is checked before every replay)
        return True

    def _capture_d_step(self, logits_S, logits_T):
        dev = logits_S.device
        static_S, static_T = logits_S.clone(), logits_T.clone()
        static_alpha = self.gp_alpha.clone() if self.gp_alpha is not None else None
        self.D_solver.zero_grad(set_to_none=True)   # the capture's backward CREATES the gradient tensors (graph-owned, re-written per replay)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            d_loss = self._d_loss(static_S, static_T, static_alpha)
            d_loss.backward()
            d_loss = d_loss.detach()
        grads = [p.grad for p in self._d_params]
        return static_S, static_T, static_alpha, graph, d_loss, grads

    def optimize_parameters(self):
        if torch.device(self.args.device).type == "cuda":
            _lib.raise_on_device_errors()      # a host load, no synchronisation: errors of the steps already executed
        try:
            self._optimize_parameters()
        finally:
            parallel_old.clear_replica_batch()     # the per-rank sample weights belong to THIS step (set_input)

    def _optimize_parameters(self):
        self.forward()
        self.G_solver.zero_grad()
        ho = self.args.ho == True  # noqa: E712
        side = self._d_stream if ho else None
        if side is None:
            self.student_backward()
            self.G_solver.step()
            if ho:
                self.discriminator_backward()
            if torch.device(self.args.device).type == "cuda":
                self._publish_scalars()
            self._finish_teacher()
            return
        # Same operations, same order per data dependency: the D step may start as soon as the student's logits have
        # received their gradient -- by then the student loss has finished back-propagating through D, so D's
        # spectral-norm state is free to advance (spectral.py:30-31) -- and runs beside the backbone backward and the
        # student's SGD update, neither of which it reads or writes.
        dev = self.preds_S[0].device
        main = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        fired = []

        def _logits_grad_ready(grad):
            ready.record(torch.cuda.current_stream(grad.device))
            fired.append(True)

        if self._teacher_pending is not None and os.environ.get("SKD_SPLIT_BACKWARD", "1") == "1":
            def _ready():                       # (the split backward knows the point itself: after the critic's backward for G and the join)
                ready.record(torch.cuda.current_stream(dev))
                fired.append(True)
            self.student_backward(_ready)
        else:
            handle = self.preds_S[0].register_hook(_logits_grad_ready)
            try:
                self.student_backward()
            finally:
                handle.remove()
        self.G_solver.step()
        if fired:
            side.wait_event(ready)
        else:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            self.discriminator_backward()
        self._publish_scalars(side)         # the G step's scalars were produced on the main stream before `ready` was recorded
        main.wait_stream(side)
        self._finish_teacher()              # the teacher's deep-supervision branch (long finished: issued behind the join event)

    def evalute_model(self, model, loader, gpu_id, input_size, num_classes, whole):
        """networks/evaluate.py via kd_model.py:178-181.  One process per GPU: the replicas are identical, so the validation
        set is SHARDED over the ranks (batch i on rank i % world) and the integer confusion matrices are all-reduced --
        every rank returns the same (mean_IU, IU_array), nobody idles at a barrier while rank 0 walks 500 images."""
        from .evaluate import evaluate_main
        return tuple(evaluate_main(model=model, loader=loader, gpu_id=gpu_id, input_size=input_size, num_classes=num_classes,
                                   whole=whole, rank=parallel_old.rank(), world=parallel_old.world_size()))

    def print_info(self, epoch, step):
        logging.info("step:{:5d} G_lr:{:.6f} G_loss:{:.5f}(mc:{:.5f} pixelwise:{:.5f} pairwise:{:.5f}) "
                     "D_lr:{:.6f} D_loss:{:.5f}".format(
                         step, self.G_solver.param_groups[-1]["lr"], self.G_loss, self.mc_G_loss,
                         self.pi_G_loss, self.pa_G_loss, self.D_solver.param_groups[-1]["lr"], self.D_loss))

    def save_ckpt(self, epoch, step, mean_IU, IU_array):
        """kd_model.py:192-193; written by rank 0 only (every rank holds the same weights), then a barrier so nobody
        races ahead into a step while the file is incomplete."""
        if parallel_old.rank() == 0:
            torch.save(self.student.state_dict(),
                       osp.join(self.args.snapshot_dir, "CS_scenes_" + str(step) + "_" + str(mean_IU) + ".pth"))
        if parallel_old.replicated():
            torch.distributed.barrier()


def default_args(**overrides):
    """The flag defaults of utils/train_options.py:18-63 that shape the step, as a namespace
    (``run_train_val.sh`` overrides: weight_decay 5e-4, lambda_pa 0.5)."""
    import argparse
    a = argparse.Namespace(
        classes_num=19, batch_size=8, input_size="512,512", momentum=0.9, num_steps=40000, power=0.9,
        weight_decay=1e-4, lr_g=1e-2, lr_d=4e-4, pi=True, pa=True, ho=True, lambda_pi=10.0, lambda_pa=1.0,
        lambda_d=0.1, lambda_gp=10.0, pool_scale=0.5, adv_loss_type="wgan-gp", imsize_for_adv=65,
        adv_conv_dim=64, preprocess_GAN_mode=1, parallel="True", gpu="0", gpu_num=1, best_mean_IU=0.0,
        T_ckpt_path=None, is_student_load_imgnet=False, student_pretrain_model_imgnet=None,
        S_resume=True, S_ckpt_path=None, D_resume=True, D_ckpt_path=None, last_step=0, start_epoch=0,
        snapshot_dir=None, device=torch.device("cuda" if torch.cuda.is_available() else "cpu"))
    for k, v in overrides.items():
        setattr(a, k, v)
    return a
