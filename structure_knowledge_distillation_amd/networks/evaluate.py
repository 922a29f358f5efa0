"""Whole-image evaluation (mIoU) on MI355X -- the ``evaluate_main`` that ``train_and_eval.py:28`` reaches through
``NetModel.evalute_model``.

Reference: networks/evaluate.py:156-206 with ``whole=True`` (the only mode train_and_eval.py uses): per validation
image, student forward on the full 1024x2048 image, bilinear (align_corners) upsample of the 129x257 logits back to
1024x2048, argmax, confusion matrix over the non-ignored pixels, IoU = tp / max(1, pos + res - tp), mean over classes.
The reference materialises the up-sampled logits (159 MB), copies them to the host and runs numpy argmax/bincount;
here the upsample + argmax + confusion accumulation is one HIP kernel (csrc/evaluate.hip) and only the 19x19 int64
matrix ever leaves the GPU.  Sliding-window inference, multi-scale / flip averaging, the test-set id remap and the
palette PNG dump (evaluate.py:62-104, 115-134, 187-191) are host-side tooling around cv2 / scipy / PIL and are not
provided.
"""
import numpy as np
import torch

from .. import functional as SF

ignore_label = 255


def get_confusion_matrix(gt_label, pred_label, class_num):
    """evaluate.py:136-154 on host arrays (kept for API parity; evaluate_main uses the fused kernel)."""
    index = (np.asarray(gt_label).astype(np.int64) * class_num + np.asarray(pred_label).astype(np.int64))
    count = np.bincount(index.ravel(), minlength=class_num * class_num)[: class_num * class_num]
    return count.reshape(class_num, class_num).astype(np.float64)


def iou_from_confusion(confusion_matrix):
    """evaluate.py:200-206."""
    cm = np.asarray(confusion_matrix, dtype=np.float64)
    pos, res, tp = cm.sum(1), cm.sum(0), np.diag(cm)
    iu = tp / np.maximum(1.0, pos + res - tp)
    return float(iu.mean()), iu


def evaluate_main(model, loader, gpu_id, input_size, num_classes, whole=False, recurrence=1, type="val", rank=0, world=1,
                  group=None):
    """Returns (mean_IU, IU_array) like evaluate.py:156-206.  ``loader`` yields (image (1,3,H,W) float, label (1,H,W),
    size, name) like CSDataSet (dataset/datasets.py:121-210); ``size[0][:2]`` is the valid (h, w) of the label.
    ``rank`` / ``world`` (one process per GPU): every rank walks the same loader but evaluates only the batches with
    ``index % world == rank``; the integer confusion matrices are summed over ``group`` with one all-reduce, so every
    rank returns the identical result of the whole validation set (the reference's single process evaluated alone)."""
    if not whole:
        raise NotImplementedError("sliding-window evaluation (evaluate.py:62-104) is not part of the MI355X path; "
                                  "train_and_eval.py evaluates with whole=True")
    if type != "val":
        raise NotImplementedError("test-split prediction dump (evaluate.py:187-191) is host-side tooling")
    if world > 1:
        # Sharding by ``index % world`` assumes every rank walks the SAME sequence of batches.  A loader that already shards
        # (DistributedSampler) or shuffles would have each rank skip most of its own subset and score a silently smaller set
        # (ADVICE r03): refuse it.
        sampler = getattr(loader, "sampler", None)
        from torch.utils.data import RandomSampler
        from torch.utils.data.distributed import DistributedSampler
        if isinstance(sampler, (DistributedSampler, RandomSampler)) or isinstance(getattr(loader, "batch_sampler", None), DistributedSampler):
            raise ValueError("evaluate_main shards the validation set itself (batch i on rank i %% world): hand it a plain, "
                             "unshuffled, unsharded loader (got sampler %s)" % sampler.__class__.__name__)   # (`type` is an argument here)
    device = torch.device("cuda", int(gpu_id) if str(gpu_id).isdigit() else 0) if torch.cuda.is_available() \
        else next(model.parameters()).device
    was_training = model.training
    model.eval()
    model.to(device)
    confusion = torch.zeros((num_classes, num_classes), dtype=torch.int64, device=device)
    # a network whose weights NetModel keeps channels-last (kd_model.py of this package) is fed channels-last images, so the
    # whole-image forward (features 129 x 257 at 1024 x 2048) stays on the NHWC ABN / pyramid / fold kernels
    w4 = [p for p in model.parameters() if p.dim() == 4 and p.shape[1] > 1 and p.shape[2] * p.shape[3] > 1]
    cl_model = device.type == "cuda" and bool(w4) and all(p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous() for p in w4)
    seen, failure = 0, None
    with torch.no_grad():
        for index, batch in enumerate(loader):
            if index % world != rank:
                continue
            if failure is not None:
                continue
            seen += 1
            try:
                _score_batch(model, batch, device, num_classes, cl_model, confusion)
            except Exception as e:                    # reported AFTER the collective below, so that no rank is left hanging in it
                if world <= 1:
                    raise
                failure = e
    if was_training:
        model.train()
    if world > 1:
        # one exchange: the confusion counts (exact, int64), how many batches were scored, and whether any rank failed
        extra = torch.tensor([seen, 1 if failure is not None else 0], dtype=torch.int64, device=device)
        flat = torch.cat([confusion.reshape(-1), extra])
        torch.distributed.all_reduce(flat, group=group)
        confusion = flat[:-2].reshape(num_classes, num_classes)
        total, failed = int(flat[-2]), int(flat[-1])
        if failure is not None:
            raise failure
        if failed:
            raise RuntimeError("evaluate_main: %d other rank(s) failed while scoring their share of the validation set" % failed)
        try:
            expected = len(loader)
        except TypeError:
            expected = None
        if expected is not None and total != expected:
            raise RuntimeError("evaluate_main: the ranks scored %d batches together but the loader holds %d" % (total, expected))
    return iou_from_confusion(confusion.cpu().numpy())


def _score_batch(model, batch, device, num_classes, cl_model, confusion):
    """One validation batch: forward on the whole image, fused upsample + argmax + confusion accumulation (csrc/evaluate.hip)."""
    image, label, size = batch[0], batch[1], batch[2]
    lab_np = np.asarray(label) if not torch.is_tensor(label) else label.numpy() if label.device.type == "cpu" else None
    if lab_np is not None and bool(((lab_np != ignore_label) & ((lab_np < 0) | (lab_np >= num_classes))).any()):
        # np.bincount of evaluate.py:188-198 would count such labels (and index out of the matrix); the fused
        # kernel skips them -- refuse instead of scoring a quietly smaller set (raw label ids not mapped to trainIds?)
        raise ValueError("label values outside [0, %d) other than ignore_label %d" % (num_classes, ignore_label))
    image = torch.as_tensor(np.asarray(image) if not torch.is_tensor(image) else image).float().to(device)
    label = torch.as_tensor(np.asarray(label) if not torch.is_tensor(label) else label).long().to(device)
    sz = np.asarray(size[0] if (torch.is_tensor(size) or isinstance(size, (list, tuple))) else size).reshape(-1)
    hh, ww = int(sz[0]), int(sz[1])
    if cl_model and image.dim() == 4:
        image = image.contiguous(memory_format=torch.channels_last)   # keep the network on its channels-last kernels
    logits = model(image)
    if isinstance(logits, (list, tuple)):
        logits = logits[0]
    # predict_whole upsamples to the tile size (1024, 2048); only [:h, :w] of the label is scored (evaluate.py:194)
    full = label.new_full(label.shape, ignore_label)
    full[:, :hh, :ww] = label[:, :hh, :ww]
    SF.seg_confusion(logits.float(), full, ignore_label, confusion, want_pred=False)
