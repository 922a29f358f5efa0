"""Host-side compatibility contract of SURVEY.md section 5: state-dict keys / shapes of the three networks are the
reference's, the teacher-checkpoint key remap (utils/utils.py:78-87) and the ImageNet key-intersection load
(utils/utils.py:97-104) behave like the reference's helpers, and default_args() mirrors train_options.py."""
import os

import pytest
import torch

from oracle import abn_torch, ref_import, step_torch as O
from structure_knowledge_distillation_amd.networks import kd_model, pspnet_combine as PC, sagan_models


def test_state_dict_keys_and_shapes_match_oracle_layout():
    for arch, block, layers in ((O.STUDENT, PC.BasicBlock, [2, 2, 2, 2]), (O.TEACHER, PC.Bottleneck, [3, 4, 23, 3])):
        net = PC.Res_pspnet(block, layers, 19)
        want = O.pspnet_init(arch, 19)
        sd = net.state_dict()
        assert sorted(sd) == sorted(want)
        assert all(tuple(sd[k].shape) == tuple(want[k].shape) for k in sd)
        assert not any(k.endswith("num_batches_tracked") for k in sd)       # libs/bn.py has no such buffer
    d = sagan_models.Discriminator(1, 19, 8, 65, 64)
    want = O.discriminator_init()
    assert sorted(d.state_dict()) == sorted(want)
    assert len(PC.Res_pspnet(PC.BasicBlock, [2, 2, 2, 2], 19).state_dict()) == 150
    assert len(PC.Res_pspnet(PC.Bottleneck, [3, 4, 23, 3], 19).state_dict()) == 565 and len(d.state_dict()) == 37


@pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not present")
def test_state_dict_keys_match_the_reference_modules():
    ref = ref_import.load_reference(abn_torch)
    for block_ref, block, layers in ((ref.pspnet.BasicBlock, PC.BasicBlock, [2, 2, 2, 2]),
                                     (ref.pspnet.Bottleneck, PC.Bottleneck, [3, 4, 23, 3])):
        a = ref.pspnet.Res_pspnet(block_ref, layers, 19).state_dict()
        b = PC.Res_pspnet(block, layers, 19).state_dict()
        assert list(a) == list(b), "same keys in the same order"
        assert all(a[k].shape == b[k].shape for k in a)
    a = ref.sagan.Discriminator(1, 19, 8, 65, 64).state_dict()
    b = sagan_models.Discriminator(1, 19, 8, 65, 64).state_dict()
    assert sorted(a) == sorted(b) and all(a[k].shape == b[k].shape for k in a)
    # a reference checkpoint loads into the product module and back
    student_ref = ref.pspnet.Res_pspnet(ref.pspnet.BasicBlock, [2, 2, 2, 2], 19)
    mine = PC.Res_pspnet(PC.BasicBlock, [2, 2, 2, 2], 19)
    mine.load_state_dict(student_ref.state_dict())
    student_ref.load_state_dict(mine.state_dict())


def test_teacher_checkpoint_key_remap(tmp_path):
    """utils/utils.py:78-87: released teacher checkpoints keep the PSP module under head.0.* and the classifier
    under head.1.*."""
    teacher = PC.Res_pspnet(PC.Bottleneck, [3, 4, 23, 3], 19)
    sd = teacher.state_dict()
    saved = {}
    for k, v in sd.items():
        if k.startswith("pspmodule."):
            saved["head.0." + k[len("pspmodule."):]] = v + 1.0
        elif k.startswith("head."):
            saved["head.1." + k[len("head."):]] = v + 1.0
        else:
            saved[k] = v + 1.0
    path = os.path.join(tmp_path, "teacher.pth")
    torch.save(saved, path)
    fresh = PC.Res_pspnet(PC.Bottleneck, [3, 4, 23, 3], 19)
    before = {k: v.clone() for k, v in fresh.state_dict().items()}
    assert kd_model.load_T_model(fresh, path)
    after = fresh.state_dict()
    for k in ("pspmodule.bottleneck.0.weight", "head.weight", "layer3.5.conv2.weight", "pspmodule.stages.2.2.running_var"):
        assert torch.equal(after[k], sd[k] + 1.0) and not torch.equal(after[k], before[k]), k
    assert not kd_model.load_T_model(fresh, os.path.join(tmp_path, "missing.pth"))


def test_student_imagenet_intersection_load(tmp_path):
    student = PC.Res_pspnet(PC.BasicBlock, [2, 2, 2, 2], 19)
    own = student.state_dict()
    saved = {"conv1.weight": torch.full_like(own["conv1.weight"], 0.5), "layer1.0.conv1.weight": torch.zeros(3, 3),   # wrong shape
             "fc.weight": torch.zeros(1000, 512)}                                                                   # unknown key
    path = os.path.join(tmp_path, "resnet18.pth")
    torch.save(saved, path)
    args = kd_model.default_args(is_student_load_imgnet=True, student_pretrain_model_imgnet=path, device=torch.device("cpu"))
    assert kd_model.load_S_model(args, student)
    assert torch.equal(student.state_dict()["conv1.weight"], saved["conv1.weight"])
    assert torch.equal(student.state_dict()["layer1.0.conv1.weight"], own["layer1.0.conv1.weight"])


def test_default_args_mirror_train_options():
    """utils/train_options.py:18-63 defaults that shape the step."""
    a = kd_model.default_args()
    want = dict(classes_num=19, batch_size=8, momentum=0.9, num_steps=40000, power=0.9, weight_decay=1e-4, lr_g=1e-2, lr_d=4e-4,
                pi=True, pa=True, ho=True, lambda_pi=10.0, lambda_pa=1.0, lambda_d=0.1, lambda_gp=10.0, pool_scale=0.5,
                adv_loss_type="wgan-gp", imsize_for_adv=65, adv_conv_dim=64, preprocess_GAN_mode=1, parallel="True")
    for k, v in want.items():
        assert getattr(a, k) == v, k


def test_student_and_discriminator_resume_round_trip(tmp_path):
    """utils/utils.py:105-151: S_resume / D_resume load <ckpt dir>/model_best.pth.tar -- a dict with 'state_dict' whose
    keys carry the nn.DataParallel 'module.' prefix when with_module is False -- and restore the counters that
    train_and_eval.py:20-21 resumes from.  A restarted run must NOT silently start from step 0."""
    student = PC.Res_pspnet(PC.BasicBlock, [2, 2, 2, 2], 19)
    sdir, ddir = os.path.join(tmp_path, "Student"), os.path.join(tmp_path, "Distriminator")
    os.makedirs(sdir)
    os.makedirs(ddir)
    saved = {"module." + k: v + 0.25 for k, v in student.state_dict().items()}
    torch.save({"state_dict": saved, "step": 1234, "epoch": 3, "best_mean_IU": 0.61, "IU_array": [0.5] * 19},
               os.path.join(sdir, "model_best.pth.tar"))
    d = sagan_models.Discriminator(1, 19, 8, 65, 64)
    torch.save({"state_dict": {"module." + k: v * 0 + 0.5 for k, v in d.state_dict().items()}, "epoch": 4, "best_mean_IU": 0.7},
               os.path.join(ddir, "model_best.pth.tar"))
    args = kd_model.default_args(S_ckpt_path=sdir, D_ckpt_path=ddir, device=torch.device("cpu"))
    assert args.S_resume and args.D_resume                                          # train_options.py:23,25 defaults
    fresh = PC.Res_pspnet(PC.BasicBlock, [2, 2, 2, 2], 19)
    want = {k: v + 0.25 for k, v in student.state_dict().items()}
    assert kd_model.load_S_model(args, fresh, False) == "resume"
    assert (args.last_step, args.start_epoch, args.best_mean_IU) == (1234, 3, 0.61)
    assert all(torch.equal(fresh.state_dict()[k], want[k]) for k in want)
    d2 = sagan_models.Discriminator(1, 19, 8, 65, 64)
    assert kd_model.load_D_model(args, d2, False)
    assert (args.start_epoch, args.best_mean_IU) == (4, 0.7)
    assert all(bool((v == 0.5).all()) for k, v in d2.state_dict().items() if v.is_floating_point())
    # with_module=True takes the keys as they are (utils.py:121-122)
    torch.save({"state_dict": student.state_dict()}, os.path.join(sdir, "model_best.pth.tar"))
    assert kd_model.load_S_model(args, fresh, True) == "resume" and args.last_step is None
    # nothing to resume from: says so, loads nothing, keeps the counters
    empty = kd_model.default_args(S_ckpt_path=os.path.join(tmp_path, "none"), D_ckpt_path=os.path.join(tmp_path, "noneD"),
                                  device=torch.device("cpu"), last_step=7)
    assert kd_model.load_S_model(empty, fresh, False) is False and empty.last_step == 7
    assert kd_model.load_D_model(empty, d2, False) is False
    # the ImageNet branch wins over S_resume, as in the reference (utils.py:97 vs 105)
    args.is_student_load_imgnet, args.student_pretrain_model_imgnet = True, os.path.join(tmp_path, "missing.pth")
    assert kd_model.load_S_model(args, fresh, False) is False
