"""Runs the REFERENCE's own train_and_eval.py -- the file itself, executed from /root/reference with runpy, never copied --
against this package through the module swap of INTEGRATION.md section 1.

    python tests/integration/run_reference_loop.py <workdir> [train_and_eval.py flags ...]

What is supplied around the reference's file (everything else -- flag parsing in utils/train_options.py, the DataLoader
construction, ``NetModel(args)``, the loop of train_and_eval.py:19-29 with adjust_learning_rate x2 / set_input /
optimize_parameters / print_info / evalute_model / save_ckpt -- is the reference's text):
  * the INTEGRATION.md section 1 ``sys.modules`` swap, verbatim;
  * ``tensorboardX`` (absent from this image; utils/utils.py:10 imports it for a dead function);
  * a stub loader: ``dataset.datasets.CSDataSet`` with the reference's constructor signature (datasets.py:121-123) that
    yields seeded synthetic (image, label, size, name) samples instead of reading Cityscapes through cv2;
  * with no GPU: the C-ABI double (oracle/libskd_ref.so) behind the same ctypes signatures, as in the other CPU tests.
Prints one JSON line with the logged scalars of the last step and the files the loop wrote.
"""
import json
import os
import runpy
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("SKD_REFERENCE_ROOT", "/root/reference")


def main():
    global sys                                          # the verbatim INTEGRATION.md block below re-imports it
    work = sys.argv[1]
    flags = sys.argv[2:]
    sys.dont_write_bytecode = True                      # nothing is written into the reference tree
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from torch.utils import data

    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = object
    sys.modules["tensorboardX"] = tbx

    from structure_knowledge_distillation_amd import _lib
    if not torch.cuda.is_available():
        from oracle import cref
        _lib.install_test_backend(cref.load(_lib.SIGNATURES))

    # ---- INTEGRATION.md section 1, verbatim ----------------------------------------------------------------------
    import sys, structure_knowledge_distillation_amd as skd  # noqa: E401,F401,F811
    from structure_knowledge_distillation_amd import libs, networks, utils, dataset  # noqa: F401
    from structure_knowledge_distillation_amd.networks import kd_model, pspnet_combine, sagan_models, spectral
    from structure_knowledge_distillation_amd.utils import criterion, parallel
    sys.modules["libs"] = libs                                  # pspnet_combine.py:11
    sys.modules["networks.kd_model"] = kd_model                 # train_and_eval.py:3
    sys.modules["networks.pspnet_combine"] = pspnet_combine
    sys.modules["networks.sagan_models"] = sagan_models
    sys.modules["networks.spectral"] = spectral
    sys.modules["utils.criterion"] = criterion                  # kd_model.py:18
    sys.modules["utils.parallel"] = parallel                    # kd_model.py:20
    # ---------------------------------------------------------------------------------------------------------------

    class CSDataSet(data.Dataset):
        """Stub loader with the constructor of dataset/datasets.py:121-123; sample layout of :205-210."""

        def __init__(self, root, list_path, max_iters=None, crop_size=(321, 321), mean=(128, 128, 128), scale=True,
                     mirror=True, ignore_label=255):
            self.crop_h, self.crop_w = crop_size
            self.n = int(max_iters) if max_iters else 1
            self.is_val = not max_iters
            self.ignore_label = ignore_label

        def __len__(self):
            return self.n

        def __getitem__(self, index):
            g = np.random.RandomState(1000 + index)
            h, w = (128, 256) if self.is_val else (self.crop_h, self.crop_w)    # a small "whole image" for the validation pass
            image = (g.randn(3, h, w) * 57.0).astype(np.float32)
            label = g.randint(0, 19, size=(h, w)).astype(np.float32)
            label[: max(1, h // 16)] = self.ignore_label
            return image.copy(), label.copy(), np.array((h, w, 3)), "synthetic_%d" % index

    ds_mod = types.ModuleType("dataset.datasets")
    ds_mod.CSDataSet = CSDataSet
    sys.modules["dataset.datasets"] = ds_mod                    # train_and_eval.py:8 (the reference's module needs cv2)

    os.chdir(work)                                              # train_options.py:66-74 creates ./ckpt/..., kd_model ./snapshots/
    sys.path.insert(0, REF)                                     # utils.train_options, utils.utils, the (empty) package inits
    sys.argv = ["train_and_eval.py"] + flags
    torch.manual_seed(2024)
    g = runpy.run_path(os.path.join(REF, "train_and_eval.py"), run_name="__main__")
    model = g["model"]
    written = sorted(os.path.join(dp, f)[2:] for dp, _, fs in os.walk(".") for f in fs)
    print(json.dumps({"losses": {k: float(getattr(model, k)) for k in ("G_loss", "mc_G_loss", "pi_G_loss", "pa_G_loss", "D_loss")},
                      "steps_seen": int(g["step"]), "lr_g": model.G_solver.param_groups[0]["lr"], "written": written,
                      "netmodel_module": type(model).__module__}), flush=True)


if __name__ == "__main__":
    main()
