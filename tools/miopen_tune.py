"""GPU tool (not product code): populate the in-tree MIOpen user find-db + kernel cache
(structure_knowledge_distillation_amd/miopen_db/) for the convolution shapes of the distillation step.

This ROCm image ships no gfx950 find-db / kernel-db, so on a fresh machine MIOpen either JIT-compiles
every solver it tries (find mode: minutes per shape) or falls back to untuned heuristics (immediate
mode: 58-70 TF/s on the dilated 3x3 convolutions instead of 106-124).  Running MIOpen's own tuner ONCE
here and committing its (small, text + sqlite) output gives every later process find-quality kernels
with no JIT: immediate mode consults the user find-db first.

    python tools/miopen_tune.py <out_dir> [phase ...]
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = {  # name: (batch, find, what, timeout_s)
    "teacher8": (8, True, "teacher", 500),
    "teacher8_nhwc": (8, True, "teacher_nhwc", 500),
    "student8": (8, True, "student", 900),
    "full8": (8, True, "full", 400),
    "full8_nhwc": (8, True, "full_nhwc", 900),
    "full2_nhwc": (2, False, "full_nhwc", 400),
    "full2": (2, False, "full", 400),
    "psp_fold": (8, True, "psp_fold", 900),   # the feature-map halves of the two PSP bottleneck convolutions (folded priors)
}


def child(phase):
    sys.path.insert(0, ROOT)
    batch, find, what, _ = PHASES[phase]
    os.environ["SKD_MIOPEN_FIND"] = "1" if find else "0"
    # (rounds 1-2 also tuned the NCHW forms of the problems -- phases without "_nhwc"; both networks are channels-last only since
    # round 5, so every phase now records the channels-last problems)
    import torch
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    t0 = time.time()
    if what == "psp_fold":
        import torch.nn.functional as F
        torch.backends.cudnn.benchmark = True
        for (b, hw) in ((8, 65), (2, 65), (2, 33)):
            for (cf, cm, train) in ((2048, 512, False), (512, 128, True)):
                x = torch.randn(b, cf, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
                w = torch.randn(cm, cf, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
                t1 = time.time()
                if train:
                    x.requires_grad_(True)
                    w.requires_grad_(True)
                    y = F.conv2d(x, w, None, 1, 1)
                    y.backward(torch.randn_like(y))
                else:
                    with torch.no_grad():
                        F.conv2d(x, w, None, 1, 1)
                torch.cuda.synchronize()
                print("  %d x %d x %dx%d -> %d: %.1f s" % (b, cf, hw, hw, cm, time.time() - t1), flush=True)
        print("phase %s done in %.1f s" % (phase, time.time() - t0), flush=True)
        return
    args = default_args(batch_size=batch, device=dev, ho=(what in ("full", "full_nhwc")), weight_decay=5e-4, lambda_pa=0.5)
    model = NetModel(args)
    x = torch.randn(batch, 3, 512, 512, device=dev) * 57
    y = torch.randint(0, 19, (batch, 512, 512), device=dev)
    model.set_input((x, y, None, None))
    if what in ("teacher", "teacher_nhwc"):
        with torch.no_grad():      # the teacher's problems only (the student forward is already in the db)
            imgs = model.images.contiguous(memory_format=torch.channels_last) if model.teacher_nhwc else model.images
            model.parallel_teacher.eval()(imgs)
    else:
        for _ in range(2):
            model.optimize_parameters()
    torch.cuda.synchronize()
    print("phase %s done in %.1f s" % (phase, time.time() - t0), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(sys.argv[2])
        sys.exit(0)
    out = os.path.abspath(sys.argv[1])
    phases = sys.argv[2:] or list(PHASES)
    os.makedirs(os.path.join(out, "cache"), exist_ok=True)
    env = dict(os.environ, MIOPEN_USER_DB_PATH=out, MIOPEN_CUSTOM_CACHE_DIR=os.path.join(out, "cache"))
    for ph in phases:
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, __file__, "child", ph], env=env, timeout=PHASES[ph][3],
                               capture_output=True, text=True)
            msg = (r.stdout.strip().splitlines() or ["(no output)"])[-1] + (" | rc=%d %s" % (r.returncode, r.stderr[-400:]) if r.returncode else "")
        except subprocess.TimeoutExpired:
            msg = "phase %s TIMEOUT after %.0f s" % (ph, time.time() - t0)
        print(msg, flush=True)
        os.system("du -sh %s | tail -1" % out)
