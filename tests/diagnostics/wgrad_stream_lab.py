"""Lab (not collected, not product): weight gradients of the student's plain convolutions on a HIP stream of their own.

The convolution becomes its own autograd node: data gradient (torch.nn.grad.conv2d_input) on the backward's stream, weight gradient
(conv2d_weight) on a second stream behind an event that marks the incoming gradient as ready; ``join()`` before the update.
    WGRAD_PRIORITY=<int> python tests/diagnostics/wgrad_stream_lab.py [steps]      # torch.cuda.Stream(priority=...): 0 normal, -1 high, 1 low
Prints ms per step of bench.py's loop with and without it (same process, alternating)."""
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.grad import conv2d_input, conv2d_weight

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


class Holder:
    def __init__(self, device, priority):
        self.stream = torch.cuda.Stream(device=device, priority=priority)
        self.used = False
        self.on = True

    def join(self):
        if self.used:
            torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)
            self.used = False


class _ConvSplitBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, conv, holder):
        ctx.conv, ctx.holder = conv, holder
        ctx.save_for_backward(x, weight)
        return F.conv2d(x, weight, None, conv.stride, conv.padding, conv.dilation, 1)

    @staticmethod
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        conv, holder = ctx.conv, ctx.holder
        gx = gw = None
        if ctx.needs_input_grad[1]:
            main = torch.cuda.current_stream(go.device)
            ready = torch.cuda.Event()
            ready.record(main)
            side = holder.stream
            side.wait_event(ready)
            with torch.cuda.stream(side):
                gw = conv2d_weight(x, weight.shape, go, conv.stride, conv.padding, conv.dilation, 1)
            go.record_stream(side)
            x.record_stream(side)
            gw.record_stream(main)
            holder.used = True
        if ctx.needs_input_grad[0]:
            gx = conv2d_input(x.shape, weight, go, conv.stride, conv.padding, conv.dilation, 1)
        return gx, gw, None, None


def enable(module, holder):
    n = 0
    for m in module.modules():
        if (type(m) is nn.Conv2d and m.groups == 1 and m.bias is None and m.padding_mode == "zeros" and not isinstance(m.padding, str)):
            def fwd(x, _m=m):
                if holder.on and x.is_cuda and torch.is_grad_enabled() and _m.weight.requires_grad:
                    return _ConvSplitBackward.apply(x, _m.weight, _m, holder)
                return nn.Conv2d.forward(_m, x)
            m.forward = fwd
            n += 1
    return n


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    dev = torch.device("cuda", 0)
    B, S = 8, 512
    args = default_args(batch_size=B, device=dev, weight_decay=5e-4, lambda_pa=0.5, num_steps=40000)
    gen = torch.Generator().manual_seed(100)
    images = (torch.randn(B, 3, S, S, generator=gen) * 57.0).to(dev)
    labels = torch.randint(0, 19, (B, S, S), generator=gen).to(dev)
    torch.manual_seed(1234)
    model = NetModel(args)
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (None, None)
    prio = int(os.environ.get("WGRAD_PRIORITY", "0"))
    holder = Holder(dev, prio)
    n = enable(model.student, holder)
    print("convolutions switched: %d, wgrad stream priority %d (range %s .. %s)" % (n, holder.stream.priority, lo, hi), flush=True)
    # join before the update: wrap G_solver.step
    g_step = model.G_solver.step

    def joined_step(*a, **k):
        holder.join()
        return g_step(*a, **k)

    model.G_solver.step = joined_step

    def step(i):
        model.adjust_learning_rate(args.lr_g, model.G_solver, i)
        model.adjust_learning_rate(args.lr_d, model.D_solver, i)
        model.set_input((images, labels, None, None))
        model.optimize_parameters()
        return model.G_loss

    def timed(on, base):
        holder.on = on
        for i in range(4):
            step(base + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(base + 4 + i)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    for rep in range(3):
        a = timed(False, 100 * rep)
        b = timed(True, 100 * rep + 50)
        print("rep %d: stock backward %.3f ms/step, weight gradients on their own stream %.3f ms/step" % (rep, a, b), flush=True)


if __name__ == "__main__":
    main()
