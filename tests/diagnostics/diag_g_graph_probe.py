"""Diagnostic (not collected): is the student's half of the step worth a hipGraph?  Captures student forward + criteria + G backward
on the capture stream WITH the D step (critic forwards, WGAN-GP double backward, d_loss.backward()) as a parallel branch forked at the
point where the logits have received their gradient (what the eager step does with an event and a second stream), replays it next to
the teacher's graph, and times it against the eager step on the same box.
    python tests/diagnostics/diag_g_graph_probe.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    dev = torch.device("cuda", 0)
    B, S = 8, 512
    args = default_args(batch_size=B, device=dev, weight_decay=5e-4, lambda_pa=0.5, num_steps=40000)
    gen = torch.Generator().manual_seed(100)
    images = (torch.randn(B, 3, S, S, generator=gen) * 57.0).to(dev)
    labels = torch.randint(0, 19, (B, S, S), generator=gen)
    labels[0, : S // 16] = 255
    labels = labels.to(dev)
    torch.manual_seed(1234)
    model = NetModel(args)

    def eager_step(i):
        model.adjust_learning_rate(args.lr_g, model.G_solver, i)
        model.adjust_learning_rate(args.lr_d, model.D_solver, i)
        model.set_input((images, labels, None, None))
        model.optimize_parameters()

    for i in range(5):
        eager_step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eager_step(5 + i)
    torch.cuda.synchronize()
    eager_ms = 1e3 * (time.perf_counter() - t0) / steps
    eager_losses = [model.G_loss, model.mc_G_loss, model.pi_G_loss, model.pa_G_loss, model.D_loss]
    print("eager: %.3f ms/step  losses %s" % (eager_ms, ["%.5g" % v for v in eager_losses]), flush=True)

    # ---- capture ----
    side = model._d_stream
    model.set_input((images, labels, None, None))
    model.preds_T = model._teacher_forward()                 # static outputs of the teacher graph
    model.G_solver.zero_grad(set_to_none=True)
    model.D_solver.zero_grad(set_to_none=True)
    # the last eager step's autograd graph (held by preds_S) keeps the parameters' AccumulateGrad nodes alive -- created on the
    # DEFAULT stream; a capture that reused them would synchronise with that stream.  Drop it: the captured forward creates new ones
    model.preds_S = None
    import gc
    gc.collect()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        cap = torch.cuda.current_stream(dev)
        model.preds_S = model._student_forward()
        ev = torch.cuda.Event()

        def fork(grad):
            ev.record(torch.cuda.current_stream(grad.device))
            side.wait_event(ev)

        mode = os.environ.get("PROBE_MODE", "fork")
        if mode == "fork":
            h = model.preds_S[0].register_hook(fork)
            model.student_backward()
            h.remove()
            with torch.cuda.stream(side):
                d_loss = model._d_loss(model.preds_S[0].detach(), model.preds_T[0].detach(), model.gp_alpha)
                d_loss.backward()
                model._scalars["D_loss"] = d_loss.detach()
            cap.wait_stream(side)
        elif mode == "serial":
            model.student_backward()
            d_loss = model._d_loss(model.preds_S[0].detach(), model.preds_T[0].detach(), model.gp_alpha)
            d_loss.backward()
            model._scalars["D_loss"] = d_loss.detach()
        else:                                                  # "gonly": the student's half alone; the D step stays eager behind the replay
            model.student_backward()
    static_scalars = dict(model._scalars)
    g_grads = [(p, p.grad) for p in list(model._s_params) + list(model._d_params)]
    print("captured", flush=True)

    def graph_step(i):
        model.adjust_learning_rate(args.lr_g, model.G_solver, i)
        model.adjust_learning_rate(args.lr_d, model.D_solver, i)
        model.set_input((images, labels, None, None))
        model.preds_T = model._teacher_forward()
        graph.replay()
        for p, g in g_grads:
            if g is not None:
                p.grad = g
        model.G_solver.step()
        if os.environ.get("PROBE_MODE", "fork") == "gonly":
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                model.discriminator_backward()
            torch.cuda.current_stream(dev).wait_stream(side)
        else:
            model.D_solver.step()
        model._scalars.update(static_scalars)
        model._publish_scalars()

    for i in range(3):
        graph_step(5 + steps + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        graph_step(8 + steps + i)
    torch.cuda.synchronize()
    graph_ms = 1e3 * (time.perf_counter() - t0) / steps
    losses = [model.G_loss, model.mc_G_loss, model.pi_G_loss, model.pa_G_loss, model.D_loss]
    print("graph: %.3f ms/step  losses %s" % (graph_ms, ["%.5g" % v for v in losses]), flush=True)
    # and eager again (same process, warm): the A/B
    for i in range(3):
        eager_step(100 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eager_step(103 + i)
    torch.cuda.synchronize()
    print("eager again: %.3f ms/step" % (1e3 * (time.perf_counter() - t0) / steps), flush=True)
    for i in range(3):
        graph_step(200 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        graph_step(203 + i)
    torch.cuda.synchronize()
    print("graph again: %.3f ms/step" % (1e3 * (time.perf_counter() - t0) / steps), flush=True)


if __name__ == "__main__":
    main()
