"""GPU tool: which torch operators launch the small element-wise / reduction / copy kernels of the step?

    python tools/op_profile.py > gpurun_out/op_profile.txt

torch.profiler over two steps of the benchmarked configuration (after three warm-up steps): operators sorted by device time,
with call counts -- the question is which of the ~700 stock launches per step are avoidable host-side choices."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    import torch
    from torch.profiler import profile, ProfilerActivity
    from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    args = default_args(batch_size=8, device=dev, weight_decay=5e-4, lambda_pa=0.5, num_steps=40000)
    model = NetModel(args)
    gen = torch.Generator().manual_seed(100)
    images = (torch.randn(8, 3, 512, 512, generator=gen) * 57.0).to(dev)
    labels = torch.randint(0, 19, (8, 512, 512), generator=gen).to(dev)

    def step(i):
        model.adjust_learning_rate(args.lr_g, model.G_solver, i)
        model.adjust_learning_rate(args.lr_d, model.D_solver, i)
        model.set_input((images, labels, None, None))
        model.optimize_parameters()
        return model.G_loss, model.D_loss

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for i in range(2):
            step(3 + i)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=70))
    # who calls the small stock operators: grouped by the innermost Python frames and by input shapes
    small = ("aten::sum", "aten::add_", "aten::add", "aten::mul", "aten::copy_", "aten::fill_", "aten::zero_", "aten::div", "aten::neg")
    rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6) if e.key in small]
    rows.sort(key=lambda e: -e.self_device_time_total)
    print("\n==== callers of the small stock operators (2 steps) ====")
    for e in rows[:60]:
        stack = [f for f in (e.stack or []) if "structure_knowledge_distillation_amd" in f or "torch/optim" in f or "autograd" in f][:3]
        print("%-12s x%-4d %8.1f us  shapes %s\n      %s" % (e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:90], " <- ".join(s.strip()[-110:] for s in stack)))
