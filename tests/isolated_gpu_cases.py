"""Cases of tests/test_step_gpu.py that run in their OWN process with a hard time limit (see the comment in front of
``_run_isolated`` there).  NOT collected by ``pytest tests`` (the file name does not match test_*.py): the wrappers call
``python -m pytest tests/isolated_gpu_cases.py::case_...``."""
import os

import pytest
import torch

from oracle import step_torch as O
from structure_knowledge_distillation_amd.networks.kd_model import NetModel, default_args

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)

# a case that hangs says WHERE before its wrapper kills it: the Python stack of every thread goes to stderr shortly before the
# wrapper's time limit (tests/test_step_gpu.py::_run_isolated passes the limit in SKD_ISOLATED_LIMIT_S)
import faulthandler
import sys

_limit = float(os.environ.get("SKD_ISOLATED_LIMIT_S", "0"))
if _limit > 30:
    faulthandler.dump_traceback_later(_limit - 12, exit=False, file=sys.stderr)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cpu_sd(mod, dtype=torch.float32):
    return {k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu().clone())
            for k, v in mod.state_dict().items()}


def case_teacher_hipgraph_equals_eager_bit_for_bit_in_deterministic_mode(monkeypatch):
    """SKD_TEACHER_GRAPH (default on): the frozen teacher's forward captured once into a hipGraph and replayed.  Same kernels,
    same order: under SKD_DETERMINISTIC=1 two steps -- the capture step and a REPLAY on a new batch -- give the same bits as
    the eager forward in every teacher output, every loss and every student / discriminator tensor."""
    monkeypatch.setenv("SKD_DETERMINISTIC", "1")
    try:
        def run(flag):
            monkeypatch.setenv("SKD_TEACHER_GRAPH", flag)
            torch.manual_seed(99)
            args = default_args(batch_size=2, device=DEV, ho=True, weight_decay=5e-4, lambda_pa=0.5)
            model = NetModel(args)
            assert model.deterministic and model._teacher_graph_on == (flag == "force")
            losses, preds = [], []
            for step in range(2):
                images, labels = O.synthetic_batch(2, 512, 512, seed=step)
                model.gp_alpha = torch.rand(2, 1, 1, 1, generator=torch.Generator().manual_seed(70 + step)).to(DEV)
                torch.manual_seed(500 + step)
                model.set_input((images, labels, None, None))
                model.optimize_parameters()
                losses.append([model.G_loss, model.mc_G_loss, model.pi_G_loss, model.pa_G_loss, model.D_loss])
                preds.append([None if t is None else t.detach().clone() for t in model.preds_T])
            torch.cuda.synchronize()
            assert len(model._teacher_graphs) == (1 if flag == "force" else 0)
            return losses, preds, cpu_sd(model.student), cpu_sd(model.D_model)

        eager, graph = run("0"), run("force")
        assert eager[0] == graph[0], (eager[0], graph[0])
        for step in range(2):
            for i, (a, b) in enumerate(zip(eager[1][step], graph[1][step])):
                assert (a is None and b is None) or torch.equal(a, b), "teacher output %d differs in step %d (capture / replay)" % (i, step)
        for which, what in ((2, "student"), (3, "D")):
            diff = [k for k, v in eager[which].items() if not torch.equal(v, graph[which][k])]
            assert not diff, "%s state differs with the teacher replayed from a hipGraph: %s" % (what, diff[:8])
    finally:
        torch.backends.cudnn.enabled = True
        torch.use_deterministic_algorithms(False)


def case_teacher_hipgraph_default_mode_replays_and_follows_weight_writes():
    """Default mode (MIOpen convolutions): the graph is on, replays track the eager forward on fresh inputs, and writing the
    teacher's tensors (load_state_dict after construction) drops the captured graph instead of replaying stale folded weights."""
    torch.manual_seed(7)
    args = default_args(batch_size=2, device=DEV, ho=True, weight_decay=5e-4, lambda_pa=0.5)
    model = NetModel(args)
    assert model._teacher_graph_on
    for step in range(3):
        images, labels = O.synthetic_batch(2, 512, 512, seed=10 + step)
        model.set_input((images, labels, None, None))
        got = model._teacher_forward()
        want = model._teacher_forward_eager(model.images)
        for a, b in zip(got[:4], want[:4]):
            assert rel(a, b) < 2e-5                       # (MIOpen's forward kernels are not bit-reproducible run to run)
    assert len(model._teacher_graphs) == 1
    first = next(iter(model._teacher_graphs.values()))
    sd = {k: v.clone() for k, v in model.teacher.state_dict().items()}
    with torch.no_grad():
        sd["head.bias"] += 1.0                            # a visible change of the logits
    model.teacher.load_state_dict(sd)
    got = model._teacher_forward()
    assert next(iter(model._teacher_graphs.values())) is not first, "a written teacher must be re-captured"
    want = model._teacher_forward_eager(model.images)
    assert rel(got[0], want[0]) < 2e-5
    model.optimize_parameters()                           # the whole step on top of a replayed teacher
    assert all(v == v for v in (model.G_loss, model.D_loss))
