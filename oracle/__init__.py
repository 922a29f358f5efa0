"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's distillation-step algorithms
(irfanICMLL/structure_knowledge_distillation), used as the parity checker for
the HIP path in ``structure_knowledge_distillation_amd``.

Rules (enforced by tests/test_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import anything from this package;
  * nothing under ``structure_knowledge_distillation_amd/`` imports it;
  * it never touches a GPU.

Parity pinning: the reference ships no tests, golden vectors or fixtures
(SURVEY.md section 4 / 8c).  The restatements here are pinned by running the
reference's own Python (``oracle/ref_import.py``, only possible in the build
container where /root/reference exists) on seeded inputs and committing the
results under ``tests/golden/`` (generator: ``tests/golden/make_golden.py``).
The native InPlace-ABN kernels (libs/src/bn.cu) cannot be built or run here
(CUDA + THC + torch.utils.ffi), so for those four kernels the pin is the
formula-level restatement in ``oracle/abn_ref.c`` checked against plain
autograd of the same closed form: **parity unpinned by any reference-run
output** for bn.cu itself.
"""
