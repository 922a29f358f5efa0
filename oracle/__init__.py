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
The native InPlace-ABN kernels (libs/src/bn.cu + common.h + bn.h) are pinned by RUNNING them: their own build
(nvcc + THC + torch.utils.ffi) is impossible here, but the three source files are self-contained CUDA-runtime code
that hipcc compiles for gfx950 once the runtime-API names are respelled (``oracle/build_ref.py``: the substitution
table is the whole recipe; nothing of the reference is copied into the repository).  The result,
``oracle/_ref/libbn_ref.so`` (git-ignored, built by ``__graft_entry__.build()`` where /root/reference exists, shipped
to the GPU box), is compared three ways -- reference kernels vs ``oracle/abn_ref.c`` vs ``libskd_hip.so`` -- by
``tests/test_ref_kernels_gpu.py`` on the nine entry points of libs/src/bn.h.
"""
