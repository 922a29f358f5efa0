#!/bin/bash
# cross-compiles tools/gemm_lab (host C++ + HIP runtime + hipBLASLt, linked against the in-tree libskd_hip.so by rpath)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
python -m structure_knowledge_distillation_amd.build > /dev/null
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -I $R/include $R/tools/gemm_lab.cpp $R/tools/gemm_lab_kernels.hip -o $R/tools/gemm_lab \
  -L $R/structure_knowledge_distillation_amd -lskd_hip -L /opt/rocm/lib -lhipblaslt \
  -Wl,-rpath,'$ORIGIN/../structure_knowledge_distillation_amd' -Wl,-rpath,/opt/rocm/lib
echo built $R/tools/gemm_lab
