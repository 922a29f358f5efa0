#!/bin/bash
# kernel resource usage of one csrc file: tools/kres.sh abn.hip [name-filter]
cd /root/repo/structure_knowledge_distillation_amd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -I ../include -I csrc -c csrc/$1 -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "error|Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' \
 | awk '/Function Name/{if(l)print l; l=$0; next}{l=l" | "$0}END{print l}' | grep -E "${2:-.}" | sed -e 's/Function Name: //' | c++filt | cut -c1-400
