#!/bin/bash
# resource usage (VGPRs, scratch, LDS, occupancy) of the kernels of one translation unit whose name matches $2
cd /root/repo/norlab_icp_mapper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage $3 -c $1 -o /tmp/kres.o 2>&1 | grep -A9 "Function Name: .*$2" | grep -v "^--" | sed 's/.*remark: [^ ]* *//' | grep -v "AGPRs\|Dynamic\|SGPRs:" 
