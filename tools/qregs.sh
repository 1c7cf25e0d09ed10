#!/bin/bash
# register / scratch / LDS use of the quad kernel instantiations (variant H)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S ${1:-/root/repo/bwa-mem-sw_amd/csrc/bsw_quad_kernel.hip} -o /tmp/t/quad.s 2>&1 | grep -E "error" -A5 | head -20
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|name):" /tmp/t/quad.s | paste - - - - - - | awk '{print "lds",$2,"scratch",$6,"sgpr",$8,"vgpr",$10,"spill",$12, substr($4,1,40)}' | grep "ELi0E"
