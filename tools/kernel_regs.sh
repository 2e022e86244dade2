#!/bin/bash
# tools/kernel_regs.sh [-DFLAG ...] : compiles the GPU library with line tables + --save-temps into /tmp/isa and prints VGPRs / SGPRs / scratch /
# LDS / occupancy per kernel (what decides the waves per SIMD); the .s and .out there feed tools/isa_lines.py / isa_inline.py
mkdir -p /tmp/isa && cd /tmp/isa && /opt/rocm/bin/hipcc $(python3 /root/repo/tools/build_id.py --flags) -gline-tables-only --save-temps "$@" -o /tmp/isa/lib.so /root/repo/misaki-render_amd/csrc/msk_gpu.hip 2>&1 | grep -v "MD5\|\.file\|\^" | grep -i "error\|warning" 
awk '/^_ZN3msk[^ ]*:/ {name=$1} /^; NumVgprs:/ {v=$3} /^; NumAgprs:/ {a=$3} /^; ScratchSize:/ {s=$3} /^; LDSByteSize:/ {l=$3} /^; Occupancy:/ {printf "%-100s vgpr %3s agpr %3s scratch %4s lds %6s occ %s\n", substr(name,1,100), v, a, s, l, $3}' /tmp/isa/msk_gpu-hip-amdgcn-amd-amdhsa-gfx950.s
