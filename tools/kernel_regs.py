#!/usr/bin/env python3
"""Static summary of every kernel of an assembly listing (hipcc --cuda-device-only -S): instructions, VGPRs, SGPRs, LDS, scratch.
usage: hipcc --offload-arch=gfx950 <COMMON flags> --cuda-device-only -S kernels.hip -o /tmp/kernels.s; python tools/kernel_regs.py /tmp/kernels.s [filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for n in re.findall(r'^(_ZN4mipt\w+):', s, re.M):
    start = s.index('\n' + n + ':'); end = s.index('.end_amdhsa_kernel', start)
    body = s[start:end]
    g = lambda k: re.search(re.escape(k) + r'\s+(\S+)', body).group(1)      # noqa: E731
    ninstr = len([ln for ln in body.split('\n') if ln.startswith('\t') and not ln.startswith('\t.') and not ln.startswith('\t;')])
    dem = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()[:100]
    if flt in dem:
        print('%5d instr vgpr %3s sgpr %3s lds %5s scratch %3s  %s' % (ninstr, g('.amdhsa_next_free_vgpr'), g('.amdhsa_next_free_sgpr'),
              g('.amdhsa_group_segment_fixed_size'), g('.amdhsa_private_segment_fixed_size'), dem))
