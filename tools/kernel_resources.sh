#!/bin/bash
# Per-kernel register / LDS / scratch use: compiles the device code to assembly and prints the code-object metadata.
cd $(dirname $0)/../wave_tracer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -S --cuda-device-only "$@" -o /tmp/wtgpu_dev.s wtgpu.hip 2>/dev/null
python3 - <<'PY'
import re
t=open('/tmp/wtgpu_dev.s').read()
i=t.index('amdhsa.kernels:')
for blk in t[i:].split('  - .agpr_count:')[1:]:
    g=lambda k: (re.search(r'\.%s:\s+(\S+)'%k,blk) or [None,'?'])[1]
    name=re.search(r'\d+(k_\w+?)E',g('name'))
    print('%-18s vgpr %3s spill %3s sgpr_spill %3s lds %6s scratch %5s'%(name.group(1) if name else g('name')[:18],g('vgpr_count'),g('vgpr_spill_count'),g('sgpr_spill_count'),g('group_segment_fixed_size'),g('private_segment_fixed_size')))
PY
