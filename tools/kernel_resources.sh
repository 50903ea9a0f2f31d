#!/bin/bash
# Per-kernel register / LDS / scratch use: compiles the device code of the kernel translation units to assembly (the flags of csrc/Makefile) and
# prints the code-object metadata.  usage: kernel_resources.sh [TU ...] [-- extra flags]   (default: every kernels_*.hip; the assembly stays in
# /tmp/wtgpu_dev_<TU>.s and, concatenated, in /tmp/wtgpu_dev.s for tools/asm_blocks.py / kernel_isa_stats.py)
cd $(dirname $0)/../wave_tracer_amd/csrc
TUS=(); FLAGS=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; FLAGS=("$@"); break; fi
  TUS+=("$1"); shift
done
[ ${#TUS[@]} -eq 0 ] && TUS=(kernels_trace kernels_walk kernels_fsd kernels_path kernels_connect)
for T in "${TUS[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=off -S --cuda-device-only "${FLAGS[@]}" -o /tmp/wtgpu_dev_$T.s $T.hip 2>/dev/null &
done
wait
: > /tmp/wtgpu_dev.s
for T in "${TUS[@]}"; do cat /tmp/wtgpu_dev_$T.s >> /tmp/wtgpu_dev.s; done
python3 - <<'PY'
import re
t=open('/tmp/wtgpu_dev.s').read()
for part in t.split('amdhsa.kernels:')[1:]:
  for blk in part.split('  - .agpr_count:')[1:]:
    g=lambda k: (re.search(r'\.%s:\s+(\S+)'%k,blk) or [None,'?'])[1]
    name=re.search(r'\d+(k_\w+?)E',g('name'))
    print('%-24s vgpr %3s spill %3s sgpr_spill %3s lds %6s scratch %5s'%(name.group(1) if name else g('name')[:24],g('vgpr_count'),g('vgpr_spill_count'),g('sgpr_spill_count'),g('group_segment_fixed_size'),g('private_segment_fixed_size')))
PY
