#!/bin/bash
# Dev probe (build container): register / LDS / scratch use of every kernel in a csrc unit.
#   tools/kernel_resources.sh mm_kernels.hip [filter-regex]
#   UNIT=4 tools/kernel_resources.sh mm_filter_shapes.hip "mm_filter_u8<"      (one of the streaming-kernel units, build.py)
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imonkey-moore_amd/csrc ${UNIT:+-DMM_FILTER_SHAPE_UNIT=$UNIT} -x hip -c monkey-moore_amd/csrc/$1 -o /tmp/kr_$$.o \
   -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m: cur = {'name': m.group(1)}; rows.append(cur); continue
    for key in ('SGPRs', 'VGPRs', 'AGPRs', 'ScratchSize \[bytes/lane\]', 'Occupancy \[waves/SIMD\]', 'LDS Size \[bytes/block\]'):
        m = re.search(key + r': (\d+)', line)
        if m and cur is not None: cur[key.split(' ')[0]] = int(m.group(1))
import subprocess
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip().split('(')[0]
    if pat and not pat.search(name): continue
    print('%-44s VGPR %3d SGPR %3d scratch %4d occ %2d LDS %6d' % (name[:44], r.get('VGPRs', -1), r.get('SGPRs', -1), r.get('ScratchSize', -1), r.get('Occupancy', -1), r.get('LDS', -1)))
" "${2:-}"
rm -f /tmp/kr_$$.o
