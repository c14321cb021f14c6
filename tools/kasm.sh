#!/bin/bash
# tools/kasm.sh <file.hip> [extra flags]: device-only assembly of one translation unit into /tmp/<name>.s and
# the register / spill summary of every kernel in it (CPU only; hipcc cross-compiles gfx950).
f=$1; shift
n=$(basename $f .hip)
cd /root/repo/deepvariant_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -I. --cuda-device-only -S $n.hip -o /tmp/$n.s "$@" 2>&1 | grep -E "error" 
python3 - /tmp/$n.s <<'P'
import re,sys
t=open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', t):
    print('%-90s scratch %4s  sgpr %3s (spill %s)  vgpr %3s (spill %s)' % (m.group(1)[:90], m.group(2), m.group(3), m.group(4), m.group(5), m.group(6)))
P
