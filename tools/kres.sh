#!/bin/bash
# registers / LDS / spills of the level kernels: tools/kres.sh [filter-regex] [extra hipcc flags, e.g. -DPF_EXPERIMENTS=1]
# (compiles kernels.hip to /tmp/kb with -save-temps)
set -e
flt=$1; shift || true
src=${PF_KRES_SRC:-kernels}
cd "$(dirname "$0")/../pi-slam-fusion_amd/csrc"
mkdir -p /tmp/kb
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 $PF_EXTRA_FLAGS "$@" -c -save-temps=obj -o /tmp/kb/$src.o -x hip $src.hip 2>&1 | grep -E "error|warning: [^s]" || true
python3 - "$flt" "$src" <<'PY'
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else 'k_levels'
s = open('/tmp/kb/%s-hip-amdgcn-amd-amdhsa-gfx950.s' % sys.argv[2]).read()
md = s[s.find('amdhsa.kernels'):]
for b in md.split('  - .agpr_count')[1:]:
    n = re.search(r'\.name:\s+(\S+)', b).group(1)
    if not re.search(flt, n): continue
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
    print(n[:70], 'vgpr', g('vgpr_count'), 'sgpr', g('sgpr_count'), 'lds', g('group_segment_fixed_size'), 'vspill', g('vgpr_spill_count'),
          'sspill', g('sgpr_spill_count'), 'scratch', g('private_segment_fixed_size'))
PY
