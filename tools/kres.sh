#!/bin/bash
# register / LDS / spill figures of the device kernels (no GPU needed):   bash tools/kres.sh [pattern]     e.g. k_nonbond
# compiles ddcmi.hip device-only for gfx950 and reads the kernel descriptors' metadata
set -e
cd "$(cd "$(dirname "$0")/.." && pwd)"
out=/tmp/ddcmi_kres_$$; mkdir -p $out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -Iinclude -Iddcmd_amd/csrc/hip --cuda-device-only -c ${KRES_SRC:-ddcmd_amd/csrc/hip/ddcmi.hip} -o $out/dev.co ${KRES_FLAGS:-} 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$out/dev.co --output=$out/dev.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out/dev.elf | python3 -c "
import sys,re
pat=sys.argv[1] if len(sys.argv)>1 else ''
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)', blk) or [None,'?'])[1]
    n=g('name')
    if pat in n:
        import subprocess
        d=subprocess.run(['c++filt', n],capture_output=True,text=True).stdout.strip()
        print('%-90s vgpr %s spill %s sgpr %s lds %s scratch %s' % (d[:90], g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
" "${1:-}"
rm -rf $out
