#!/bin/bash
# usage: tools/kernel_resources.sh <object.o> [name filter]  -- scratch bytes / VGPRs / AGPRs / LDS of every gfx950 kernel in a hipcc object
set -e
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
$LLVM/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.o
$LLVM/llvm-readelf --notes $T/dev.o > $T/notes.txt
python3 - "$T/notes.txt" "${2:-}" <<'PY'
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2]
for blk in t.split('- .agpr_count:')[1:]:
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk)
    name = g('name').group(1)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dn:
        continue
    ag = re.match(r'\s*(\d+)', blk).group(1)
    print("scratch %5s vgpr %3s agpr %3s lds %6s  %s" % (g('private_segment_fixed_size').group(1), g('vgpr_count').group(1), ag,
                                                         g('group_segment_fixed_size').group(1), dn[:120]))
PY
rm -rf $T
