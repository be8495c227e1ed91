#!/bin/bash
# Registers, spills, scratch and LDS of every kernel of one source file of the library, as the compiler reports them in the code
# object's metadata (no GPU needed):   tools/kernel_meta.sh point_in_tet.hip [-DNAME=V ...] [| grep k_tet_scan_wave]
# The flags are the build's own (deftet_amd/build.py: FLAGS + the per-file ones).
set -e
cd "$(dirname "$0")/.."
src=$1; shift
flags=$(python - "$src" <<'PY'
import sys
from deftet_amd import build
print(" ".join(f for f in build.FLAGS + build._file_flags(sys.argv[1]) if f not in ("-fPIC", "-Wall", "-Wno-unused-function")))
PY
)
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc $flags "$@" -I deftet_amd/csrc -x hip --cuda-device-only -S deftet_amd/csrc/$src -o $tmp/k.s 2>/dev/null
grep -E "^\s+\.name:|\.vgpr_count|\.vgpr_spill_count|\.group_segment_fixed_size|\.private_segment_fixed_size" $tmp/k.s | paste - - - - - |
  sed -E 's/\s+/ /g' | awk '{for(i=1;i<=NF;i++){if($i==".name:")n=$(i+1);if($i==".vgpr_count:")v=$(i+1);if($i==".vgpr_spill_count:")s=$(i+1);if($i==".group_segment_fixed_size:")l=$(i+1);if($i==".private_segment_fixed_size:")p=$(i+1)} print n, "vgpr", v, "spilled", s, "scratch_bytes", p, "lds_bytes", l}' |
  while read n rest; do echo "$(echo $n | c++filt | sed -E 's/\(.*//; s/^void //') $rest"; done
rm -rf $tmp
