#!/bin/bash
# Register / scratch / LDS usage of every kernel in libgraphik_amd.so (code-object metadata):
#   tools/isa_report.sh [lib]  ->  name, vgpr, agpr, sgpr, scratch bytes, static LDS bytes
set -e
LIB=$(readlink -f "${1:-$(dirname "$0")/../graphik_amd/lib/libgraphik_amd.so}")
TMP=$(mktemp -d)
cp "$LIB" "$TMP/lib.so"
cd "$TMP"
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null
# (one code object per translation unit: gik_k_*.hip)
for CO in $(ls lib.so.*gfx950*); do
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$CO" | python3 -c '
import sys, yaml
txt = sys.stdin.read()
if "amdhsa.kernels" not in txt: sys.exit(0)
y = txt[txt.index("amdhsa.kernels"):].split("\n...")[0]
for k in yaml.safe_load(y)["amdhsa.kernels"]:
    print("%-64s vgpr %3d agpr %3d sgpr %3d scratch %5d lds %6d" % (
        k[".name"], k[".vgpr_count"], k.get(".agpr_count", 0), k[".sgpr_count"],
        k[".private_segment_fixed_size"], k[".group_segment_fixed_size"]))
' | c++filt -_ 2>/dev/null || true
done
rm -rf "$TMP"
