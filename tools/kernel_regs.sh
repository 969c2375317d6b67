#!/bin/bash
# Register / scratch / LDS use of every kernel in the built objects (eventclip_amd/csrc/*.o).
#   bash tools/kernel_regs.sh [name filter (regex on the demangled name)]
FILTER=${1:-.}
DIR=$(dirname "$0")/../eventclip_amd/csrc
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
LLVM=/opt/rocm/lib/llvm/bin
for obj in "$DIR"/*.o; do
  case "$obj" in *.diag.o) continue;; esac
  objcopy -O binary --only-section=.hip_fatbin "$obj" "$TMP/fat.bin" 2>/dev/null || continue
  [ -s "$TMP/fat.bin" ] || continue
  $LLVM/clang-offload-bundler --unbundle --type=o --input="$TMP/fat.bin" \
      --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/dev.o" 2>/dev/null || continue
  $LLVM/llvm-readelf --notes "$TMP/dev.o" | awk '
    /^ +\.agpr_count:/ {a=$2} /^ +\.group_segment_fixed_size:/ {l=$2} /^ +\.name:/ {n=$2}
    /^ +\.private_segment_fixed_size:/ {p=$2} /^ +\.sgpr_count:/ {s=$2} /^ +\.vgpr_count:/ {v=$2}
    /^ +\.wavefront_size:/ {printf "vgpr %-4s agpr %-4s sgpr %-4s scratch %-5s lds %-7s %s\n", v, a, s, p, l, n; a=0}'
done | while read -r line; do
  name=$(echo "${line##* }" | c++filt | sed 's/(anonymous namespace):://g' | cut -c1-120)
  echo "${line% *} $name"
done | grep -E "$FILTER"
