#!/bin/bash
# Developer tool: list the s_waitcnt vmcnt instructions the COMPILER inserted into a kernel (those outside inline asm) and
# the kernel's scratch accesses, each with the loop it sits in.  hipcc cannot see LDS-DMA copies issued from inline asm, so
# every one of these waits drains the copy ring: none may sit inside a hot loop.
# usage: tools/isa_waits.sh <file.hip> <substring of the mangled kernel name> [extra hipcc flags]
#   e.g. tools/isa_waits.sh vit_colmap_amd/csrc/matcher.hip pair2_kernelILi12
set -e
src="$1"; k="$2"; shift; shift
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -I"$(dirname "$src")" "$@" -S --cuda-device-only -o "$tmp/k.s" "$src"
L=$(grep -n "^_ZN.*$k.*:" "$tmp/k.s" | head -1 | cut -d: -f1)
[ -z "$L" ] && { echo "no kernel matching $k"; exit 1; }
awk -v l="$L" 'NR>=l{print} NR>l && /^\.Lfunc_end/{exit}' "$tmp/k.s" > "$tmp/kk.s"
echo "== $k: $(wc -l < "$tmp/kk.s") lines, $(grep -c scratch_ "$tmp/kk.s" || true) scratch accesses, $(grep -A80 "amdhsa_kernel.*$k" "$tmp/k.s" | grep -m1 -o "NumVgprs: [0-9]*") $(grep -A80 "amdhsa_kernel.*$k" "$tmp/k.s" | grep -m1 -o "ScratchSize: [0-9]*")"
awk '/^\.LBB|^; %bb/ {blk=$0} /ASMSTART/ {asm=1} /ASMEND/ {asm=0}
     (/vmcnt/ && !asm) || /scratch_/ {print NR": "$1" "$2" "$3" | "blk}' "$tmp/kk.s" | sed 's/  */ /g; s/\t/ /g' | cut -c1-150
rm -rf "$tmp"
