#!/bin/bash
# Per kernel of one HIP source: the order of global/buffer loads (L), vmcnt waits (w<n>), branches (|) and loop back-edges,
# read off the gfx950 ISA -- the quick way to see a load that the compiler sank into a branch with a full wait behind it
# ("| L w0 | L w0 ..." = one exposed memory round trip per load; "L L L L w3 w2 .." = all in flight together).
# usage: tools/isa_loads.sh opental_amd/csrc/pool3d.hip [kernel-name-regex] [extra -D flags]
src=$1; pat=${2:-.}; shift; shift
out=/tmp/isa_$(basename $src .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Iopental_amd/csrc "$@" -S --cuda-device-only $src -o $out 2>/dev/null || exit 2
awk '/^_Z.*:.*; @/{name=$1} /global_load|buffer_load/{l=l" L"} /s_waitcnt.*vmcnt/{match($0,/vmcnt\([0-9]+\)/); l=l" w"substr($0,RSTART+6,RLENGTH-7)} /s_cbranch/{l=l" |"} /s_endpgm/{print name, l; l=""}' $out | c++filt | grep -E "$pat" | sed -e 's/(anonymous namespace):://g' | cut -c1-400
