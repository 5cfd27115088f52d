#!/bin/bash
# Builds libopental_hip.so of ANOTHER commit (sources exported to /tmp) into ab/<name>.so -- git-ignored, but it travels to
# the GPU box -- so that two kernel versions can be timed on ONE box: boxes differ by ~3 %, more than most single changes.
#   tools/build_lib_at.sh <commit> <name>   ;   then on the box:  tools/ab_lanes.sh "OTAL_LIB_PATH=ab/<name>.so" "-" ...
# (only the library is swapped: the Python of the working tree must still be able to drive the old entry points)
set -e
commit=$1; name=$2
repo=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/otal_at_$name; rm -rf $d; mkdir -p $d/obj $repo/ab
git -C $repo archive $commit opental_amd/csrc include | tar -x -C $d
pids=()
for f in $d/opental_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$d/include -I$d/opental_amd/csrc -c $f -o $d/obj/$b.o 2> $d/$b.log &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "build failed (see $d/*.log)"; exit 2; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $repo/ab/$name.so $d/obj/*.o
echo $repo/ab/$name.so
