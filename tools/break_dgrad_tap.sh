#!/bin/bash
# Shows that the bf16 backward pins (tests/test_bf16_parity_gpu.py::test_bf16_backward_matches_the_operand_rounding_oracle_block_by_block,
# end to end; tests/test_bf16_layer_pin_gpu.py, every block alone) catch a mis-routed gradient: builds a SECOND library in /tmp whose direct 3x3x3 data-gradient weight pack does not flip
# the temporal taps (-DOTAL_BREAK_DGRAD_TAP: gradient norms barely move, directions do) and runs the test against it through
# OTAL_LIB_PATH -- it must FAIL; the product library in opental_amd/lib is not touched.   usage (GPU box): tools/break_dgrad_tap.sh
set -u
repo=$(pwd)
out=/tmp/otal_broken; mkdir -p $out/obj
pids=()
for f in opental_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Iopental_amd/csrc -DOTAL_BREAK_DGRAD_TAP -c $f -o $out/obj/$b.o 2> $out/$b.log &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "build failed"; exit 2; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libopental_hip.so $out/obj/*.o || exit 2
OTAL_LIB_PATH=$out/libopental_hip.so python -m pytest tests/test_bf16_parity_gpu.py -q -s -k backward 2>&1 | grep -E "bf16 backward parity|AssertionError|passed|failed" | cut -c1-1500
echo "(expected: 1 failed)"
# ... and the layer-by-layer pin names the first broken layer (Conv3d_2c: the first 3x3x3 data gradient)
OTAL_LIB_PATH=$out/libopental_hip.so python -m pytest tests/test_bf16_layer_pin_gpu.py -q -s 2>&1 | grep -E "^E +AssertionError|passed|failed" | cut -c1-400
echo "(expected: failures from Conv3d_2c on; Conv3d_1a / 2b and the 1x1-only gradients pass)"
