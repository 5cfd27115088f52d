#!/bin/bash
# Everything a round commits under profiles/ from ONE box (run through gpurun): kernel statistics on one stream, HBM traffic
# (separate --pmc passes), MFMA utilisation, the per-layer convolution table, the lane-graph step's kernel timeline and the
# default bench line.   usage: tools/evidence_round.sh r03b   -> gpurun_out/<tag>_*
tag=${1:-rXX}
bash tools/profile_round.sh $tag
bash tools/pmc_mfma.sh gpurun_out/${tag}_pmc_mfma_util.txt > /dev/null 2>&1
OTAL_TOP=400 python tools/profile_convs.py 8 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${tag}_conv_layers_b8_bf16.txt
bash tools/trace_lanes.sh ${tag}x lanes > /dev/null 2>&1
mv gpurun_out/${tag}x_step_timeline.txt gpurun_out/${tag}_step_timeline_lanes.txt
ls -la gpurun_out/${tag}_*
