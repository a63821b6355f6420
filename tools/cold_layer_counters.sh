#!/bin/bash
# TCC (L2) and fabric counters of the 128 -> 128 @28x28 forward convolution, first launches after an eviction
# against the 40th and later (the layer that loses 14-17 % when its input is cold: DESIGN.md 6.3).  Two PMC passes
# (4 TCC slots each), no trace domain beside them.  Run ON the GPU box from the repo root:
#   bash tools/cold_layer_counters.sh r03   ->  gpurun_out/r03_cold_layer_counters.txt
tag=${1:-r03}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d gpurun_out/${tag}_cold_a -- python3 tools/gpu_cold_layer.py 64 > gpurun_out/${tag}_cold_a.log 2>&1 &&
rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_cold_b -- python3 tools/gpu_cold_layer.py 64 > gpurun_out/${tag}_cold_b.log 2>&1 &&
python3 tools/cold_layer_summary.py gpurun_out/${tag}_cold_a gpurun_out/${tag}_cold_b gpurun_out/${tag}_cold_a.log > gpurun_out/${tag}_cold_layer_counters.txt
rc=$?
rm -rf gpurun_out/${tag}_cold_a gpurun_out/${tag}_cold_b
cat gpurun_out/${tag}_cold_layer_counters.txt
exit $rc
