#!/bin/bash
# L2 (TCC) hit / miss and fabric-read counters of every bf16-mode GEMM launch of tools/gpu_bf16_layers.py, per kernel
# name: where do the re-reads of the nine taps come from - the XCD's L2, or beyond it?  One PMC pass, no trace domain.
# Run ON the GPU box from the repo root:  bash tools/bf16_layer_counters.sh [mnist|laion64]  ->  gpurun_out/bf16_layer_counters_<which>.txt
which=${1:-mnist}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/bf16_ctr_$which -- python3 tools/gpu_bf16_layers.py $which > gpurun_out/bf16_ctr_$which.log 2>&1 || exit 1
python3 - "$which" <<'PY' > gpurun_out/bf16_layer_counters_$which.txt
import csv, glob, sys, collections
which = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/bf16_ctr_{which}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        rows[k]["_n"] += 0.25
print(f"{'kernel':72s} launches   hit_rate   L2_req(MB@128B)   fabric_read(MB)")
for k, c in sorted(rows.items()):
    hit, miss = c["TCC_HIT_sum"], c["TCC_MISS_sum"]
    rd = c["TCC_EA0_RDREQ_sum"]; rd32 = c["TCC_EA0_RDREQ_32B_sum"]
    fabric = (rd - rd32) * 64 + rd32 * 32
    if hit + miss == 0: continue
    print(f"{k:72s} {int(c['_n']):6d}   {hit / (hit + miss):8.3f}   {(hit + miss) * 128 / 1e6:12.1f}   {fabric / 1e6:12.1f}")
PY
rm -rf gpurun_out/bf16_ctr_$which
cat gpurun_out/bf16_layer_counters_$which.txt
