#!/bin/bash
# rocprofv3 evidence of one round, run ON the GPU box from the repo root (gpurun -- 'bash tools/collect_evidence.sh r02'):
# kernel stats of the roofline leg, the three PMC passes over it (FETCH_SIZE, WRITE_SIZE, MfmaUtil: one counter
# set per pass, never together with a trace domain), the kernel trace of the training leg and the kernel stats of
# the default bench command.  Everything lands under gpurun_out/<tag>_*; tools/rocpd_export.py, tools/insitu.py,
# tools/pmc_traffic.py and tools/pmc_summary.py turn it into the files kept under profiles/.
set -o pipefail
tag=${1:-r03}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
run() { echo "[evidence] $*" >&2; "$@"; }
# which sources the profiled library was built from: the converters stamp it into profiles/<tag>_*.json and bench.py
# reports a committed profile as a measurement of the running build only when the hashes agree
python3 -c "from tiny_diffusion_amd import _build; print(_build.source_hash())" > gpurun_out/${tag}_src_sha256.txt
run rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_roof -- python3 bench.py --roofline-only > gpurun_out/${tag}_roof.json 2> gpurun_out/${tag}_roof.err && echo roof ok &&
run rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_f -- python3 bench.py --roofline-only > /dev/null 2> gpurun_out/${tag}_pmc_f.err && echo pmc_f ok &&
run rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_w -- python3 bench.py --roofline-only > /dev/null 2> gpurun_out/${tag}_pmc_w.err && echo pmc_w ok &&
run rocprofv3 --pmc MfmaUtil --output-format csv -d gpurun_out/${tag}_pmc_mfma -- python3 bench.py --roofline-only > /dev/null 2> gpurun_out/${tag}_pmc_mfma.err && echo pmc_mfma ok &&
run rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_train -- python3 bench.py --train-only --steps 30 --warmup 5 > gpurun_out/${tag}_train.json 2> gpurun_out/${tag}_train.err && echo train ok &&
run rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_bench -- python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_bench.err && echo bench ok
rc=$?
# the bf16 mode: kernel stats + one step's timeline of the MNIST leg, every GEMM launch in isolation, L2 counters
run rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_bf16 -- python3 tools/gpu_bf16_bench.py mnist 25 > gpurun_out/${tag}_bf16.log 2>&1 && echo bf16 ok
# conversions into the files kept under profiles/ (small; done here so that they travel back with gpurun_out/)
sha=gpurun_out/${tag}_src_sha256.txt
out=gpurun_out/${tag}_profiles
mkdir -p $out
python3 tools/pmc_traffic.py gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w $out/${tag}_pmc_hbm_traffic $sha > /dev/null
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_mfma MfmaUtil $out/${tag}_pmc_MfmaUtil.txt > /dev/null
trace=$(ls gpurun_out/${tag}_train/*/*kernel_trace.csv | head -n 1)
python3 tools/insitu.py $trace $out/${tag}_insitu.json $sha > /dev/null
python3 tools/step_timeline.py $trace 10 > $out/${tag}_step_timeline.txt
cp $(ls gpurun_out/${tag}_roof/*/*kernel_stats.csv | head -n 1) $out/${tag}_roofline_leg_kernel_stats.csv
cp $(ls gpurun_out/${tag}_train/*/*kernel_stats.csv | head -n 1) $out/${tag}_train_only_kernel_stats.csv
cp $(ls gpurun_out/${tag}_bench/*/*kernel_stats.csv | head -n 1) $out/${tag}_bench_default_kernel_stats.csv
cp gpurun_out/${tag}_roof.json $out/${tag}_roofline_leg.json
cp gpurun_out/${tag}_bench_under_rocprof.json $out/${tag}_bench_default_under_rocprof.json
cp $sha $out/
bf=$(ls gpurun_out/${tag}_bf16/*/*kernel_trace.csv 2>/dev/null | head -n 1)
if [ -n "$bf" ]; then
  python3 tools/step_timeline.py $bf 10 > $out/${tag}_bf16_step_timeline.txt
  cp $(ls gpurun_out/${tag}_bf16/*/*kernel_stats.csv | head -n 1) $out/${tag}_bf16_mnist_kernel_stats.csv
fi
rm -rf gpurun_out/${tag}_bf16
python3 tools/gpu_bf16_layers.py mnist > $out/${tag}_bf16_layers_mnist.txt 2>/dev/null
python3 tools/gpu_bf16_layers.py laion64 > $out/${tag}_bf16_layers_laion64.txt 2>/dev/null
bash tools/bf16_layer_counters.sh mnist > /dev/null 2>&1 && cp gpurun_out/bf16_layer_counters_mnist.txt $out/${tag}_bf16_layer_counters_mnist.txt
rm -rf gpurun_out/${tag}_roof gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w gpurun_out/${tag}_pmc_mfma gpurun_out/${tag}_train gpurun_out/${tag}_bench
# the per-dispatch traces of the long legs are large (two 1000-step chains): keep the stats only
find gpurun_out/${tag}_bench gpurun_out/${tag}_roof -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh gpurun_out/${tag}_* 2>/dev/null
exit $rc
