#!/bin/bash
# The measurement set behind DESIGN.md section 6 / profiles/README.md, in one call on the GPU box:
#   bash tools/final_measure.sh        -> gpurun_out/final/*   (copy what is to be judged into profiles/)
# Steps are chained: a failing step stops the script (no GPU step is started after a failed one).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf "$O" && mkdir -p "$O"
RN_TEST_MEASURE=0 timeout -k 10 900 python3 -m pytest tests -q -s -m gpu > "$O/gpu_tests.log" 2>&1
echo "tests done" && tail -1 "$O/gpu_tests.log"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > "$O/bench.log" 2>&1
echo "bench done"
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 --dtype bf16 --no-cpu-baseline > "$O/bench_bf16.log" 2>&1
echo "bench bf16 done"
# (the towers on ONE stream here: a kernel's duration in the trace means something only when nothing else shares the GPU)
RN_TOWER_STREAMS=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$O/bench_under_rocprof.log" 2>&1
cp "$(ls $O/stats/*/*kernel_stats.csv | head -1)" "$O/kernel_stats.csv"
echo "stats done"
bash tools/collect_traffic.sh
python3 tools/summarize_traffic.py gpurun_out/traffic > "$O/pmc_traffic.json"
echo "traffic done"
timeout -k 10 300 python3 tools/bench_loss.py > "$O/loss_microbench.txt" 2>&1
timeout -k 10 300 python3 tools/bench_conv_bf16.py > "$O/conv_bf16_microbench.txt" 2>&1
timeout -k 10 300 python3 tools/profile_layers.py > "$O/bf16_step_by_shape.txt" 2>&1
timeout -k 10 300 python3 tools/bench_conv.py --mfma native > "$O/conv_microbench_native.txt" 2>&1
timeout -k 10 300 python3 tools/bench_conv.py --mfma split > "$O/conv_microbench_split.txt" 2>&1
timeout -k 10 300 python3 tools/bench_infer.py > "$O/infer_cfg4.txt" 2>&1
timeout -k 10 300 python3 tools/bench_infer.py --dtype bf16 >> "$O/infer_cfg4.txt" 2>&1
RN_FP32_MFMA=native timeout -k 10 300 python3 tools/bench_infer.py >> "$O/infer_cfg4.txt" 2>&1
RN_DETERMINISTIC=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > "$O/bench_deterministic.log" 2>&1
timeout -k 10 300 python3 tools/profile_layers.py --dtype fp32 > "$O/fp32_step_by_shape.txt" 2>&1
timeout -k 10 500 bash tools/rehearse_ddp.sh > "$O/ddp_rehearsal.txt" 2>&1
bash tools/pmc_conv_split.sh > "$O/pmc_conv_split.txt" 2>&1
RN_TOWER_STREAMS=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > "$O/bench_one_stream.log" 2>&1
echo "all done"
