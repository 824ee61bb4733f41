#!/bin/bash
# The measurement set behind DESIGN.md section 6 / profiles/README.md on the GPU box, in three calls (a gpurun call lasts 20 min at most):
#   bash tools/final_measure.sh 1|2|3   -> gpurun_out/final/*   (copy what is to be judged into profiles/)
# Steps are chained: a failing step stops the script (no GPU step is started after a failed one).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
mkdir -p "$O"
part=${1:-1}
if [ "$part" = 1 ]; then
  RN_TEST_MEASURE=0 timeout -k 10 900 python3 -m pytest tests -q -s -m gpu > "$O/gpu_tests.log" 2>&1
  echo "tests done" && tail -1 "$O/gpu_tests.log"
  timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > "$O/bench.log" 2>&1
  echo "bench done"
fi
if [ "$part" = 2 ]; then
  # the headline's own steps only, towers and weight gradients on ONE stream (a kernel's duration in the trace means something only when nothing else
  # shares the GPU): Sum(duration) / (warmup + 2 * steps) of a family = kernels.<family>.ms_per_step of the line it prints
  RN_TOWER_STREAMS=0 RN_WGRAD_STREAMS=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --steps 5 --warmup 2 --sections headline --no-cpu-baseline > "$O/bench_headline_under_rocprof.log" 2>&1
  cp "$(ls $O/stats/*/*kernel_stats.csv | head -1)" "$O/kernel_stats_headline.csv"
  echo "stats done"
  bash tools/collect_traffic.sh
  python3 tools/summarize_traffic.py gpurun_out/traffic > "$O/pmc_traffic.json"
  echo "traffic done"
  timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 --dtype bf16 --no-cpu-baseline > "$O/bench_bf16.log" 2>&1
  timeout -k 10 600 python3 bench.py --arch resnet101 --dtype fp8 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_fp8_train_resnet101_b16.log" 2>&1
  timeout -k 10 400 python3 tools/fp8_error_budget.py > "$O/fp8_error_budget.txt" 2>&1
  timeout -k 10 300 python3 tools/bench_loss.py > "$O/loss_microbench.txt" 2>&1
  echo "part 2 done"
fi
if [ "$part" = 3 ]; then
  timeout -k 10 300 python3 tools/bench_conv.py --mfma split3 > "$O/conv_microbench_split3.txt" 2>&1
  timeout -k 10 300 python3 tools/bench_conv.py --mfma split > "$O/conv_microbench_split.txt" 2>&1
  for d in normal relu wide widepix; do timeout -k 10 200 python3 tools/fp32_mode_errors.py --data $d; done > "$O/fp32_split3_errors.txt" 2>&1
  timeout -k 10 300 python3 tools/profile_layers.py --dtype fp32 > "$O/fp32_step_by_shape.txt" 2>&1
  timeout -k 10 300 python3 tools/profile_layers.py > "$O/bf16_step_by_shape.txt" 2>&1
  timeout -k 10 300 python3 tools/profile_fp8_layers.py > "$O/fp8_forward_by_shape.txt" 2>&1
  timeout -k 10 300 python3 tools/bench_conv_bf16.py --no-fp32 > "$O/conv_microbench_bf16.txt" 2>&1
  timeout -k 10 300 python3 tools/bench_conv_fp8.py > "$O/conv_microbench_fp8.txt" 2>&1
  timeout -k 10 300 python3 tools/bench_infer.py > "$O/infer_cfg4.txt" 2>&1
  timeout -k 10 300 python3 tools/bench_infer.py --sharded >> "$O/infer_cfg4.txt" 2>&1
  RN_REHEARSE_ONE_GPU=1 timeout -k 10 400 python3 tools/bench_infer.py --gpus 2 --iters 3 >> "$O/infer_cfg4.txt" 2>&1
  timeout -k 10 300 python3 tools/bench_infer.py --dtype bf16 >> "$O/infer_cfg4.txt" 2>&1
  timeout -k 10 300 python3 tools/bench_infer.py --dtype fp8 >> "$O/infer_cfg4.txt" 2>&1
  timeout -k 10 500 bash tools/rehearse_ddp.sh > "$O/ddp_rehearsal.txt" 2>&1
  RN_REHEARSE_ONE_GPU=1 timeout -k 10 500 python3 tools/train_ddp.py --gpus 2 --epochs 2 --iters 3 --batch 2 --out "$O/train_ddp_ck" > "$O/train_ddp_rehearsal.txt" 2>&1
  rm -rf "$O/train_ddp_ck"
  RN_DETERMINISTIC=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --sections headline --no-cpu-baseline --no-kernel-timing > "$O/bench_deterministic.log" 2>&1
  echo "part 3 done"
fi
