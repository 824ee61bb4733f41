cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/p8
RN_BF16_P8=2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p8/prof -- python3 tools/bench_conv_bf16.py --no-fp32 --only "head tower" > gpurun_out/p8/prof.log 2>&1
cp "$(ls gpurun_out/p8/prof/*/*kernel_stats.csv | head -1)" gpurun_out/p8/kernel_stats_head_tower.csv
timeout -k 10 100 tools/probes/gemm8_probe 259200 256 2304 1 256 | tail -2
