#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a ONE-GPU box: `python bench.py --gpus 2` starts its own two ranks
# (bench.launch_ranks: a torch.distributed.run child), which share cuda:0 and average gradients over gloo
# (RN_REHEARSE_ONE_GPU, see retinanet_mi355x/ddp.py).  Checks that the real engine backward -> flat gradient buffer ->
# in-place bucketed all-reduce -> fused optimizer chain runs and prints one JSON line with n_gpus = 2, then prints the
# per-bucket overlap timeline of one step (bucket ready / all-reduce issued / done vs. the end of backward).
# Throughput of this run means nothing.
#   bash tools/rehearse_ddp.sh
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export RN_REHEARSE_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 2 --no-kernel-timing --ddp-timeline
