#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a ONE-GPU box: 2 ranks share cuda:0, gradients averaged over gloo
# (RN_REHEARSE_ONE_GPU, see retinanet_mi355x/ddp.py).  Checks that the real engine backward -> GradReducer -> fused
# optimizer chain runs and prints one JSON line with n_gpus = 2.  Throughput of this run means nothing.
#   bash tools/rehearse_ddp.sh
set -e
cd "$GRAFT_REPO_ROOT"
export RN_REHEARSE_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 2 --warmup 1 --batch 2 --no-kernel-timing
