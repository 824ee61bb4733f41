#!/usr/bin/env python3
"""Runnable counterpart of the reference's trainer (train_detector_3D_angle.py:337-417) on synthetic data, one process per GPU.

  python tools/train_ddp.py --gpus N [--epochs 3] [--iters 4] [--batch 8] [--height 1080 --width 1920] [--arch resnet50]
                            [--dtype fp32|bf16] [--out gpurun_out/train_ddp] [--resume PATH]

  * model: the drop-in ``resnet50(num_classes=8)`` (directional 3D-RetinaNet), same random-init weights on every rank, heads
    re-initialised as the reference does (:290-291); ``--resume`` loads a checkpoint first -- also one written by the
    reference's 4-GPU DataParallel run (``module.``-prefixed keys);
  * data: synthetic frames + random 3D boxes, a different seed per (epoch, iteration, rank): rank r's shard;
  * step: forward + 3 losses, backward with the bucketed gradient all-reduce inside (ddp.GradReducer over RCCL),
    ``clip_grad_norm_(0.1)`` + Adam(1e-4) fused (optim.ClipAdam), ``ReduceLROnPlateau(patience 4, mode min)`` stepped once per
    epoch on the mean loss over ranks, ``torch.save(net.state_dict(), corrected_data_e{epoch}.pt)`` by rank 0 (retinanet_mi355x/trainer.py);
  * `python tools/train_ddp.py --gpus N` starts its own N ranks (a torch.distributed.run child, before this process touches
    the GPU); under torchrun it reads the rendezvous from the environment.  Rank 0 prints one JSON line at the end.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--iters", type=int, default=4, help="iterations per epoch")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--patience", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "train_ddp"))
    ap.add_argument("--resume", default=None, help="checkpoint (state_dict, with or without the DataParallel 'module.' prefix)")
    ap.add_argument("--dataparallel-keys", action="store_true",
                    help="write checkpoints with the 'module.' key prefix of the reference's multi-GPU trainer (train_detector_3D_angle.py:415-417)")
    return ap.parse_args()


def launch_ranks(args):
    have = torch.cuda.device_count()                       # does not initialise the GPU on this image
    if have < args.gpus and not os.environ.get("RN_REHEARSE_ONE_GPU"):
        raise SystemExit("train_ddp.py: --gpus %d but this node shows %d GPU(s)" % (args.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    from retinanet_mi355x import ddp, modules, optim, synth, trainer
    rank, local, world = ddp.init_from_env()
    if world != args.gpus:
        raise SystemExit("train_ddp.py: --gpus %d but the process group has %d rank(s)" % (args.gpus, world))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    B, H, W = args.batch, args.height, args.width
    net = getattr(modules, args.arch)(num_classes=8)
    net.load_state_dict(synth.state_dict(args.arch, 8, 12, seed=2))               # same weights on every rank
    if args.resume:
        net.load_state_dict(torch.load(args.resume, map_location="cpu", weights_only=True))
    net = net.to(dev)
    if args.dtype == "bf16":
        net.set_compute_dtype("bf16")
    reducer = None
    if world > 1:
        reducer = ddp.GradReducer(defer_scale=True)
        net.set_gradient_reducer(reducer)
    else:
        net.use_flat_gradients()
    opt = optim.ClipAdam([p for p in net.parameters() if p.requires_grad], lr=args.lr, max_norm=0.1,
                         grad_scale=reducer.grad_scale if reducer else 1.0)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=args.patience, mode="min")

    def batches(epoch):
        for it in range(args.iters):
            seed = 1000 * epoch + 10 * it + rank                                   # rank r's shard of iteration `it`
            yield synth.frames(B, H, W, seed=seed).to(dev), synth.labels_dir(B, 10, H, W, 8, seed=seed + 5).to(dev)

    t0 = time.time()
    hist = trainer.train(net, opt, sched, batches, args.epochs, checkpoint=os.path.join(args.out, "corrected_data_e{}.pt"),
                         rank=rank, log=lambda m: print(m, flush=True), dataparallel_keys=args.dataparallel_keys)
    torch.cuda.synchronize()
    dt = time.time() - t0
    # every rank holds the same weights: compare a checksum across ranks
    chk = torch.stack([p.detach().double().sum() for p in net.parameters()]).sum().reshape(1)
    same = True
    if world > 1:
        hi, lo = chk.clone(), chk.clone()                  # (all_reduce: also works over gloo with device tensors, the one-GPU rehearsal)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        same = bool(hi == lo)
    if rank == 0:
        print(json.dumps({"tool": "train_ddp", "arch": args.arch, "dtype": args.dtype, "ranks": world,
                          "backend": dist.get_backend() if world > 1 else None, "global_batch": world * B,
                          "epochs": hist, "seconds": round(dt, 2), "weights_identical_across_ranks": same,
                          "checkpoints": sorted(os.listdir(args.out)) if os.path.isdir(args.out) else []}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not same:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
