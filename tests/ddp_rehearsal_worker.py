"""Worker of tests/test_gpu_ddp_rehearsal.py (not a test module): one rank of a 2-rank run on ONE GPU over gloo
(RN_REHEARSE_ONE_GPU).  Every rank trains one step on its own images through the real engine with the gradient
reducer attached; rank 0 then recomputes both ranks' gradients locally, without the reducer, and checks that what the
reducer handed to autograd is their mean."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "3d-playground_amd"))
sys.path.insert(0, HERE)
import golden_cases as gc                                  # noqa: E402
from retinanet_mi355x import ddp, modules, synth           # noqa: E402


def grads_of(net, img, ann):
    for p in net.parameters():
        p.grad = None
    cls_l, reg_l, vp_l = net([img, ann])
    (cls_l.mean() + reg_l.mean() + vp_l.mean()).backward()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters()}


def main():
    rank, local, world = ddp.init_from_env()
    assert world == 2 and local == 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    sd, _, _ = gc.model_inputs("resnet18", True)
    net = modules.resnet18(num_classes=4)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    net.freeze_bn()
    H, W = gc.MODEL_HW

    def data(r):
        img = synth.frames(2, H, W, seed=100 + r).to(dev)
        ann = synth.labels_dir(2, 5, H, W, num_classes=4, seed=200 + r, size_px=(24, 60)).to(dev)
        return img, ann
    net.set_gradient_reducer(ddp.GradReducer(bucket_bytes=4 << 20))        # several buckets on this small model
    reduced = grads_of(net, *data(rank))
    dist.barrier()
    ok, worst = True, 0.0
    if rank == 0:
        net.set_gradient_reducer(None)
        g0, g1 = grads_of(net, *data(0)), grads_of(net, *data(1))
        for n in g0:
            want = 0.5 * (g0[n] + g1[n])
            err = float((reduced[n] - want).norm() / (want.norm() + 1e-12))
            worst = max(worst, err)
            ok = ok and err <= 2e-3                                        # fp32 atomics in wgrad: order-dependent last bits
        print(json.dumps({"ok": ok, "worst_rel_l2": worst, "params": len(g0)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
