import os, sys
sys.path.insert(0, "/root/repo/3d-playground_amd")
import torch
from torch.profiler import profile, ProfilerActivity
from retinanet_mi355x import modules, optim, synth
dev = torch.device("cuda:0")
B, H, W = 8, 1080, 1920
net = modules.resnet50(num_classes=8); net.load_state_dict(synth.state_dict("resnet50", 8, 12, seed=2)); net = net.to(dev)
net.train(); net.freeze_bn(); net.use_flat_gradients()
opt = optim.ClipAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)
img = torch.randn(B, 3, H, W, device=dev); ann = synth.labels_dir(B, 10, H, W, 8, seed=1).to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = sum(l.mean() for l in net([img, ann])); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
import collections
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::contiguous", "aten::_to_copy"):
        st = [s for s in (e.stack or []) if "retinanet_mi355x" in s or "bench" in s or "modules" in s]
        c[(e.name, st[0] if st else "?", str(e.input_shapes)[:60])] += 1
for k, v in c.most_common(40):
    print(v, k)
