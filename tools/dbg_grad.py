import sys, os
sys.path[:0]=['3d-playground_amd','tests','.']
import numpy as np, torch
import golden_cases as gc
from retinanet_mi355x import modules
z=np.load('tests/golden/model.npz')
dev=torch.device('cuda:0')
for arch in ('resnet18','resnet50'):
    sd,img,ann=gc.model_inputs(arch,True)
    net=getattr(modules,arch)(num_classes=4); net.load_state_dict(sd); net=net.to(dev); net.train(); net.freeze_bn()
    l=net([img.to(dev),ann.to(dev)]); sum(l).sum().backward()
    rows=[]
    for name,p in net.named_parameters():
        full="%s_dir_g_%s"%(arch,name)
        g=p.grad.cpu().numpy().astype(np.float64)
        nr=z["%s_dir_gsum_%s"%(arch,name)][2]
        nerr=abs(np.sqrt((g**2).sum())-nr)/nr
        if full in z.files:
            ref=z[full]; e=np.abs(g-ref).max()/(np.abs(ref).max()+1e-12)
        else: e=float('nan')
        rows.append((e if e==e else -1,nerr,name))
    rows.sort(reverse=True)
    print(arch)
    for r in rows[:12]: print("  %.2e normerr %.2e %s"%r)
