import os, sys, torch
sys.path.insert(0,'.')
from oracle import model_ref, synth
from tests.util import CFG, TAGS, load_cfg, synth_sd
from multiyolov5_amd.models.yolo import Model
tag = sys.argv[1] if len(sys.argv) > 1 else 's_base'
DEV='cuda:0'
cfg=load_cfg(tag); sd=synth_sd(tag)
x=synth.synth_images(2,64,128,seed=1)
params={k:v.clone().requires_grad_() for k,v in sd.items() if v.dtype.is_floating_point and 'running' not in k and 'anchor' not in k}
sdt={k:(params[k] if k in params else v.clone()) for k,v in sd.items()}
rdet,rseg=model_ref.forward(cfg,sdt,x,training=True,dropout_p=0.0)
gen=torch.Generator().manual_seed(5)
rd=[torch.randn(d.shape,generator=gen) for d in rdet]; rs=torch.randn(rseg.shape,generator=gen)*0.1
(sum((a*b).sum() for a,b in zip(rdet,rd))+(rseg*rs).sum()).backward()
rel=lambda a,b:((a.detach().cpu().float()-b).norm()/b.norm()).item()
for rep in range(3):
    m=Model(os.path.join(CFG,TAGS[tag])); m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p=0.0
    m=m.to(DEV).train()
    det,seg=m(x.to(DEV))
    (sum((a.float()*b.to(DEV)).sum() for a,b in zip(det,rd))+(seg.float()*rs.to(DEV)).sum()).backward()
    worst=max((rel(p.grad,params[k].grad),k) for k,p in m.named_parameters())
    print(rep, 'fwd seg', rel(seg,rseg), 'worst grad', worst)
    if worst[0] > 1e-2:
        for k,p in m.named_parameters():
            if k.startswith('model.24') and 'conv.weight' in k or k.startswith('model.24.m.3') or k.startswith('model.16') and 'cv3.conv' in k:
                print('   ', k, f'{rel(p.grad,params[k].grad):.2e}')
        # second backward on the same plan: deterministic?
        g1={k:p.grad.clone() for k,p in m.named_parameters()}
        for p in m.parameters(): p.grad=None
        det,seg=m(x.to(DEV))
        (sum((a.float()*b.to(DEV)).sum() for a,b in zip(det,rd))+(seg.float()*rs.to(DEV)).sum()).backward()
        w2=max((rel(p.grad,params[k].grad),k) for k,p in m.named_parameters())
        print('    second run on same plan: worst', w2, 'max diff run1-run2', max(rel(p.grad, g1[k].cpu().float()) for k,p in m.named_parameters()))
        break
