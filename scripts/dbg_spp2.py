import os, sys, torch, torch.nn.functional as F
sys.path.insert(0,'.')
from oracle import model_ref, synth
from tests.util import CFG, TAGS, load_cfg, synth_sd
from multiyolov5_amd.models.yolo import Model
from multiyolov5_amd import engine as E, _lib as L
tag='s_base'; DEV='cuda:0'
cfg=load_cfg(tag); sd=synth_sd(tag); x=synth.synth_images(2,64,128,seed=1)
cap={}
orig_spp=model_ref.spp
def spp(ctx,p,xx,ks=(5,9,13)):
    x1=model_ref.conv_block(ctx,p+'.cv1',xx)
    if p.startswith('model.24'): x1.retain_grad(); cap['x']=x1
    pools=[F.max_pool2d(x1,k,1,k//2) for k in ks]
    cat=torch.cat([x1]+pools,1)
    if p.startswith('model.24'): cat.retain_grad(); cap['cat']=cat
    return model_ref.conv_block(ctx,p+'.cv2',cat)
model_ref.spp=spp
params={k:v.clone().requires_grad_() for k,v in sd.items() if v.dtype.is_floating_point and 'running' not in k and 'anchor' not in k}
sdt={k:(params[k] if k in params else v.clone()) for k,v in sd.items()}
rdet,rseg=model_ref.forward(cfg,sdt,x,training=True,dropout_p=0.0)
gen=torch.Generator().manual_seed(5)
rd=[torch.randn(d.shape,generator=gen) for d in rdet]; rs=torch.randn(rseg.shape,generator=gen)*0.1
(sum((a*b).sum() for a,b in zip(rdet,rd))+(rseg*rs).sum()).backward()
rel=lambda a,b:((a.detach().cpu().float()-b.detach().cpu().float()).norm()/b.detach().cpu().float().norm().clamp_min(1e-20)).item()
for rep in range(3):
    m=Model(os.path.join(CFG,TAGS[tag])); m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p=0.0
    m=m.to(DEV).train()
    det,seg=m(x.to(DEV))
    (sum((a.float()*b.to(DEV)).sum() for a,b in zip(det,rd))+(seg.float()*rs.to(DEV)).sum()).backward()
    worst=max((rel(p.grad,params[k].grad),k) for k,p in m.named_parameters())
    plan=list(m._plans.values())[0].plan
    op=[o for o in plan.ops if isinstance(o,E.SppPoolOp)][-1]
    nchw=lambda t:t.permute(0,3,1,2)
    full=op.outs[0].buf
    print(rep,'worst',worst[0])
    print('   fwd x', rel(nchw(op.x.torch_view()),cap['x']), 'fwd cat', rel(nchw(full.t),cap['cat']))
    gcat=nchw(full.g)
    print('   grad cat slices (after all bwd): x', rel(gcat[:, :64],cap['x'].grad), ' pools', rel(gcat[:,64:],cap['cat'].grad[:,64:]), ' cat[:64] direct part', rel(gcat[:, :64], cap['cat'].grad[:, :64]))
    # argmax agreement of the 13x13 pool between CPU oracle activations and GPU activations
    xc=cap['x'].detach(); xg=nchw(op.x.torch_view()).cpu().float()
    _,ic=F.max_pool2d(xc,13,1,6,return_indices=True); _,ig=F.max_pool2d(xg,13,1,6,return_indices=True)
    print('   argmax13 mismatch frac', (ic!=ig).float().mean().item())
    # CPU vs GPU maxpool backward on identical data
    xg2=xg.clone().requires_grad_(); g=torch.randn(xg.shape, generator=torch.Generator().manual_seed(1))
    (F.max_pool2d(xg2,13,1,6)*g).sum().backward()
    xd=xg.to(DEV).requires_grad_(); (F.max_pool2d(xd,13,1,6)*g.to(DEV)).sum().backward()
    print('   torch cpu vs gpu maxpool13 bwd', rel(xd.grad, xg2.grad))
