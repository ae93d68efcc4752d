import os, sys, torch, torch.nn.functional as F
sys.path.insert(0,'.')
from oracle import model_ref, synth
from tests.util import CFG, TAGS, load_cfg, synth_sd
from multiyolov5_amd.models.yolo import Model
from multiyolov5_amd import engine as E, _lib as L
tag='s_base'; DEV='cuda:0'
cfg=load_cfg(tag); sd=synth_sd(tag); x=synth.synth_images(2,64,128,seed=1)
params={k:v.clone().requires_grad_() for k,v in sd.items() if v.dtype.is_floating_point and 'running' not in k and 'anchor' not in k}
sdt={k:(params[k] if k in params else v.clone()) for k,v in sd.items()}
rdet,rseg=model_ref.forward(cfg,sdt,x,training=True,dropout_p=0.0)
gen=torch.Generator().manual_seed(5)
rd=[torch.randn(d.shape,generator=gen) for d in rdet]; rs=torch.randn(rseg.shape,generator=gen)*0.1
(sum((a*b).sum() for a,b in zip(rdet,rd))+(rseg*rs).sum()).backward()
rel=lambda a,b:((a.detach().cpu().float()-b.detach().cpu().float()).norm()/b.detach().cpu().float().norm().clamp_min(1e-20)).item()
for rep in range(6):
    m=Model(os.path.join(CFG,TAGS[tag])); m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p=0.0
    m=m.to(DEV).train()
    det,seg=m(x.to(DEV))
    (sum((a.float()*b.to(DEV)).sum() for a,b in zip(det,rd))+(seg.float()*rs.to(DEV)).sum()).backward()
    worst=max((rel(p.grad,params[k].grad),k) for k,p in m.named_parameters())
    print(rep,'worst',worst)
    if worst[0]<1e-2: continue
    h=list(m._plans.values())[0]; plan=h.plan
    # replay the backward op by op with the gradients still sitting in the output-grad tensors
    st=L.stream_ptr()
    plan._arena[1][:plan._used[1]].zero_(); plan.flat_grad.zero_()
    for oi in range(len(plan.ops)-1,-1,-1):
        op=plan.ops[oi]
        chk=isinstance(op,E.ConvOp) and op.bn is not None and not op.det
        if chk:
            gout=op.out.torch_view(grad=True).clone()
            xb=op.x.torch_view(grad=True).clone() if op.x.requires_grad else None
            wg_before=plan.pgrad(op.weight).clone()
        if isinstance(op,E.SppPoolOp):
            before=op.x.torch_view(grad=True).clone(); gs=[o.torch_view(grad=True).clone() for o in op.outs]
        for c in op.bwd_calls: c(st)
        if isinstance(op,E.SppPoolOp):
            after=op.x.torch_view(grad=True).clone()
            xv=op.x.torch_view().clone().permute(0,3,1,2).float().requires_grad_()
            tot=0
            for k,g in zip((5,9,13),gs): tot=tot+(F.max_pool2d(xv,k,1,k//2)*g.permute(0,3,1,2).float()).sum()
            tot.backward()
            print('   spp op',oi,'err',rel(after-before, xv.grad.permute(0,2,3,1)))
        if chk:
            xv=op.x.torch_view()[..., :op.cin].permute(0,3,1,2).float().detach().clone().requires_grad_(op.x.requires_grad)
            w=op.weight.detach().clone().requires_grad_()
            gam=op.bn.weight.detach().clone().requires_grad_(); bet=op.bn.bias.detach().clone().requires_grad_()
            y=F.conv2d(xv,w,None,op.s,op.pad,op.d)
            z=F.batch_norm(y,None,None,gam,bet,True,0.03,1e-3)
            o=F.silu(z) if op.act==L.ACT_SILU else z
            (o*gout.permute(0,3,1,2).float()).sum().backward()
            e_w=rel(plan.pgrad(op.weight)-wg_before, w.grad)
            e_x=-1
            if op.x.requires_grad:
                got=(op.x.torch_view(grad=True).float()-(xb.float() if op.acc_x else 0))[..., :op.cin]
                e_x=rel(got, xv.grad.permute(0,2,3,1))
            if max(e_w,e_x)>2e-3:
                print('   BAD conv op',oi,'cin',op.cin,'cout',op.cout,'k',op.k,'hw',op.out.h,op.out.w,'acc_x',op.acc_x,'zf',[(a,b) for _,a,b in op.zero_first],'xcoff',op.x.coff,'xc',op.x.c,'bufc',op.x.buf.c,f'e_w {e_w:.2e} e_x {e_x:.2e}')
    # compare the replayed flat grads with the oracle again
    off=0; w2=[]
    for (k,p) in m.named_parameters():
        n=p.numel(); g=plan.flat_grad[off:off+n].view(p.shape); off+=n
        w2.append((rel(g,params[k].grad),k))
    print('   replay worst', max(w2))
    break
