import os, sys, torch, torch.nn.functional as F
sys.path.insert(0,'.')
from oracle import synth
from tests.util import CFG, TAGS, synth_sd
from multiyolov5_amd.models.yolo import Model
from multiyolov5_amd import engine as E, _lib as L
tag=sys.argv[1] if len(sys.argv)>1 else 's_base'; DEV='cuda:0'
sd=synth_sd(tag); x=synth.synth_images(2,64,128,seed=1)
rel=lambda a,b:((a.float()-b.float()).norm()/b.float().norm().clamp_min(1e-20)).item()
for rep in range(3):
    m=Model(os.path.join(CFG,TAGS[tag])); m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p=0.0
    m=m.to(DEV).train()
    det,seg=m(x.to(DEV))
    h=list(m._plans.values())[0]; plan=h.plan
    gen=torch.Generator().manual_seed(5)
    for s in range(h.ospec.nslots):
        t=h.output_grad_tensor(s); t.copy_(torch.randn(t.shape,generator=gen).to(DEV)*0.1)
    st=L.stream_ptr()
    plan._arena[1][:plan._used[1]].zero_(); plan.flat_grad.zero_()
    nbad=0
    for oi in range(len(plan.ops)-1,-1,-1):
        op=plan.ops[oi]
        chk=isinstance(op,E.ConvOp) and op.bn is not None and not op.det
        if chk:
            gout=op.out.torch_view(grad=True).clone()
            xb=op.x.torch_view(grad=True).clone() if op.x.requires_grad else None
            wg_before=plan.pgrad(op.weight).clone()
            rb=op.res.torch_view(grad=True).clone() if (op.res is not None and op.res.requires_grad) else None
        for c in op.bwd_calls: c(st)
        if chk:
            xv=op.x.torch_view()[..., :op.cin].permute(0,3,1,2).float().detach().clone().requires_grad_(op.x.requires_grad)
            w=op.weight.detach().clone().requires_grad_()
            gam=op.bn.weight.detach().clone().requires_grad_(); bet=op.bn.bias.detach().clone().requires_grad_()
            y=F.conv2d(xv,w,None,op.s,op.pad,op.d)
            z=F.batch_norm(y,None,None,gam,bet,True,0.03,1e-3)
            o=F.silu(z) if op.act==L.ACT_SILU else z
            if op.res is not None: o=o+op.res.torch_view().permute(0,3,1,2).float()
            (o*gout.permute(0,3,1,2).float()).sum().backward()
            e_w=rel(plan.pgrad(op.weight)-wg_before, w.grad)
            e_x=-1
            if op.x.requires_grad:
                got=(op.x.torch_view(grad=True).float()-(xb.float() if op.acc_x else 0))[..., :op.cin]
                e_x=rel(got, xv.grad.permute(0,2,3,1))
            if max(e_w,e_x)>2e-3:
                nbad+=1
                print(rep,'op',oi,'cin',op.cin,'cout',op.cout,'k',op.k,'s',op.s,'hw',op.out.h,op.out.w,'acc_x',op.acc_x,'zero_first',[(a,b) for _,a,b in op.zero_first],'x.coff',op.x.coff,'x.c',op.x.c,'bufc',op.x.buf.c,f'e_w {e_w:.2e} e_x {e_x:.2e}')
    print(rep,'bad convs',nbad)
