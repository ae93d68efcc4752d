import sys, torch
sys.path.insert(0, '.')
from multiyolov5_amd.models import common as C
B=16; dt=torch.float16
SHAPES=[(64,64,3,1,64,128),(64,64,1,1,128,256),(256,128,3,1,64,128)]
for cin,cout,k,s,H,W in SHAPES:
    m=C.Conv(cin,cout,k,s).to('cuda').train()
    xs=[torch.randn(B,cin,H,W,device='cuda',dtype=dt).contiguous(memory_format=torch.channels_last).requires_grad_() for _ in range(3)]
    for r in range(9):
        y=m(xs[r%3]); y.backward(torch.ones_like(y)*1e-3)
    torch.cuda.synchronize()
