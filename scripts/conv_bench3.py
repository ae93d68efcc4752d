import sys, ctypes, torch
sys.path.insert(0, '.')
from multiyolov5_amd.models import common as C
B=16; dt=torch.float16
mode=sys.argv[1]
SHAPES=[(64,64,1,1,128,256),(128,128,1,1,64,128),(64,64,3,1,64,128)]
for cin,cout,k,s,H,W in SHAPES:
    m=C.Conv(cin,cout,k,s).to('cuda')
    m.train(mode=='train')
    xs=[torch.randn(B,cin,H,W,device='cuda',dtype=dt).contiguous(memory_format=torch.channels_last) for _ in range(5)]
    with torch.set_grad_enabled(False):
        for r in range(15):
            m(xs[r%5])
    torch.cuda.synchronize()
