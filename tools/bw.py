import torch, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
from edadm import ops
dev=torch.device("cuda",0)
n=409600*192
a=torch.randn(n,device=dev); b=torch.empty(n,device=dev)
def t(fn,k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/k
ms=t(lambda: b.fill_(1.0)); print("fill  %.3f ms  %.2f TB/s write"%(ms, n*4/ms/1e9))
ms=t(lambda: b.copy_(a)); print("copy  %.3f ms  %.2f TB/s total"%(ms, 2*n*4/ms/1e9))
ms=t(lambda: ops.add(a,a)); print("edadm add %.3f ms  %.2f TB/s total (2r+1w incl alloc)"%(ms, 3*n*4/ms/1e9))
x=torch.randn(409600,192,device=dev)
qp=ops.qp_tensor([(0.02,128.0,255)],dev)
ms=t(lambda: ops.quant_i8(x,qp)); print("quant_i8 %.3f ms  %.2f TB/s"%(ms, n*5/ms/1e9))
