import torch, time, sys
sys.path.insert(0, '.')
from snag_amd import ops, _lib
dev='cuda'
g=torch.Generator().manual_seed(1234)
N,R,D,K=15000,558,128,32
ent=(torch.randn(N,D,generator=g)).to(dev); rel=torch.randn(R,D,generator=g).to(dev)
for B in (1024, 8192, 79872):
    h=torch.randint(0,N,(B,),generator=g,dtype=torch.int32).to(dev); r=torch.randint(0,R,(B,),generator=g,dtype=torch.int32).to(dev)
    t=torch.randint(0,N,(B,),generator=g,dtype=torch.int32).to(dev); neg=torch.randint(0,N,(B,K),generator=g,dtype=torch.int32).to(dev)
    pos=torch.empty(B,device=dev); ng=torch.empty(B,K,device=dev); lp=torch.empty(B,device=dev)
    de=torch.zeros_like(ent); dr=torch.zeros_like(rel)
    st=_lib.stream()
    def fwd(): _lib.call("snag_score_margin_fwd", ent.data_ptr(),D,N,rel.data_ptr(),R,D,h.data_ptr(),r.data_ptr(),t.data_ptr(),neg.data_ptr(),4,B,K,12.0,pos.data_ptr(),ng.data_ptr(),lp.data_ptr(),st)
    def bwd(): _lib.call("snag_score_margin_bwd", ent.data_ptr(),D,N,rel.data_ptr(),R,D,h.data_ptr(),r.data_ptr(),t.data_ptr(),neg.data_ptr(),4,B,K,12.0,1.0,None,None,de.data_ptr(),D,dr.data_ptr(),st)
    for name,fn in (("fwd",fwd),("bwd",bwd)):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        n=200
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)*1e3/n
        bytes_=B*18192
        print(f"B={B} {name}: {us:.2f} us/launch  alg {bytes_/us/1e6:.2f} TB/s ({bytes_/us/1e6/8*100:.1f}% of 8TB/s) {B/us:.1f} Mtriples/s")
