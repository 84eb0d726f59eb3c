import torch, time, sys
sys.path.insert(0,'.')
from consistencytta_amd import _native as N
L=N.lib(); st=N.stream_ptr()
dev='cuda:0'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for (K,Nn,S) in [(9216,1024,1),(4608,512,1),(2304,256,8),(1024,8192,1),(256,2048,16)]:
    slab=torch.randn(S,K,Nn,device=dev); slabT=torch.randn(S,Nn,K,device=dev)
    grad=torch.zeros(Nn,K,device=dev)
    ro=(torch.arange(Nn,dtype=torch.int32)*K).to(dev)
    a=t(lambda: N.check(L.ctta_wgrad_scatter(N.ptr(slab),S,K*Nn,Nn,K,Nn,N.ptr(ro),None,None,None,0,N.ptr(grad),1,st)))
    b=t(lambda: N.check(L.ctta_wgrad_scatter_rows(N.ptr(slabT),S,K*Nn,K,K,Nn,N.ptr(ro),None,N.ptr(grad),1,st)))
    c=t(lambda: grad.add_(slabT[0]))
    gb=(S+2)*K*Nn*4/1e9
    print("K=%d N=%d S=%d tiled %.1f us (%.0f GB/s)  rows %.1f us (%.0f GB/s)  torch add_ %.1f us" % (K,Nn,S,a,gb/a*1e6,b,gb/b*1e6,c))
