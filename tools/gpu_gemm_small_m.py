import sys, torch
sys.path.insert(0, ".")
import ai_toolkit_amd  # noqa: F401  (registers the `ai_toolkit_amd` package alias)
from ai_toolkit_amd import ops
def t(fn, n=5, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(n):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/it)
    return sorted(ts)[n//2]
for (M,N,K) in ((2048,12288,3072),(2048,3072,12288),(2048,3072,3072),(4096,3072,3072),(2048,9216,3072)):
    a=torch.randn(M,K,device="cuda").to(torch.bfloat16); b=(torch.randn(N,K,device="cuda")*0.02).to(torch.bfloat16)
    a2=torch.randn(M,16,device="cuda").to(torch.bfloat16); b2=torch.randn(N,16,device="cuda").to(torch.bfloat16)
    bias=torch.randn(N,device="cuda").to(torch.bfloat16); out=torch.empty(M,N,dtype=torch.bfloat16,device="cuda")
    res={}
    for name,kw in (("auto",{}),("8ph",dict(stage_mode=4,tile_mode=2)),("old256",dict(stage_mode=5,tile_mode=2)),("old128",dict(stage_mode=5,tile_mode=1))):
        ms=t(lambda: ops.gemm_nt(a,b,out,bias=bias,a2=a2,b2=b2,**kw))
        res[name]=(round(ms*1000,1), round(2*M*N*K/ms/1e9))
    res["hipblaslt"]=(lambda ms:(round(ms*1000,1), round(2*M*N*K/ms/1e9)))(t(lambda: torch.matmul(a,b.t(),out=out)))
    print(M,N,K,res,flush=True)
