"""8-phase GEMM on the prefill / ViT shapes with padded leading dimensions (experiment: 2^13-byte row strides alias)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vcoder_amd import _lib
lib=_lib.load(); dev=torch.device("cuda:0"); P=lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/iters*1e3
ws=torch.zeros(16<<20, device=dev)
shapes=[(9728,12288,4096,0,"7b qkv"),(9728,4096,4096,4,"7b o"),(9728,22016,4096,5,"7b gate-up"),(9728,4096,11008,4,"7b down"),
        (19456,15360,5120,0,"13b qkv"),(19456,5120,5120,4,"13b o"),(19456,27648,5120,5,"13b gate-up"),
        (13848,3072,1024,0,"vit qkv"),(13848,1024,1024,4,"vit out"),(13848,4096,1024,1,"vit fc1"),(13848,1024,4096,4,"vit fc2"),(13824,4096,4096,0,"adapter2")]
for (M,N,K,epi,name) in shapes:
    res=[]
    for (pa,pw) in [(0,0),(0,64),(64,0),(64,64)]:
        A=(torch.randn(M,K+pa,device=dev)).to(torch.bfloat16); W=(torch.randn(N,K+pw,device=dev)*0.02).to(torch.bfloat16)
        out=torch.zeros((M,N),dtype=torch.float32 if epi in (3,4) else torch.bfloat16,device=dev)
        ldo=N//2 if epi==5 else N
        us=timeit(lambda: lib.vck_gemm_ws(P(A),P(W),None,P(out),M,N,K,K+pa,K+pw,ldo,epi,P(ws),C.c_size_t(64<<20),None))
        res.append(us)
    print(f"{name:12s} M{M} N{N} K{K}: lda/ldw pad (0,0) {res[0]:7.1f} us | (0,64) {res[1]:7.1f} | (64,0) {res[2]:7.1f} | (64,64) {res[3]:7.1f}   best/base {min(res)/res[0]:.3f}", flush=True)
