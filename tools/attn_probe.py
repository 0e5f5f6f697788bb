import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from v1t_amd import lib as L
lib = L.load(); dev = torch.device("cuda:0")
def run(B,H,T,DP,p=0.0):
    g = torch.Generator().manual_seed(B*1000+T+DP)
    qkv = (torch.randn(B*T, 3*H*DP, generator=g)*0.7).to(dev).bfloat16()
    scale = torch.tensor([DP**-0.5], device=dev)
    o = torch.empty(B*T, H*DP, device=dev, dtype=torch.bfloat16); lse = torch.empty(B,H,T, device=dev)
    L.check(lib.v1t_attention_forward(qkv.data_ptr(), B,H,T,DP, scale.data_ptr(), 0,0, p, 4242, 16, o.data_ptr(), lse.data_ptr(), L.stream()))
    dO = (torch.randn(B*T, H*DP, generator=g)*0.5).to(dev).bfloat16()
    q,k,v = qkv.float().view(B,T,3,H,DP).permute(2,0,3,1,4)
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    a = torch.softmax((q@k.transpose(-1,-2))*scale, -1)
    ref = (a@v).permute(0,2,1,3).reshape(B*T, H*DP)
    gq,gk,gv = torch.autograd.grad(ref, (q,k,v), dO.float())
    delta = torch.empty(B,H,T, device=dev)
    d1 = torch.empty_like(qkv)
    L.check(lib.v1t_attention_backward(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B,H,T,DP, scale.data_ptr(),0,0,p,4242,16, delta.data_ptr(), d1.data_ptr(), None, L.stream()))
    nb = int(lib.v1t_attention_backward_ws_bytes(B,H,T)); ws = torch.full((nb,),0xFF,dtype=torch.uint8,device=dev)
    d2 = torch.zeros_like(qkv)
    L.check(lib.v1t_attention_backward_ws(qkv.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), B,H,T,DP, scale.data_ptr(),0,0,p,4242,16, delta.data_ptr(), d2.data_ptr(), None, ws.data_ptr(), nb, L.stream()))
    torch.cuda.synchronize()
    out = []
    for nm, d in (("recompute", d1), ("dS'", d2)):
        dd = d.float().view(B,T,3,H,DP).permute(2,0,3,1,4)
        errs = []
        for i,(r,n) in enumerate(((gq,"q"),(gk,"k"),(gv,"v"))):
            e = (dd[i]-r).abs(); errs.append(f"{n} {float(e.max()/r.abs().max()):.2e} worst row {int(e.amax(-1).flatten().argmax()) % T}")
        out.append(f"{nm}: " + ", ".join(errs))
    print(f"B{B} H{H} T{T} DP{DP}: " + " | ".join(out), flush=True)
for cfg in ((2,1,130,128),(2,1,130,160),(1,1,130,128),(2,1,300,128),(1,2,200,160),(1,1,129,160),(1,1,160,160),(1,1,161,160),(1,1,97,160),(1,1,33,160)):
    run(*cfg)
