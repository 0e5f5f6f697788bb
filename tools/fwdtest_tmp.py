import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from v1t_amd import lib as L
lib = L.load(); dev = torch.device("cuda:0")
for (B,H,T,DP,p) in [(1,1,1,32,0.0),(1,2,100,160,0.0),(2,4,1654,64,0.0),(2,4,1654,160,0.0),(1,4,1654,160,0.2544)]:
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(B*T, 3*H*DP, generator=g)*0.7).to(dev).bfloat16()
    scale = torch.tensor([DP**-0.5], device=dev)
    o = torch.empty(B*T, H*DP, device=dev, dtype=torch.bfloat16); lse = torch.empty(B,H,T, device=dev)
    print("launch", B,H,T,DP,p, flush=True)
    L.check(lib.v1t_attention_forward(qkv.data_ptr(), B,H,T,DP, scale.data_ptr(), 0,0, p, 1, 8, o.data_ptr(), lse.data_ptr(), L.stream()))
    torch.cuda.synchronize()
    print("ok", float(o.float().abs().mean()), flush=True)
