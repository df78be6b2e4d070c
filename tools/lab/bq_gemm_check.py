import sys; sys.path.insert(0,'/root/repo')
import torch
from adalog_amd import ops
torch.manual_seed(0)
dev='cuda'
worst=0
for (M,N,K) in [(64,128,16),(100,36,40),(197,64,197*0+200),(300,260,52),(1000,16,1000),(33,1000//4*4,384),(129,257*4,36)]:
  for ta in (0,1):
    for tb in (0,1):
      for bias in (False, True):
        if bias and N%16: continue
        a = torch.randn(K,M,device=dev).t() if ta else torch.randn(M,K,device=dev)
        b = torch.randn(K,N,device=dev).t() if tb else torch.randn(N,K,device=dev)
        bi = torch.randn(N,device=dev) if bias else None
        if not ops.gemm_f32x3_ok(a,b,bi): print('skip',M,N,K,ta,tb); continue
        out = ops.gemm_f32x3(a,b,bi)
        ex = a.double()@b.double().t() + (bi.double() if bias else 0)
        err = ((out.double()-ex).abs().max()/ex.abs().max()).item()
        worst=max(worst,err)
        if err>2e-6: print('BAD',M,N,K,ta,tb,bias,err)
# attention shapes: 197 tokens, head dim 64, unaligned rows / N
for (G,S,C) in [(12,197,64),(6,49,32)]:
    q=torch.randn(G,S,C,device=dev); k=torch.randn(G,S,C,device=dev); v=torch.randn(G,S,C,device=dev)
    kt=k.transpose(-1,-2).contiguous()          # [G, C, S] as the layer hands it over
    out=ops.gemm_f32x3(q, kt.transpose(-1,-2)); ex=q.double()@kt.double()
    e1=((out.double()-ex).abs().max()/ex.abs().max()).item()
    pr=torch.softmax(out,-1)
    o2=ops.gemm_f32x3(pr, v.transpose(-1,-2)); ex2=pr.double()@v.double()
    e2=((o2.double()-ex2).abs().max()/ex2.abs().max()).item()
    gy=torch.randn(G,S,C,device=dev)
    gp=ops.gemm_f32x3(gy, v); ex3=gy.double()@v.double().transpose(-1,-2)            # dL/dprobs = gy . v^T
    e3=((gp.double()-ex3).abs().max()/ex3.abs().max()).item()
    gv=ops.gemm_f32x3(pr.transpose(-1,-2), gy.transpose(-1,-2)); ex4=pr.double().transpose(-1,-2)@gy.double()   # dL/dv = probs^T . gy
    e4=((gv.double()-ex4).abs().max()/ex4.abs().max()).item()
    print('attention', G,S,C, e1,e2,e3,e4); worst=max(worst,e1,e2,e3,e4)
# batched
a=torch.randn(6,197,64,device=dev); b=torch.randn(6,200,64,device=dev)
out=ops.gemm_f32x3(a,b); ex=a.double()@b.double().transpose(-1,-2)
print('batched', ((out.double()-ex).abs().max()/ex.abs().max()).item())
# exact-integer operand forms: forward (a integer, K-contiguous both) and dL/dw (b integer, K-major both)
for (M,N,K) in [(300,64,48),(6304,384,384)]:
    a = torch.randint(-15,16,(M,K),device=dev).float(); b = torch.randn(N,K,device=dev); sc = torch.tensor([0.37],device=dev)
    out = ops.gemm_f32x3(a,b,alpha_dev=sc,exact_a=True); ex = (a.double()@b.double().t())*0.37
    e1=((out.double()-ex).abs().max()/ex.abs().max()).item()
    g = torch.randn(K*4, M//4*4 if M<1000 else 384, device=dev)   # [tokens, O]
    xi = torch.randint(-15,16,(K*4, N),device=dev).float()       # [tokens, I]
    out2 = ops.gemm_f32x3(g.t(), xi.t(), alpha_dev=sc, exact_b=True); ex2=(g.double().t()@xi.double())*sc.double()
    e2=((out2.double()-ex2).abs().max()/ex2.abs().max()).item()
    print('exact forms', M,N,K, e1, e2); worst=max(worst,e1,e2)
# pre-split B planes
for (M,N,K) in [(300,64,48),(200,144,40),(6304,384,1152),(6304,1536,384)]:
    for ex in (False, True):
        a = torch.randint(-15,16,(M,K),device=dev).float() if ex else torch.randn(M,K,device=dev)
        b = torch.randn(N,K,device=dev); bi = torch.randn(N,device=dev) if N%16==0 else None
        bp = ops.pack_split3(b.view(1,N,K), 64)
        out = ops.gemm_f32x3_planes(a, bp, K, bi, exact_a=ex)
        exa = a.double()@b.double().t() + (bi.double() if bi is not None else 0)
        e=((out.double()-exa).abs().max()/exa.abs().max()).item()
        print('planes', M,N,K,ex,e); worst=max(worst,e)
print('worst',worst)
