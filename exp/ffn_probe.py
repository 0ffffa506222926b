import sys; sys.path.insert(0,"aloception-oss_amd")
import torch, alo_hip
dev="cuda:0"
def t(fn, reps=12):
    for _ in range(4): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/reps*1e3
for (M,F) in [(177784,1024),(2400,1024)]:
    n=4 if M>10000 else 1
    xs=[torch.randn(M,256,device=dev,dtype=torch.bfloat16) for _ in range(n)]
    w1=(torch.randn(F,256,device=dev)*0.06).bfloat16(); b1=torch.randn(F,device=dev).bfloat16()
    w2=(torch.randn(256,F,device=dev)*0.03).bfloat16(); b2=torch.randn(256,device=dev).bfloat16()
    i=[0]
    def g():
        i[0]+=1; return alo_hip.ffn256(xs[i[0]%n], w1, b1, w2, b2)
    x=xs[0]
    h=(x.float()@w1.float().t()+b1.float()).relu().bfloat16().float()
    ref=h@w2.float().t()+b2.float()
    y=alo_hip.ffn256(x,w1,b1,w2,b2)
    print((M,F), "fused us", round(t(g),1), "err", round((y.float()-ref).abs().max().item(),4))
