import sys,torch,time
sys.path.insert(0,"aloception-oss_amd"); sys.path.insert(0,"tools")
import alo_hip, kbench
f1,f2=kbench.corr_inputs(4)
levels=alo_hip.corr_build(f1,f2,4)
H,W=90,160
coords=torch.stack(torch.meshgrid(torch.arange(W,device="cuda"),torch.arange(H,device="cuda"),indexing="xy")).float()[None].repeat(4,1,1,1)+torch.randn(4,2,H,W,device="cuda")*4
w=torch.randn(256,324,1,1,device="cuda")/18; b=torch.randn(256,device="cuda")
t=kbench.time_launches(lambda: alo_hip.corr_lookup_conv1x1(levels,coords,w,b,4,True),20); print("fused lookup+convc1 ms",t*1e3)
t=kbench.time_launches(lambda: alo_hip.corr_lookup(levels,coords,4),20); print("lookup ms",t*1e3)
def unf():
    c=alo_hip.corr_lookup(levels,coords,4); o=torch.nn.functional.conv2d(c,w); return alo_hip.bias_act_nchw_(o,b,True)
t=kbench.time_launches(unf,20); print("lookup+conv+bias_relu ms",t*1e3)
