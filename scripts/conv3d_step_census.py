import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dfmir_amd import ops
from dfmir_amd.registration3d import Registration3DModel
PLUGIN = [[16, 32, 32, 64, 64, 64], [64, 64, 64, 32, 32, 32, 16]]
small = os.environ.get("CENSUS") == "128"          # CENSUS=128: the 128^3 volume with the plugin's 6-level features
shape=(128,128,128) if small else (160,192,224)
torch.manual_seed(0)
m=Registration3DModel(shape, PLUGIN if small else None)
A=torch.rand(1,1,*shape,device="cuda")*2-1; B=0.5*A+0.5*(torch.rand(1,1,*shape,device="cuda")*2-1)
recs=[]
def prof(kind, flops, launch):
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()                                   # (an eager step is host-bound at 128^3: time each launch alone)
    s.record(); launch(); e.record(); recs.append((kind,flops,s,e))
for _ in range(2):
    m.set_input({"A":A,"B":B}); m.optimize_parameters()
ops.set_conv_profiler(prof)
m.set_input({"A":A,"B":B}); m.optimize_parameters()
torch.cuda.synchronize()
tot=0
for kind,fl,s,e in recs:
    ms=s.elapsed_time(e); tot+=ms
    print("%-16s %8.3f ms %8.2f GF %7.1f TF"%(kind,ms,fl/1e9,fl/ms/1e9))
print("conv total", tot)
