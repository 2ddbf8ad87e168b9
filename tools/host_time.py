import os, sys, time, torch
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd")); sys.path.insert(0, ROOT)
import bench
from hypernet_attention import HyperNet
from caphn.engine import FusedTrainer
dev=torch.device("cuda",0)
B,T,P,D,F,E,H,V=128,20,49,2048,200,200,200,9684
torch.manual_seed(1234)
net=HyperNet(F,E,H,V,bench._Vocab()).to(dev)
tr=FusedTrainer(net, lr=1e-3, max_norm=5.0)
batches=bench.synth_batches(4,B,T,P,D,V,dev,seed=1234)
nxt={batches[i][0].data_ptr(): batches[(i+1)%4][0] for i in range(4)}
def step(f,c): return tr.step(f,c,style_token=4,next_style_token=4,next_features=nxt[f.data_ptr()])
for i in range(5): step(*batches[i%4])
torch.cuda.synchronize()
N=40
t0=time.perf_counter()
for i in range(N): step(*batches[i%4])
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print("host enqueue per step %.3f ms, total per step %.3f ms"%((t1-t0)/N*1e3,(t2-t0)/N*1e3))
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for i in range(20): step(*batches[i%4])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
