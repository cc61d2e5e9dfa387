import sys, os, time
sys.path[:0]=[os.environ.get("GRAFT_REPO_ROOT","/root/repo"), os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"spateo-release_amd")]
import numpy as np, torch
from spateo_amd._kernels import HipKernels
k=HipKernels("cuda:0","float64")
rng=np.random.default_rng(0)
for m in (100, 500):
    A=rng.standard_normal((m,3*m)); G=torch.from_numpy(A@A.T/m).cuda(); K=torch.zeros(m,m,dtype=torch.float64,device="cuda")
    for nrhs in (3,4,5,8):
        R=torch.randn(m,nrhs,dtype=torch.float64,device="cuda"); C=torch.empty_like(R); info=torch.zeros(1,dtype=torch.int32,device="cuda"); piv=torch.zeros(2,dtype=torch.float64,device="cuda")
        for _ in range(5): k.solve(G,K,0.1,0.0,R,C,info,piv)
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): k.solve(G,K,0.1,0.0,R,C,info,piv)
        e1.record(); torch.cuda.synchronize()
        print(m, nrhs, f"{e0.elapsed_time(e1)/200*1e3:.1f} us per solve")
