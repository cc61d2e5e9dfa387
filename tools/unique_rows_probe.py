import os, sys, time
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path[:0]=[R, os.path.join(R,"spateo-release_amd")]
import numpy as np, torch
from spateo_amd._kernels import HipKernels
from spateo_amd._synthetic import make_config
k=HipKernels("cuda:0","float64")
X,_,_=make_config("C4",N=8_000_000)
X[5]=X[7]; X[100:200]=X[300:400]
for rep in range(3):
    torch.cuda.synchronize(); t=time.perf_counter(); S,idx=k.unique_rows(X); torch.cuda.synchronize(); print("device unique 8M rows: %.1f ms (incl. H2D/D2H), %d unique" % (1e3*(time.perf_counter()-t), len(S)))
from spateo_amd.vectorfield import unique_rows
import spateo_amd.vectorfield as vfm
import spateo_amd.preprocess as _pre; old=_pre._DEVICE_UNIQUE_MIN_ROWS; _pre._DEVICE_UNIQUE_MIN_ROWS=10**12
t=time.perf_counter(); Sh,ih=unique_rows(X); print("host path %.2f s" % (time.perf_counter()-t))
print("identical:", np.array_equal(S,Sh), np.array_equal(idx,ih))
