"""CPU prototype, second stage: the partial-eigenpair truncated solve with ONLY the operations the device has or can get
cheaply - explicit inverse of the graded Cholesky factor (all solves become GEMMs), Cholesky-QR orthonormalisation, a start
block on the trailing (smallest-pivot) rows, Rayleigh-Ritz on a b x b matrix.  python tools/lrproto_partial2.py [npz] [block] [iters]"""
import sys, numpy as np, scipy.linalg as sl
EPS=np.finfo(float).eps
d=np.load(sys.argv[1] if len(sys.argv)>1 and sys.argv[1].endswith(".npz") else "/tmp/proto/sys_C3_20000_2000_0.02.npz"); U=d["U"]
def pchol(A, tol):
    n=A.shape[0]; dg=np.diag(A).copy(); Lm=np.zeros((n,n)); r=0; used=np.zeros(n,bool)
    while True:
        dm=np.where(used,-np.inf,dg); p=int(np.argmax(dm))
        if dm[p]<=tol: break
        c=A[:,p]-Lm[:,:r]@Lm[p,:r]; c[used]=0.0; c/=np.sqrt(c[p]); Lm[:,r]=c; dg-=c*c; used[p]=True; r+=1
    return Lm[:,:r].copy()
def ferr(C,Cr): V,Vr=U@C,U@Cr; return np.abs(V-Vr).max()/np.abs(Vr).max()
def cholqr_rows(Zt, passes=2):
    for _ in range(passes):
        G=Zt@Zt.T; Lg=np.linalg.cholesky(G); Zt=np.linalg.inv(Lg)@Zt
    return Zt
for it in (1,3,5):
    A=d[f"lhs{it}"]; R=d[f"rhs{it}"]; A=0.5*(A+A.T); Cref=d[f"C{it}"]
    lmax=np.linalg.eigvalsh(A)[-1]; cut=EPS*lmax
    L=pchol(A,0.25*EPS*lmax); r=L.shape[1]
    Uu,s,_=np.linalg.svd(L,full_matrices=False); kk=s**2>EPS*s[0]**2; Cx=Uu[:,kk]@((Uu[:,kk].T@R)/(s[kk]**2)[:,None])
    S2=L.T@L; Rt=np.linalg.cholesky(S2); Ri=np.linalg.inv(Rt)        # explicit inverse of the graded factor
    s2inv_rows=lambda Zt: (Zt@Ri.T)@Ri                                 # rows of Zt <- S2^-1 applied (S2^-1 = Ri^T Ri)
    t=L.T@R
    for b in (192,256):
        for start in ("trailing","random"):
            Zt=np.zeros((b,r)); 
            if start=="trailing": Zt[np.arange(b), r-b+np.arange(b)]=1.0
            else: Zt=np.random.default_rng(0).standard_normal((b,r))
            for nit in range(1,5):
                Zt=cholqr_rows(s2inv_rows(Zt))
                H=(Zt@S2)@Zt.T; th,Yh=np.linalg.eigh((H+H.T)/2)
                sel=th<cut; Wt=(Yh[:,sel].T)@Zt                         # rows = Ritz vectors below the cut
                proj=lambda B: B-Wt.T@(Wt@B)
                z=proj(t); z=proj(Ri.T@(Ri@z)); z=proj(Ri.T@(Ri@z)); C=L@z
                print(f"it {it} r {r} block {b} start {start} iterations {nit}: below cut {int(sel.sum())} (exact {int((~kk).sum())}); cond(H) {th.max()/max(th.min(),1e-300):.1e}; field vs exact-truncated {ferr(C,Cx):.2e} vs lstsq {ferr(C,Cref):.2e}",flush=True)
