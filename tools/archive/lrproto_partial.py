"""CPU prototype (design probe): the truncated minimum-norm solve WITHOUT a full eigendecomposition.
  A ~ L L^T (pivoted factor, r columns, stopped at tolf eps lmax).  Un-truncated minimum-norm solution through S2 = L^T L:
  C0 = L S2^-2 L^T R.  gelsd's cut-off removes the eigenpairs of A below eps lmax - the SMALLEST few dozen eigenpairs of S2:
  C = L Pc S2^-1 Pc S2^-1 Pc L^T R with Pc = I - W W^T (subtracting the amplified terms from C0 cancels catastrophically; the
  deflated solve does not).  W comes from a block inverse iteration on S2 (Cholesky
  factor at hand) + Rayleigh-Ritz.  Question: block size / iterations needed for the field to match the exact truncated solve."""
import sys, numpy as np, scipy.linalg as sl
EPS=np.finfo(float).eps
d=np.load(sys.argv[1] if len(sys.argv)>1 else "/tmp/proto/sys_C3_20000_2000_0.02.npz"); U=d["U"]
def pchol(A, tol):
    n=A.shape[0]; dg=np.diag(A).copy(); Lm=np.zeros((n,n)); r=0; used=np.zeros(n,bool)
    while True:
        dm=np.where(used,-np.inf,dg); p=int(np.argmax(dm))
        if dm[p]<=tol: break
        c=A[:,p]-Lm[:,:r]@Lm[p,:r]; c[used]=0.0; c/=np.sqrt(c[p]); Lm[:,r]=c; dg-=c*c; used[p]=True; r+=1
    return Lm[:,:r].copy()
def ferr(C,Cr): V,Vr=U@C,U@Cr; return np.abs(V-Vr).max()/np.abs(Vr).max()
rng=np.random.default_rng(0)
for it in (1,3,5):
    A=d[f"lhs{it}"]; R=d[f"rhs{it}"]; A=0.5*(A+A.T); Cref=d[f"C{it}"]
    w,Q=np.linalg.eigh(A); lmax=w[-1]; cut=EPS*lmax; k=w>cut; Ce=Q[:,k]@((Q[:,k].T@R)/w[k][:,None])
    L=pchol(A,0.25*EPS*lmax); r=L.shape[1]
    # exact truncated solve of the factor (what the Jacobi iteration delivers)
    Uu,s,_=np.linalg.svd(L,full_matrices=False); kk=s**2>EPS*s[0]**2; Cx=Uu[:,kk]@((Uu[:,kk].T@R)/(s[kk]**2)[:,None])
    S2=L.T@L; Rt=np.linalg.cholesky(S2)
    s2inv=lambda B: sl.solve_triangular(Rt, sl.solve_triangular(Rt,B,lower=True), trans="T", lower=True)
    t=L.T@R; C0=L@s2inv(s2inv(t))
    nbelow=int((s**2<=EPS*s[0]**2).sum())
    print(f"it {it}: r {r}, eigenpairs of the factor below the cut: {nbelow}; floor(eigh vs lstsq) {ferr(Ce,Cref):.2e}; exact-truncated vs lstsq {ferr(Cx,Cref):.2e}; un-truncated vs exact {ferr(C0,Cx):.2e}")
    for blk in (128,192,256):
        Z=rng.standard_normal((r,blk))
        for nit in (1,2,3,4):
            Z=s2inv(Z); Z,_=np.linalg.qr(Z)
            H=Z.T@S2@Z; th,Y=np.linalg.eigh((H+H.T)/2); Wr=Z@Y
            sel=th<EPS*th.max()*0+cut*1.0   # Ritz values below the cut (lmax of A == max eigenvalue of S2)
            W=Wr[:,sel]; thr=th[sel]
            proj=lambda B: B-W@(W.T@B)          # projector onto the complement of the small-eigenvalue subspace
            C=L@proj(s2inv(proj(s2inv(proj(t)))))   # deflated solve: the amplified rounding noise lives in span(W) and is projected out
            print(f"    block {blk} inverse iterations {nit}: Ritz values below cut {int(sel.sum())}; field vs exact-truncated {ferr(C,Cx):.2e}; vs lstsq {ferr(C,Cref):.2e}")
