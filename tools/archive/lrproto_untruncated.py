import sys, numpy as np, scipy.linalg as sl
sys.path.insert(0, "/root/repo/tools")
EPS=np.finfo(float).eps
d=np.load("/tmp/proto/sys_C3_20000_2000_0.02.npz"); U=d["U"]
def pchol(A, tol):
    n=A.shape[0]; dg=np.diag(A).copy(); Lm=np.zeros((n,n)); r=0; used=np.zeros(n,bool); order=[]
    while True:
        dm=np.where(used,-np.inf,dg); p=int(np.argmax(dm))
        if dm[p]<=tol: break
        c=A[:,p]-Lm[:,:r]@Lm[p,:r]; c[used]=0.0; c/=np.sqrt(c[p]); Lm[:,r]=c; dg-=c*c; used[p]=True; order.append(p); r+=1
    return Lm[:,:r].copy(), np.array(order)
def ferr(C,Cr): V,Vr=U@C,U@Cr; return np.abs(V-Vr).max()/np.abs(Vr).max()
for it in (1,3,5):
    A=d[f"lhs{it}"]; R=d[f"rhs{it}"]; A=0.5*(A+A.T); Cref=d[f"C{it}"]
    w,Q=np.linalg.eigh(A); lmax=w[-1]; k=w>EPS*lmax; Ce=Q[:,k]@((Q[:,k].T@R)/w[k][:,None])
    for tolf in (0.25, 1.0, 4.0):
        L,order=pchol(A, tolf*EPS*lmax); r=L.shape[1]
        # un-truncated minimum-norm solution of L L^T C = R through QR of L (orthogonal: accurate)
        Qh,Rh=np.linalg.qr(L)             # L = Qh Rh, Rh r x r upper
        t=Qh.T@R; y=sl.solve_triangular(Rh, t, lower=False); z=sl.solve_triangular(Rh, y, trans="T", lower=False); C=Qh@z
        print(f"it {it} tolf {tolf}: r {r} kept(eigh) {k.sum()}  field vs lstsq: eigh-variant {ferr(Ce,Cref):.2e}  un-truncated pivoted factor {ferr(C,Cref):.2e}")
print("--- Cholesky-of-L^T L route (what existing device kernels could do)")
for it in (3,5):
    A=d[f"lhs{it}"]; R=d[f"rhs{it}"]; A=0.5*(A+A.T); Cref=d[f"C{it}"]
    w=np.linalg.eigvalsh(A); lmax=w[-1]
    L,order=pchol(A, 1.0*EPS*lmax)
    Qh,Rh=np.linalg.qr(L); t=Qh.T@R; y=sl.solve_triangular(Rh,t,lower=False); z=sl.solve_triangular(Rh,y,trans="T",lower=False); Cqr=Qh@z
    S2=L.T@L; Rt=np.linalg.cholesky(S2)   # S2 = Rt Rt^T
    t=L.T@R
    def s2solve(b):
        y=sl.solve_triangular(Rt,b,lower=True); return sl.solve_triangular(Rt,y,trans="T",lower=True)
    Cch=L@s2solve(s2solve(t))
    print(f"it {it}: QR route vs lstsq {ferr(Cqr,Cref):.2e}; Cholesky route vs lstsq {ferr(Cch,Cref):.2e}; Cholesky vs QR route {ferr(Cch,Cqr):.2e}")
