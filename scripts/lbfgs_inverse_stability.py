"""Numerical experiment behind the dense step of the resident round kernel (csrc/frx_round_kernel.hpp): an L-BFGS run of one
headline-size candidate on the CPU oracle objective; at every accepted step the direction from the INCREMENTALLY maintained explicit
inverse of R = S^T Y (compact representation) is compared with the two-loop recursion.  Prints the worst relative difference."""
import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
from oracle import binding as ob
cand = sc.make_candidate(0, 64, 16, perturb_id=3)
o = ob.Oracle(cand, sc.ZHANGJIAJIE, qd_intervals=16); o.set_abscissa_mode(False)
x = o.initial_guess(); n=x.size; m=128
f,g = o.objective(x)
S=np.zeros((m,n)); Y=np.zeros((m,n)); ys=np.zeros(m)
Rinv=np.zeros((m,m))   # slot-indexed explicit inverse of R (old->new upper-triangular)
R=np.zeros((m,m)); YtY=np.zeros((m,m))
end=0; k=0
def two_loop(gn,end,bound):
    d=-gn.copy(); j=end; alpha=np.zeros(m)
    for i in range(bound):
        j=(j+m-1)%m; alpha[j]=S[j]@d/ys[j]; d-=alpha[j]*Y[j]
    jn=(end+m-1)%m; d*=ys[jn]/(Y[jn]@Y[jn])
    for i in range(bound):
        beta=Y[j]@d/ys[j]; d+=(alpha[j]-beta)*S[j]; j=(j+1)%m
    return d
d=-g; step=1.0/np.linalg.norm(d)
worst_inv=0; worst_tri=0; t0=time.time(); log=[]
for it in range(1,3501):
    # crude line search: backtracking Armijo + accept (enough for realistic histories)
    dg0=g@d; st=step
    for ls in range(30):
        xn=x+st*d; fn,gn=o.objective(xn)
        if np.isfinite(fn) and fn<=f+1e-4*st*dg0 and abs(gn@d)<=0.9*abs(dg0): break
        if np.isfinite(fn) and fn<=f+1e-4*st*dg0 and gn@d<0.9*dg0: st*=2.1; continue
        st*=0.5
    s=xn-x; y=gn-g
    if s@y<=1e-300: print("curvature fail at",it); break
    jnew=end; bound=min(m,it)
    S[jnew]=s; Y[jnew]=y; ys[jnew]=s@y
    age=(jnew-np.arange(m))%m; valid=age<bound
    c=S@y; e=Y@y; a=S@gn; b=Y@gn
    # update R (older row, newer col), YtY
    for j in np.where(valid)[0]:
        R[j,jnew]=c[j]; YtY[j,jnew]=e[j]; YtY[jnew,j]=e[j]
    # explicit inverse update: drop overwritten slot (row/col jnew), append
    Rinv[jnew,:]=0; Rinv[:,jnew]=0
    older=valid.copy(); older[jnew]=False
    z=Rinv[np.ix_(older,older)]@c[older]
    rho=c[jnew]
    Rinv[older,jnew]=-z/rho; Rinv[jnew,jnew]=1/rho
    gamma=rho/e[jnew]
    # compact direction with explicit inverse (Rinv[i,j] nonzero only for i older-or-equal j)
    w=Rinv@np.where(valid,a,0)
    v=np.where(valid, np.diag(R)*w+gamma*(YtY@w)-gamma*b, 0)
    u=Rinv.T@v
    d_inv=-gamma*gn-S.T@u+gamma*(Y.T@w)
    end=(end+1)%m
    d_ref=two_loop(gn,end,bound)
    err=np.abs(d_inv-d_ref).max()/np.abs(d_ref).max()
    worst_inv=max(worst_inv,err)
    if it%250==0 or err>1e-6: log.append((it,err,f)); print(it,"err %.2e"%err,"f %.6e"%fn, "cond(R)~%.1e"%np.linalg.cond(R[np.ix_(valid,valid)][np.argsort(-age[valid])][:,np.argsort(-age[valid])]) if it%500==0 else "", flush=True)
    x,g,f=xn,gn,fn; d=d_ref; step=1.0
print("worst explicit-inverse direction error", worst_inv, "time",time.time()-t0)
