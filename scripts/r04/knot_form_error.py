"""Where the knot form of the MINCO map loses digits against the reference's banded LU: raw and duration-normalised coefficient errors
of the host emulation of frx_minco.hpp (tests/hostcheck) against the CPU oracle, per case, and both against a float128-refined solution of A c = b.
CPU only:  python scripts/r04/knot_form_error.py > profiles/r04_knot_form_error.txt"""
import sys, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from frx_import import frx
from fast_racing_amd import scenario as sc
from oracle import binding as ob
H = C.CDLL('/root/repo/tests/hostcheck/libhostcheck.so')
dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
H.hostcheck_minco_forward.argtypes = [C.c_int, dp, dp, dp, dp, dp, C.c_void_p, C.c_void_p]
for (B,N,gates,kappa,obst) in [(3,32,8,8,False),(2,64,16,16,True),(2,8,2,48,True),(1,100,25,8,False),(1,128,32,8,False)]:
    cands = sc.make_batch(0,B,N,gates,obstacles=obst)
    for c in cands:
        o = ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=kappa)
        x0=o.initial_guess()
        for it in (0,15,60,400):
            x = x0 if it==0 else o.optimize(1e-6,max_iterations=it,x0=x0)["x"]
            T,P,Cf = o.forward(x)
            q = np.ascontiguousarray(P.reshape(-1)) if P.shape[1]==3 else np.ascontiguousarray(P.T.reshape(-1))
            head = np.ascontiguousarray(c.ini_state.T.reshape(-1)); tail=np.ascontiguousarray(c.fin_state.T.reshape(-1))
            Ck = np.zeros(18*N)
            H.hostcheck_minco_forward(N, T, q, head, tail, Ck, None, None)
            Ck = Ck.reshape(-1,3)
            err = np.abs(Ck-Cf)
            relraw = err.max()/np.abs(Cf).max()
            i = np.unravel_index(err.argmax(), err.shape)[0]
            piece, power = divmod(i,6)
            # normalised: c_k h^k
            hk = np.repeat(T,6)**np.tile(np.arange(6),N)
            en = (err*hk[:,None]); cn=(np.abs(Cf)*hk[:,None])
            # per piece normalised error relative to the piece's largest normalised coefficient beyond the constant (shape) and to the global position scale
            relnorm = en.max()/cn.max()
            per_piece = (en.reshape(N,6,3).max(axis=(1,2)) / np.maximum(cn.reshape(N,6,3)[:,1:,:].max(axis=(1,2)),1e-300)).max()
            print(f"N={N} it={it:3d} raw {relraw:.1e} at piece {piece} (h={T[piece]:.4f}, min h {T.min():.4f}) power {power} | normalised/global {relnorm:.1e} | normalised/per-piece-shape {per_piece:.1e}")

print("---- against an extended-precision solution of A c = b (iterative refinement, residuals in float128) ----")
cands = sc.make_batch(0,2,64,16,obstacles=True); c=cands[1]; N=64
o = ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=16); x0=o.initial_guess()
for it in (0,60,400):
    x = x0 if it==0 else o.optimize(1e-6,max_iterations=it,x0=x0)["x"]
    T,P,Cf = o.forward(x)
    A = o.dense_A(T)
    head = np.ascontiguousarray(c.ini_state.T.reshape(-1)); tail=np.ascontiguousarray(c.fin_state.T.reshape(-1))
    b = np.zeros((6*N,3))
    b[0]=c.ini_state[:,0]; b[1]=c.ini_state[:,1]; b[2]=c.ini_state[:,2]
    for i in range(N-1): b[6*i+5]=P[i]
    b[6*N-3]=c.fin_state[:,0]; b[6*N-2]=c.fin_state[:,1]; b[6*N-1]=c.fin_state[:,2]
    xs = np.linalg.solve(A,b).astype(np.float128)
    AL = A.astype(np.float128); bL=b.astype(np.float128)
    for k in range(6):
        r = (bL - AL@xs)
        xs = xs + np.linalg.solve(A, r.astype(np.float64)).astype(np.float128)
    Cex = xs.astype(np.float64)
    Ck = np.zeros(18*N); q=np.ascontiguousarray(P.reshape(-1))
    H.hostcheck_minco_forward(N, T, q, head, tail, Ck, None, None); Ck=Ck.reshape(-1,3)
    m = np.abs(Cex).max()
    print(f"it={it}: residual of the refined solution {np.abs((bL-AL@xs)).max():.1e}; oracle (banded LU) vs exact {np.abs(Cf-Cex).max()/m:.1e}; knot form vs exact {np.abs(Ck-Cex).max()/m:.1e}; knot vs oracle {np.abs(Ck-Cf).max()/m:.1e}")
