import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
os.environ["GIK_DBG"] = "4"
from graphik_amd.engine import Template
from graphik_amd import _ffi
from oracle import c_oracle as co
np.set_printoptions(linewidth=220, precision=4)
nm, g = sys.argv[1], int(sys.argv[2])
d = np.load(f"tests/golden/{nm}.npz"); use_lim = bool(int(d["use_limits"])); k = int(d["dim"])
om, pL, pU, D = d["omega"], d["psi_L"], d["psi_U"], d["D_goal"][g]
T = Template.from_matrices(om, pL, pU, k=k, use_limits=use_lim)
r = T.solve(d["Y_init"][g:g+1], T.targets_from_D(D), trace_cap=48); torch.cuda.synchronize()
buf = np.zeros(64*128*4); L = C.CDLL(_ffi.LIB_PATH); L.gik_debug_fetch(buf.ctypes.data_as(C.c_void_p), buf.size); buf = buf.reshape(64,128,4)
its = int(r["iterations"][0]); numit = r["trace"]["numit"][0].cpu().numpy()
# python restatement of the outer loop on top of the oracle's C kernels, dumping inner values
inds = co.limit_inds(om,pL,pU) if use_lim else np.nonzero(np.triu(om))
if use_lim:
    cost=lambda Y: co.lcost(Y,D,om,pL,pU,inds); grad=lambda Y: co.lgrad(Y,D,om,pL,pU,inds); hess=lambda Y,W: co.proj(Y, co.lhess(Y,W,D,om,pL,pU,inds))
else:
    cost=lambda Y: co.jcost(Y,D,inds); grad=lambda Y: co.jgrad(Y,D,inds); hess=lambda Y,W: co.proj(Y, co.jhess(Y,W,D,inds))
x = d["Y_init"][g].copy(); fx=cost(x); gx=grad(x); Delta=(10+k)/8; Dbar=10+k; kit=0; dumps=[]
dot=lambda a,b: float(np.dot(a.ravel(),b.ravel()))
while True:
    eta=np.zeros_like(x); Heta=np.zeros_like(x); rr=gx.copy(); e_Pe=0.; r_r=dot(rr,rr); nr0=np.sqrt(r_r); z_r=r_r; d_Pd=z_r; delta=-rr; e_Pd=0.; model=0.; stop=4; inner=[]
    for j in range(10000):
        Hd=hess(x,delta); d_Hd=dot(delta,Hd); alpha=z_r/d_Hd; e_new=e_Pe+2*alpha*e_Pd+alpha*alpha*d_Pd
        if d_Hd<=0 or e_new>=Delta**2:
            tau=(-e_Pd+np.sqrt(e_Pd*e_Pd+d_Pd*(Delta**2-e_Pe)))/d_Pd; eta=eta+tau*delta; Heta=Heta+tau*Hd; stop=0 if d_Hd<=0 else 1; break
        inner.append((r_r,d_Hd,alpha,model))
        e_Pe=e_new; ne=eta+alpha*delta; nH=Heta+alpha*Hd; nm_=dot(ne,gx)+0.5*dot(ne,nH)
        if nm_>=model: stop=5; break
        eta,Heta,model=ne,nH,nm_; rr=rr+alpha*Hd; r_r=dot(rr,rr)
        if j>=1 and np.sqrt(r_r)<=nr0*min(nr0,0.1): stop=3 if not 0.1<nr0 else 2; break
        zo=z_r; z_r=r_r; beta=z_r/zo; delta=-rr+beta*delta; e_Pd=beta*(e_Pd+alpha*d_Pd); d_Pd=z_r+beta*beta*d_Pd
    dumps.append((j,stop,inner))
    xp=x+eta; fp=cost(xp); rhonum=fx-fp; rhoden=-dot(gx,eta)-0.5*dot(eta,Heta); reg=max(1,abs(fx))*2.220446049250313e-16*1e3; rhonum+=reg; rhoden+=reg
    md=rhoden>=0; rho=rhonum/rhoden
    if rho<0.25 or not md or np.isnan(rho): Delta/=4
    elif rho>0.75 and stop in (0,1): Delta=min(2*Delta,Dbar)
    if md and rho>0.1: x=xp; fx=fp; gx=grad(x)
    kit+=1
    if kit>=3000 or np.linalg.norm(gx)<0.5e-9: break
print("oracle-py its", kit, "gpu its", its)
# first outer iteration where inner counts differ
for kk in range(min(kit, its)):
    if dumps[kk][0] != numit[kk]:
        print("first differing outer iteration", kk, "oracle numit", dumps[kk][0], "stop", dumps[kk][1], "gpu numit", numit[kk])
        oi = np.array(dumps[kk][2]); gi = buf[kk][:numit[kk]+1]
        n = max(len(oi), 0)
        print(" j   r_r(oracle)    r_r(gpu)     d_Hd(oracle)   d_Hd(gpu)    alpha(o)   alpha(g)")
        for j in range(min(len(oi), len(gi), 40)):
            print("%3d  %.6e  %.6e   %.6e  %.6e  %.6e %.6e" % (j, oi[j,0], gi[j,0], oi[j,1], gi[j,1], oi[j,2], gi[j,2]))
        break
