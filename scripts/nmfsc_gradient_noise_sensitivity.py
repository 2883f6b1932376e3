"""How far does nmfsc.m's FLOAT64 algorithm move when (a) its state is rounded to fp32 after every accepted step, (b) its two gradients carry fp32-MFMA-sized noise
(3e-7 of their RMS)?  Both line searches active (W_sparsity 0.4, H_sparsity 0.6), 512 x 2048, K = 64, 60 iterations, a transcription of nmfsc.m:141-245 with hooks
(oracle.projfunc for the projection).  CPU only, ~20 s.  Result (round 6): state rounding 3e-7 / 4e-7 on W / H; gradient noise 4.5e-5 / 5.6e-5 with identical try
counts and the cost at 2e-10 -- the factors drift along a flat direction of the objective, x150 what the gradient carries.  This is why
tests/test_gpu_fullsize_oracle.py holds the both-searches case to 5e-5 on W / H while every discrete decision and the cost are held exactly."""
import sys, numpy as np, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import synth
from oracle.nmf_oracle import projfunc
def run(V,W,H,sW,sH,iters,rstate=None,rgrad=None):
    V=V/V.max(); m,n=V.shape; K=W.shape[1]
    L1a=np.sqrt(m)-(np.sqrt(m)-1)*sW; L1s=np.sqrt(n)-(np.sqrt(n)-1)*sH
    W=W.copy(); H=H.copy()
    for k in range(K): W[:,k]=projfunc(W[:,k],L1a,1.0,True)[0]
    for k in range(K): H[k,:]=projfunc(H[k,:],L1s,1.0,True)[0]
    if rstate: W=rstate(W); H=rstate(H)
    sWs=sHs=1.0; tH=[];tW=[]
    Vh=W@H; cost=[0.5*np.sum((V-Vh)**2)]
    for it in range(iters):
        dH=W.T@(Vh-V)
        if rgrad: dH=rgrad(dH)
        beg=cost[-1]; t=0
        while True:
            t+=1; Hn=H-sHs*dH
            for k in range(K): Hn[k,:]=projfunc(Hn[k,:],L1s,1.0,True)[0]
            if rstate: Hn=rstate(Hn)
            Vh=W@Hn; no=0.5*np.sum((V-Vh)**2)
            if no<=beg: break
            sHs/=2
        tH.append(t); sHs*=1.2; H=Hn
        Vh=W@H; beg=0.5*np.sum((V-Vh)**2)
        dW=(Vh-V)@H.T
        if rgrad: dW=rgrad(dW)
        t=0
        while True:
            t+=1; Wn=W-sWs*dW
            for k in range(K): Wn[:,k]=projfunc(Wn[:,k],L1a,1.0,True)[0]
            if rstate: Wn=rstate(Wn)
            Vh=Wn@H; no=0.5*np.sum((V-Vh)**2)
            if no<=beg: break
            sWs/=2
        tW.append(t); sWs*=1.2; W=Wn
        Vh=W@H; cost.append(0.5*np.sum((V-Vh)**2))
    return W,H,np.array(cost),tH,tW
m,n,K=512,2048,64
V,W0,H0=synth(m,n,K)
t0=time.time()
ref=run(V,W0,H0,0.4,0.6,60); print("ref",time.time()-t0)
rel=lambda a,b: np.linalg.norm(a-b)/np.linalg.norm(b)
f32=lambda x: x.astype(np.float32).astype(np.float64)
rs=np.random.RandomState(5)
gn=lambda g: g+1e-7*np.abs(g).max()*rs.standard_normal(g.shape)*0+ g*0 + 3e-7*np.linalg.norm(g)/np.sqrt(g.size)*rs.standard_normal(g.shape)
for name,kw in (("state fp32",dict(rstate=f32)),("grad noise 3e-7*rms",dict(rgrad=gn)),("both",dict(rstate=f32,rgrad=gn))):
    r=run(V,W0,H0,0.4,0.6,60,**kw)
    print(name,"W %.2e H %.2e cost %.2e tries same %s"%(rel(r[0],ref[0]),rel(r[1],ref[1]),rel(r[2],ref[2]),r[3]==ref[3] and r[4]==ref[4]))
