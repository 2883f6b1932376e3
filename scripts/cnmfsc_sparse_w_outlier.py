import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import synth, rel_fro
import nmf_toolbox_amd as A
from oracle import nmf_oracle as O
f32 = lambda x: np.asarray(x, np.float64).astype(np.float32).astype(np.float64)
for (m,n,K,T,it) in [(222,100,32,2,3),(222,100,32,2,6),(222,100,32,3,3),(230,100,32,2,3),(222,120,32,2,3),(222,100,20,2,3),(150,90,32,2,4)]:
    V,W0,H0 = synth(m,n,K,T=T)
    cfg = dict(W_init=W0,H_init=H0,maxiter=it,tolerance=1e-300,W_sparsity=0.6,H_fixed=True)
    i0,i1={},{}
    ref=O.cnmfsc(V,K,T,cfg,info=i0); got=A.cnmfsc(V,K,T,cfg,info=i1)
    r32=O.cnmfsc(f32(V/V.max()),K,T,dict(cfg,W_init=f32(W0),H_init=f32(H0)))
    print((m,n,K,T,it),"W %.2e H %.2e cost %.2e | intrinsic W %.2e | tries equal %s %s" % (rel_fro(got[0],ref[0]),rel_fro(got[1],ref[1]),rel_fro(got[2],ref[2]),rel_fro(r32[0],ref[0]), i0.get('triesW')==i1.get('triesW'), i0.get('triesW')), flush=True)
