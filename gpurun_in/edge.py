import sys, os
sys.path.insert(0,'/root/repo/exp-trmf-nips16_amd'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, scipy.sparse as smat
import oracle_py as O, trmf
from helpers import make_model, relfro
from trmf import synth
def case(n,T,k,lags,dens,dtype,iters=3,seed=0):
    p = synth.sparse_problem(n=n,T=T,k=k,nlag=len(lags),density=dens,dtype=dtype,seed=seed)
    Y=p['Y']; lag=np.array(lags,dtype=np.uint32)
    m0=synth.initial_model(Y,lag,k,seed=seed)
    W,H,Th=m0.W.copy(),m0.H.copy(),np.asfortranarray(m0.lag_val.copy())
    O.train_port(Y,lag,W,H,Th,synth.HYPER,max_iter=iters)
    model=make_model(m0.W,m0.H,m0.lag_val,lag)
    trmf.train(Y,model,max_iter=iters,missing=True,**synth.HYPER)
    print('n=%d T=%d k=%d lags=%s %s: W %.1e H %.1e Th %.1e' % (n,T,k,lags,np.dtype(dtype).name,relfro(model.W,W),relfro(model.H,H),relfro(model.lag_val,Th)))
case(50,30,1,[1],0.3,np.float64)
case(50,30,1,[1],0.3,np.float32)
case(40,12,2,[1,2,3],0.5,np.float64)
case(64,26,4,[1,24],0.4,np.float32)
case(200,300,8,[1,2,100],0.1,np.float64)
case(200,300,40,[1,7],0.2,np.float32)
case(33,65,7,[2,5],0.3,np.float32)
case(500,90,17,[1,2,3,4],0.1,np.float64)
