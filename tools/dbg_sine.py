import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, flan_amd as fa, oracle_lib as O
sr=48000.0
for (W,hop,dft) in ((512,128,512),(1024,256,1024),(2048,512,2048)):
    x=O.sine(48000)
    ref=O.analyze(x,sr,W,hop,dft)
    for g in (0,1):
        with fa.debug_options(force_generic=g):
            got=fa.analyze(x,sr,W,hop,dft)
        m=ref[...,0].astype(np.float64); df=got[...,1].astype(np.float64)-ref[...,1].astype(np.float64)
        turns=np.rint(df/(sr/hop)); df-=turns*(sr/hop)
        w=m**2
        wr=np.sqrt((w*df**2).sum()/w.sum())
        # contributions per bin
        c=(w*df**2).sum(axis=(0,1))
        top=np.argsort(c)[-3:]
        print(dft,"generic" if g else "tuned","wrms",wr,"top bins",top,c[top]/c.sum(), "max|df| at top bin", np.abs(df[...,top[-1]]).max(), "m there", m[...,top[-1]].mean())
