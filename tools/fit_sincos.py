import numpy as np
from numpy.polynomial import chebyshev as Ch
np.set_printoptions(precision=17)
f32=np.float32
def hexf(x): return float(f32(x)).hex()
pio2=np.pi/2
c1=f32(pio2); c2=f32(pio2-float(c1)); c3=f32(pio2-float(c1)-float(c2))
print('c1',hexf(c1),'c2',hexf(c2),'c3',hexf(c3), '2/pi', hexf(2/np.pi))
# fit on [-pi/4,pi/4] using weighted least squares on Chebyshev nodes in u=r^2
a=np.pi/4*1.0001
N=4000
r=a*np.cos(np.pi*(np.arange(N)+0.5)/N)
r=r[r>1e-6]
u=r*r
# sin: (sin(r)-r)/r^3 = P(u), deg 3
ys=(np.sin(r)-r)/r**3
def minimax_fit(u,y,deg,w=None,iters=30):
    # iteratively reweighted LS approximating minimax (Lawson)
    w=np.ones_like(u) if w is None else w
    lw=np.ones_like(u)
    for _ in range(iters):
        V=np.vander(u,deg+1,increasing=True)
        W=np.sqrt(lw)[:,None]
        coef=np.linalg.lstsq(V*W,(y*np.sqrt(lw)),rcond=None)[0]
        err=np.abs(V@coef-y)*w
        lw=lw*(err/err.max()+1e-3); lw/=lw.sum()
    return coef
ps=minimax_fit(u,ys,3, w=r**3/np.abs(np.sin(r)))
print('sin coeffs',[hexf(c) for c in ps], ps)
yc=(np.cos(r)-1+u/2)/u**2
pc=minimax_fit(u,yc,2, w=u**2/np.abs(np.cos(r)))
print('cos coeffs(deg2)',[hexf(c) for c in pc], pc)
pc3=minimax_fit(u,yc,3, w=u**2/np.abs(np.cos(r)))
print('cos coeffs(deg3)',[hexf(c) for c in pc3], pc3)
# evaluate in float32 emulation
def fma(a,b,c): return f32(np.float64(a)*np.float64(b)+np.float64(c))
def sincos(x, pcs, pcc):
    x=f32(x)
    k=np.rint(f32(x*f32(2/np.pi))).astype(f32)
    r=fma(-k,c1,x); r=fma(-k,c2,r); r=fma(-k,c3,r)
    r2=f32(r*r)
    sp=f32(pcs[-1])
    for c in pcs[-2::-1]: sp=fma(sp,r2,f32(c))
    sr=fma(f32(r*r2),sp,r)
    cp=f32(pcc[-1])
    for c in pcc[-2::-1]: cp=fma(cp,r2,f32(c))
    cp=fma(cp,r2,f32(-0.5))
    cr=fma(cp,r2,f32(1.0))
    q=k.astype(np.int64)
    ss=np.where(q&1,cr,sr); cc=np.where(q&1,sr,cr)
    s=np.where(q&2,-ss,ss); c=np.where((q+1)&2,-cc,cc)
    return s,c
rng=np.random.default_rng(0)
for lo,hi in [(0,6.3),(-20,20),(-8000,8000)]:
    x=rng.uniform(lo,hi,2000000).astype(f32)
    for name,pcc in (('deg2',pc),('deg3',pc3)):
        s,c=sincos(x,ps,pcc)
        xs=x.astype(np.float64)
        es=np.abs(s.astype(np.float64)-np.sin(xs)); ec=np.abs(c.astype(np.float64)-np.cos(xs))
        ulp_s=es/np.spacing(np.abs(np.sin(xs)).astype(f32)).astype(np.float64)
        ulp_c=ec/np.spacing(np.abs(np.cos(xs)).astype(f32)).astype(np.float64)
        print(lo,hi,name,'max abs err sin %.3e cos %.3e  max ulp sin %.2f cos %.2f'%(es.max(),ec.max(),ulp_s.max(),ulp_c.max()))
