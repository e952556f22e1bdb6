"""CPU check of the algorithm behind mkws_frontend.hip sqrt48_round (the rounded integer square root of a mel sum < 2^48 from a float
estimate + one exact integer correction) against the integer definition (bits.h Sqrt64 + round to nearest with its saturation points),
with the float estimate forced off by -4..+4 for roots above 4096 (below that sqrtf is exact).  The GPU tests hold the implementation to
the oracle bit for bit; this holds the ALGORITHM to its definition on every boundary class.   python tools/sqrt48_check.py [samples]"""
import sys
import numpy as np, math
rng=np.random.default_rng(0)
def ref(x):
    # bits.h Sqrt64 semantics as in sqrt64_round
    out=np.zeros(len(x),dtype=np.uint64)
    for i,v in enumerate(x.tolist()):
        if v==0: out[i]=0; continue
        r=math.isqrt(v)
        rem=v-r*r
        sat=0xFFFF if (v>>32)==0 else 0xFFFFFFFF
        if rem>r and r!=sat: r+=1
        out[i]=r
    return out
def fast(x, pert_r0=0, pert_k=0):
    x=x.astype(np.uint64)
    hi=(x>>np.uint64(32)).astype(np.uint32); lo=(x&np.uint64(0xFFFFFFFF)).astype(np.uint32)
    xf=np.float32(hi.astype(np.float32))*np.float32(4294967296.0)+lo.astype(np.float32)   # two roundings like cvt+fma (approx)
    r0=np.sqrt(xf.astype(np.float32)).astype(np.float32)
    r0=r0.astype(np.int64); r0=np.minimum(np.where(r0>4096, r0+pert_r0, r0), 0xFFFFFF)
    r0=np.maximum(r0,1)
    r0sq_lo=(r0*r0)&0xFFFFFFFF
    d=((lo.astype(np.int64)-r0sq_lo)&0xFFFFFFFF)
    d=np.where(d>=2**31,d-2**32,d)          # int32
    kf=(d.astype(np.float32)*(np.float32(1.0)/(2*r0).astype(np.float32))).astype(np.float32)
    k=np.floor(kf).astype(np.int64)+pert_k
    def t(k): return k*(2*r0+k)
    for _ in range(1):
        dec=t(k)>d
        inc=(~dec)&(t(k+1)<=d)
        k=np.where(dec,k-1,np.where(inc,k+1,k))
    r=r0+k
    rem=d-t(k)
    m=(x!=0); assert (rem[m]>=0).all() and (rem[m]<=2*r[m]).all(), (x[m][(rem[m]<0)|(rem[m]>2*r[m])][:5])
    sat=np.where(hi==0,0xFFFF,0xFFFFFFFF)
    r=np.where((rem>r)&(r!=sat),r+1,r)
    r=np.where(x==0,0,r)
    return r.astype(np.uint64)
N=int(sys.argv[1]) if len(sys.argv)>1 else 200000
xs=[rng.integers(0,2**48,N,dtype=np.uint64), rng.integers(0,2**20,N//4,dtype=np.uint64), rng.integers(0,2**33,N//4,dtype=np.uint64)]
n=rng.integers(1,2**24,N//3,dtype=np.uint64)
for dlt in (-2,-1,0,1,2):
    v=(n*n).astype(np.int64)+dlt; xs.append(v[v>=0].astype(np.uint64))
# around n(n+1) rounding boundary
for dlt in (-1,0,1,2):
    v=(n*n+n).astype(np.int64)+dlt; xs.append(v.astype(np.uint64))
xs.append(np.array([0,1,2,3,4,0xFFFF*0xFFFF, 0xFFFF*0xFFFF+0xFFFF, 0xFFFF*0xFFFF+0xFFFF+1, 0xFFFFFFFF, 0x100000000, 0x100000001, 2**48-1, (2**24-1)**2, (2**24-1)**2+2**24-1, (2**24-1)**2+2**24],dtype=np.uint64))
x=np.concatenate(xs); x=x[x<2**48]
want=ref(x)
for pr in (-4,-3,-2,-1,0,1,2,3,4):
    for pk in (0,):
        got=fast(x,pr,pk)
        bad=(got!=want)
        print(pr,pk,'mismatches',int(bad.sum()), x[bad][:3], got[bad][:3], want[bad][:3])
        assert not bad.any()
