"""LDS bank conflicts of the micro-frontend FFT exchanges (mkws_frontend.hip frame_to_sig) under candidate index swizzles.
Passes per 64-lane 4-byte access = the largest number of lanes on one of the 32 banks (2 = the minimum).  No GPU needed."""
import itertools, numpy as np
lanes=np.arange(64)
def stages():
    S={}
    S['A']=[4*lanes+q for q in range(4)]
    S['B']=[(lanes>>2)*16+(lanes&3)+4*q for q in range(4)]
    S['C']=[(lanes>>4)*64+(lanes&15)+16*q for q in range(4)]
    S['D']=[lanes+64*q for q in range(4)]
    S['P']=[lanes+1, 255-lanes, lanes+65, 191-lanes]   # post-pass: fftbuf[k], fftbuf[256-k], k=lane+1 / lane+65
    return S
S=stages()
def cost(f):
    tot={}
    for name,accs in S.items():
        c=0
        for a in accs:
            b=f(a)%32
            c+=np.bincount(b,minlength=32).max()
        tot[name]=c
    return tot
ident=lambda i:i
print('identity',cost(ident))
best=[]
# XOR swizzles: i ^ (((i>>s1)&m1)<<t1) ^ (((i>>s2)&m2)<<t2)
for s1,m1,t1 in itertools.product(range(2,8),[1,3,7,15,31],range(0,5)):
    f=lambda i,s1=s1,m1=m1,t1=t1: i ^ (((i>>s1)&m1)<<t1)
    # must be a bijection on 0..255
    if len(set(f(np.arange(256)).tolist()))!=256 or f(np.arange(256)).max()>255: continue
    c=cost(f); tot=c['B']*2+c['C']*2+c['D']*2+c['P']+c['A']
    best.append((tot,c,('x1',s1,m1,t1)))
for a,b in itertools.product(range(0,5),range(0,5)):
    f=lambda i,a=a,b=b: i + a*(i>>4) + b*(i>>6)
    c=cost(f); tot=c['B']*2+c['C']*2+c['D']*2+c['P']+c['A']
    best.append((tot,c,('pad',a,b, int(f(np.array([255]))[0])+1)))
best.sort(key=lambda t:t[0])
for t in best[:12]: print(t)
