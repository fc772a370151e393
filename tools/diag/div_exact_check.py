"""phk_div_row (phamers_amd/csrc/phk_common.h) replayed in exact rational arithmetic: x / T from a shared correctly rounded
reciprocal and two fused Newton steps on the quotient equals the IEEE quotient on every case tried (row sums 1 .. 400, around
every power of two up to 2^32, the benchmark lengths, 300 000 random pairs).  The one-step form is counted too: it never
differed here either, but only the two-step form is covered by Markstein's theorem for every input."""
import random, sys
from fractions import Fraction
def fma(a,b,c): return float(Fraction(a)*Fraction(b)+Fraction(c))
def div5(x,T,y):
    q0 = x*y
    e = fma(-q0,T,x)
    q1 = fma(e,y,q0)
    e1 = fma(-q1,T,x)
    return fma(e1,y,q1)
def div3(x,T,y):
    q0 = x*y
    e = fma(-q0,T,x)
    return fma(e,y,q0)
random.seed(1)
bad5=bad3=0; n=0
def cases():
    for T in list(range(1,400))+[2**k+d for k in range(8,32) for d in (-3,-1,0,1,3)]+[9996,4996,9995,49996,499996,10**7,2**32-1,2**32-5]:
        for x in set([0,1,2,3,T//3,T//2,T-1,T, max(T-2,0), T//7+1]+[random.randrange(0,T+1) for _ in range(8)]):
            if 0<=x<=T: yield x,T
    for _ in range(300000):
        T=random.randrange(1,2**32) if random.random()<0.5 else random.randrange(1,200000)
        x=random.randrange(0,T+1) if random.random()<0.7 else random.randrange(0,min(T,3000)+1)
        yield x,T
for x,T in cases():
    xf,Tf=float(x),float(T); y=1.0/Tf
    w=xf/Tf
    n+=1
    if div5(xf,Tf,y)!=w: bad5+=1; print("bad5",x,T) if bad5<5 else None
    if div3(xf,Tf,y)!=w: bad3+=1
print(n,"cases; mismatches 5-op:",bad5," 3-op:",bad3)
