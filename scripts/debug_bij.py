import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cleanrl_jl_amd as crl
import oraclelib as O
M32 = 0xffffffff
def rnd(x,mask,bits,k0,k1):
    x=(x+k0)&mask; x=(x*(k1|1))&mask; x^=x>>((bits+1)>>1); x=(x*0x9E3779B1)&mask; x^=x>>((bits+2)//3); return x&mask
def unx(x,s,bits):
    sh=s
    while sh<bits: x^=x>>sh; sh<<=1
    return x
def inv_odd(a):
    x=a
    for i in range(5): x=(x*((2-a*x)&M32))&M32
    return x
def rinv(x,mask,bits,k0,k1inv):
    x=unx(x,(bits+2)//3,bits); x=(x*0x0E8B2F51)&mask; x=unx(x,(bits+1)>>1,bits); x=(x*k1inv)&mask; x=(x-k0)&mask; return x
nt,k=8,128
agent = crl.Agent(crl.PPOConfig(num_envs=nt, num_steps=k), shuffle_mode=1, seed=0x5EED)
h=agent.handle
h.shuffle(11)
perm=h.read(crl._lib.F_PERM)
n=nt*k; bits=1
while (1<<bits)<n: bits+=1
out=(C.c_uint32*4)()
O.lib().orc_philox(0x51, 11, 0, 0xB1D, 0x5EED, 0, out)
key=list(out)
kk=[key[0],key[1],key[2],key[3],key[1]^0xA5A5A5A5,key[0]^0x3C3C3C3C]
mask=(1<<bits)-1
def fwd(p):
    x=p
    while True:
        x=rnd(rnd(rnd(x,mask,bits,kk[0],kk[1]),mask,bits,kk[2],kk[3]),mask,bits,kk[4],kk[5])
        if x<n: return x
inv=[inv_odd(kk[1]|1),inv_odd(kk[3]|1),inv_odd(kk[5]|1)]
def bwd(x):
    while True:
        x=rinv(rinv(rinv(x,mask,bits,kk[4],inv[2]),mask,bits,kk[2],inv[1]),mask,bits,kk[0],inv[0])
        if x<n: return x
pf=np.array([fwd(p) for p in range(n)])
print("forward matches GPU perm:", np.array_equal(pf, perm))
pb=np.array([bwd(x) for x in range(n)])
print("python inverse consistent:", np.array_equal(perm[pb], np.arange(n)))
adv=(3*np.random.default_rng(0).standard_normal((nt,k))+0.7).astype(np.float32)
h.write(crl._lib.F_ADVANTAGE, adv)
h.adv_stats()
M=n//4; flat=adv.ravel(order="F").astype(np.float64)
for mb in range(4):
    st=h.update_minibatch(mb,0.0,apply_update=False)
    sl=flat[perm[mb*M:(mb+1)*M]]
    print(mb, st["adv_mean"], sl.mean(), st["adv_std"], sl.std(ddof=1))
