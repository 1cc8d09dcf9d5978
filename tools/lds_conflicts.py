#!/usr/bin/env python3
"""Model of the LDS bank conflicts of nttl.hpp's limb passes under gfx950's per-instruction banking (the guide's LDS table:
ds_read_b64 = 2 groups of 32 lanes, bank (a/4) mod 64; ds_write_b64 = 4 groups of 16 lanes, bank (a/4) mod 32; every extra distinct
address on a busy bank of a group costs one more LDS cycle).  For every tile shape (LOG_R, LOG_C) it walks the rounds' read / write
patterns and the store phase exactly as limb_round / ntt_limbpass_kernel index the tile, under a candidate word map f:

    python tools/lds_conflicts.py            pad_idx (ntt.hpp), identity, and nttl.hpp's swz: LDS cycles / conflict-free cycles
    python tools/lds_conflicts.py family     searches the one- and two-term XOR maps (how swz was picked)
"""
import itertools, sys
TILE_LOG=12; NT=512
def n_rounds(l): return (l+2)//3
def round_bits(l,r):
    rem=l; nr=n_rounds(l); part=0
    for q in range(r+1):
        part=(rem+(nr-q)-1)//(nr-q); rem-=part
    return part
def round_log_rb(l,r):
    rb=l
    for q in range(r): rb-=round_bits(l,q)
    return rb
def pad(i): return i+(i>>4)
def cost(addrs, kind):
    # addrs: list of 64 word addresses (8-byte words)
    tot=0; ideal=0
    if kind=='r':
        groups=[range(0,32),range(32,64)]; mod=32
    else:
        groups=[range(g*16,g*16+16) for g in range(4)]; mod=16
    for g in groups:
        banks={}
        for l in g:
            banks.setdefault(addrs[l]%mod,set()).add(addrs[l])
        tot+=max(len(v) for v in banks.values()); ideal+=1
    return tot,ideal
def sim(LOG_R,LOG_C,f,verbose=False):
    res=[]
    C=1<<LOG_C
    nr=n_rounds(LOG_R)
    total=[0,0]
    for RI in range(nr):
        P=round_bits(LOG_R,RI); LOG_RB=round_log_rb(LOG_R,RI); S_LOG=LOG_RB-P
        UPT=(1<<(TILE_LOG-P))//NT; UW=512>>P
        rc=[0,0]; wc=[0,0]
        for wave in range(8):
            for uu in range(UPT):
                idx=[]
                for lane in range(64):
                    u=wave*UW+lane+64*uu
                    c=u&(C-1); rest=u>>LOG_C; lo=rest&((1<<S_LOG)-1); hi=rest>>S_LOG
                    i0=(hi<<LOG_RB)+lo
                    idx.append((i0,c))
                for q in range(1<<P):
                    a=[f((((i0+(q<<S_LOG))<<LOG_C)+c)) for (i0,c) in idx]
                    if RI>0:
                        t,i=cost(a,'r'); rc[0]+=t; rc[1]+=i
                    last = S_LOG==0
                    if not last or (LOG_C==0 or P==3):
                        t,i=cost(a,'w'); wc[0]+=t; wc[1]+=i
        res.append((RI,P,S_LOG,rc,wc))
        total[0]+=rc[0]+wc[0]; total[1]+=rc[1]+wc[1]
    # store phase
    LAST_P=round_bits(LOG_R,nr-1)
    sc=[0,0]
    if LOG_C==0 or LAST_P==3:
        for wave in range(8):
            for j in range(8):
                a=[f(wave*512+lane+64*j) for lane in range(64)]
                t,i=cost(a,'r'); sc[0]+=t; sc[1]+=i
    total[0]+=sc[0]; total[1]+=sc[1]
    if verbose:
        for r in res: print("  round",r)
        print("  store",sc)
    return total
def bit(i,k): return (i>>k)&1
def swz(i):   # nttl.hpp
    return i ^ (((i >> 2) ^ (i >> 3)) & 31)
def swz1(i):  # a five-term map with no conflict at all on the C3 shapes; costs more address arithmetic
    b0=bit(i,0)^bit(i,4); b1=bit(i,1)^bit(i,5); b2=bit(i,2)^bit(i,6); b3=bit(i,3)^bit(i,6); b4=bit(i,4)^bit(i,6)^bit(i,7)
    return (i&~31)|b0|(b1<<1)|(b2<<2)|(b3<<3)|(b4<<4)
if __name__=="__main__":
    shapes=[(12,0),(10,2),(9,3),(8,4),(7,5),(6,6),(5,7),(4,8)]
    for name,f in (("pad_idx (ntt.hpp, rounds 1-4)",pad),("identity",lambda i:i),("swz (nttl.hpp)",swz),("swz1 (five terms)",swz1)):
        print(name)
        for s in shapes:
            t=sim(*s,f)
            print("  ",s,t, "%.3f"%(t[0]/t[1]))

def family():
    shapes=[(12,0),(10,2),(9,3),(8,4),(7,5),(6,6),(5,7),(4,8)]
    out=[]
    for s1 in range(1,9):
        for m1 in (7,15,31):
            f=lambda i,s1=s1,m1=m1: i ^ ((i>>s1)&m1)
            tot=[sim(*s,f) for s in shapes]
            out.append((sum(t[0] for t in tot)/sum(t[1] for t in tot), "x1 s=%d m=%d"%(s1,m1), [round(t[0]/t[1],2) for t in tot]))
    for s1 in range(1,9):
        for m1 in (7,15,31):
            for s2 in range(s1+1,10):
                for m2 in (7,15,31,8,16,24):
                    f=lambda i,s1=s1,m1=m1,s2=s2,m2=m2: i ^ ((i>>s1)&m1) ^ ((i>>s2)&m2)
                    tot=[sim(*s,f) for s in shapes]
                    out.append((sum(t[0] for t in tot)/sum(t[1] for t in tot), "x2 s=%d m=%d s2=%d m2=%d"%(s1,m1,s2,m2), [round(t[0]/t[1],2) for t in tot]))
    out.sort()
    for o in out[:25]: print(o)
if len(sys.argv)>1 and sys.argv[1]=="family": family()
