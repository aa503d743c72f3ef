#!/usr/bin/env python3
"""Search for the slot offsets of the eight shifted tap copies of dsp_fir_f16.hip that make its B-fragment ds_read_b128 conflict-free for every
window alignment e (tap_slot[e][r] in the kernel): lanes are served in four fixed groups of 16, a group is conflict-free when its distinct
addresses fall into different 16-byte slots of the 256-byte bank line."""
import itertools, random
groups=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
groups+= [[l+32 for l in g] for g in groups]
def cost(B):
    tot=0
    for e in range(8):
        for g in groups:
            slots={}
            for l in g:
                j=l&15; h4=l>>4
                r=(8-((j+e)&7))&7
                q=(j+e+r)//8
                addr=(r, h4-q)
                slot=(B[r]+h4-q)%16
                slots.setdefault(slot,set()).add(addr)
            tot+=sum(len(v)-1 for v in slots.values())
    return tot
best=None
# structured: B_r = a*r + c*(r>0)
for a in range(16):
    for c in range(16):
        B=[(a*r + (c if r else 0))%16 for r in range(8)]
        k=cost(B)
        if best is None or k<best[0]: best=(k,B,a,c)
print("structured best",best)
print("current (a=5,c=1):",cost([(5*r+(1 if r else 0))%16 for r in range(8)]), "old (a=3? 816/16=51->3, c=0):",cost([(3*r)%16 for r in range(8)]))
random.seed(1)
bestr=None
for it in range(200000):
    B=[random.randrange(16) for _ in range(8)]
    k=cost(B)
    if bestr is None or k<bestr[0]:
        bestr=(k,B); 
        if k==0: break
print("random best",bestr)

def cost_e(B,e):
    tot=0
    for g in groups:
        slots={}
        for l in g:
            j=l&15; h4=l>>4
            r=(8-((j+e)&7))&7
            q=(j+e+r)//8
            slots.setdefault((B[r]+h4-q)%16,set()).add((r,h4-q))
        tot+=sum(len(v)-1 for v in slots.values())
    return tot
import random
res={}
for e in range(8):
    best=None
    random.seed(e)
    for restart in range(300):
        B=[random.randrange(16) for _ in range(8)]
        k=cost_e(B,e)
        improved=True
        while improved and k>0:
            improved=False
            for r in range(8):
                for v in range(16):
                    if v==B[r]: continue
                    B2=B[:]; B2[r]=v
                    k2=cost_e(B2,e)
                    if k2<k: B,k=B2,k2; improved=True
        if best is None or k<best[0]: best=(k,B)
        if k==0: break
    res[e]=best
    print(e,best)
