#!/usr/bin/env python3
"""Tuning aid (CPU, oracle): connected components of the merge's listed-pair graph at configs[1] -- rows per component, with all listed
partners as edges and with the initially passing ones only (round 6: the component-parallel replay, measured and dropped)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg=load_package(); sc=pkg.scenarios
from oracle import binding as ob
n=16
scen=sc.make_scenario(n,200,30,seed=12345)
o=ob.OracleFilter(n); sc.load_scenario(o,scen)
o.update_map(scen["Z"]); o.importance_weighting()
t2=0.5**2; f=1.5
for i in range(4):
    w,wp,mu,S=o.export_gm(i)[:4]
    N=len(w)
    tr=S[:,0,0]+S[:,1,1]
    rad=np.sqrt(t2*tr*(1+1e-6))
    iS=np.linalg.inv(S)
    def passes(ma,Sa_inv,mj,Sj_inv):
        e=mj-ma
        d1=e@Sa_inv@e
        if d1<=t2: return True
        return e@Sj_inv@e<=t2
    pairs=[];rows=set()
    lists={}
    for a in range(N):
        for j in range(a+1,N):
            e=mu[j]-mu[a]; d2=e@e; T=max(rad[a],rad[j])**2
            if d2<=T or d2<4*T:
                c=d2<=T
                p=c and passes(mu[a],iS[a],mu[j],iS[j])
                lists.setdefault(a,[]).append((j,p))
                if p: rows.add(a)
    # components over edges of ISROW rows
    lab=list(range(N))
    def find(x):
        while lab[x]!=x:
            lab[x]=lab[lab[x]]; x=lab[x]
        return x
    for a in rows:
        for j,p in lists[a]:
            ra,rj=find(a),find(j)
            if ra!=rj: lab[max(ra,rj)]=min(ra,rj)
    comp={}
    for a in sorted(rows): comp.setdefault(find(a),[]).append(a)
    sizes=sorted([len(v) for v in comp.values()],reverse=True)
    # passing-only components
    lab=list(range(N))
    for a in rows:
        for j,p in lists[a]:
            if p:
                ra,rj=find(a),find(j)
                if ra!=rj: lab[max(ra,rj)]=min(ra,rj)
    comp2={}
    for a in sorted(rows): comp2.setdefault(find(a),[]).append(a)
    sizes2=sorted([len(v) for v in comp2.values()],reverse=True)
    npairs=sum(len(v) for v in lists.values())
    print("particle",i,"N",N,"listed pairs",npairs,"rows",len(rows),"components(all listed)",len(comp),"rows/comp",sizes[:12],"| passing-only comps",len(comp2),sizes2[:12])
