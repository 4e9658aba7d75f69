"""Rows a row plan evaluates per scale on the generator\x27s frames: distinct rows, rounded to granules of 8 / 4 / 2 rows, after next-fit
packing into 32-row tiles (rounds 2-5) and after tight prefix packing (round 6) -- from the oracle\x27s own ball counts.
    python tools/plan_padding_sim.py"""
import sys, os, importlib, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'oracle'))
syn=importlib.import_module('3dssd_amd.synthetic'); cfgs=importlib.import_module('3dssd_amd.configs')
import sa_oracle as O
arch=cfgs.KITTI_3DSSD_ARCH; params=syn.random_backbone_params(arch)
def nextfit(g, GPT):
    # g: granules per ball (in order); ball of <= GPT granules never straddles; returns total granule slots incl. padding
    pos=0
    for x in g:
        x=int(x)
        if x<=GPT:
            ph=pos%GPT
            if ph+x>GPT: pos+=GPT-ph
            pos+=x
        else:
            ph=pos%GPT
            if ph: pos+=GPT-ph
            pos+=x
    return pos
for variant in ('default','rings64'):
    pts=np.stack([syn.frame_of(variant, 900+f, 16384) for f in range(2)])
    trace=[]
    O.sa_backbone(pts, arch, params, cfgs.KITTI_MAX_TRANSLATE_RANGE, trace=trace)
    macs={}
    print(variant)
    for t in trace:
        cnt=t['cnt'].reshape(-1); ns=t['idx'].shape[-1]
        d=np.clip(cnt,1,ns)
        out=[]
        for GR in (8,4,2):
            g=(d+GR-1)//GR
            GPT=32//GR
            nf=nextfit(g,GPT)*GR
            ps=int(np.ceil(g.sum()/GPT))*32
            out.append((GR, int(g.sum()*GR), nf, ps))
        print('  %s s%d ns %2d balls %6d nominal %8d distinct %8d |'%(t['scope'],t['scale'],ns,len(cnt),len(cnt)*ns,d.sum()), ' '.join('GR%d: rounded %d nextfit %d prefix %d'%o for o in out))
