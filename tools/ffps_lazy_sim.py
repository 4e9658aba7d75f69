"""Host simulation of a LAZY-GREEDY exact F-FPS (round 6 analysis): keep stale upper bounds of the running
minimum, refresh only the top-K candidates against the picks they have not seen, stop when the arg-max is fresh.  Counts rounds per
pick and (candidate, pick) distance evaluations against the full update (511 x 4096) on the oracle's own layer-1 features.
    python tools/ffps_lazy_sim.py"""
import sys, os, importlib, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn=importlib.import_module('3dssd_amd.synthetic'); cfgs=importlib.import_module('3dssd_amd.configs')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')); import sa_oracle as O
arch=cfgs.KITTI_3DSSD_ARCH; params=syn.random_backbone_params(arch)
for variant in ('default','rings64'):
    pts=np.stack([syn.frame_of(variant, 700+f, 16384) for f in range(1)])
    row=arch[0]
    x1,f1,i1=O.pointnet_sa_module_msg(pts[:,:,:3],pts[:,:,3:],row[2],row[3],row[4],row[5],row[6],row[7],row[8],None,row[12],row[13],params,aggregation_channel=row[15])
    F=np.concatenate([x1[0],f1[0]],-1)[None]
    D=O.calc_square_dist(F,F)[0]
    n=D.shape[0]; m=512
    for K in (16,64,256):
        U=np.full(n,1e38,np.float32); last=np.zeros(n,int); picks=[0]
        td=np.full(n,1e38,np.float32); cur=0
        rounds=[]; pairs=0; cands=0; maxun=[]
        for it in range(1,m):
            td=np.minimum(td,D[cur]); nxt=int(np.argmax(td))
            r=0
            while True:
                c=int(np.argmax(U))
                if last[c]==len(picks): break
                top=np.argpartition(-U,K)[:K]
                stale=top[last[top]<len(picks)]
                for c2 in stale:
                    un=picks[last[c2]:]; pairs+=len(un); maxun.append(len(un))
                    U[c2]=min(U[c2], D[c2][un].min()); last[c2]=len(picks)
                cands+=len(stale); r+=1
            assert c==nxt or U[c]==td[nxt]
            rounds.append(r); picks.append(nxt); cur=nxt
        rounds=np.array(rounds); maxun=np.array(maxun)
        print(variant,'K',K,'rounds/pick mean %.2f p90 %d max %d | refreshed cands/pick %.1f | pairs total %d (full %d) | unapplied per refresh mean %.1f p99 %d'%(rounds.mean(),np.quantile(rounds,0.9),rounds.max(),cands/(m-1),pairs,(m-1)*n,maxun.mean(),np.quantile(maxun,0.99)))
