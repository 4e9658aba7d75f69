"""Round 5, VERDICT r4 item 4iii: what ONE aggregation conv1d would lose in a single fp16 MFMA pass (numpy emulation on the
bench weights and the oracle's pooled features), against the split-bf16 three-pass form the library uses.  CPU only.
    python tools/agg_fp16_error.py > profiles/r05_agg_fp16_error.txt"""
import sys, importlib, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sa_oracle as O
cfgs=importlib.import_module('3dssd_amd.configs'); syn=importlib.import_module('3dssd_amd.synthetic')
arch=cfgs.KITTI_3DSSD_ARCH
params=syn.random_backbone_params(arch)
pts=np.stack([syn.frame_of('default',f,16384) for f in range(2)])
tr=[]
O.sa_backbone(pts,arch,params,cfgs.KITTI_MAX_TRANSLATE_RANGE,trace=tr)
by={}
for t in tr: by.setdefault(t['scope'],[]).append(t['pooled'])
for scope,ps in by.items():
    x=np.concatenate(ps,-1).reshape(-1,sum(p.shape[-1] for p in ps)).astype(np.float32)
    w,b=O.fold_conv_bn(params,scope+'/ensemble',True)
    ref=np.maximum(x.astype(np.float64)@w.astype(np.float64)+b,0)
    y32=np.maximum(x@w+b,0)
    x16=x.astype(np.float16).astype(np.float32); w16=w.astype(np.float16).astype(np.float32)
    y16=np.maximum((x16.astype(np.float64)@w16.astype(np.float64)).astype(np.float32)+b,0)
    # bf16x3: hi/lo split
    def bf(v):
        u=v.view(np.uint32); r=((u+0x7FFF+((u>>16)&1))&0xFFFF0000).astype(np.uint32); return r.view(np.float32)
    xh=bf(x.copy()); xl=bf((x-xh).copy()); wh=bf(w.copy()); wl=bf((w-wh).copy())
    y3=np.maximum((xh.astype(np.float64)@wh+xh.astype(np.float64)@wl+xl.astype(np.float64)@wh).astype(np.float32)+b,0)
    d=np.abs(ref).max()
    print(scope,x.shape,w.shape,'max|x|',np.abs(x).max(),'fp32 err',np.abs(y32-ref).max()/d,'fp16 err',np.abs(y16-ref).max()/d,'bf16x3 err',np.abs(y3-ref).max()/d)
