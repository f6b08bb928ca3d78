import os, sys
import numpy as np
ROOT="/root/repo"
sys.path[:0]=[os.path.join(ROOT,"tests"),os.path.join(ROOT,"oracle"),os.path.join(ROOT,"mech5845m-wbc-for-legged-manipulator_amd")]
import common, oracle
from wbc_batch import WbcBatch
wx,px=common.models()
B=2048; DT=0.002
cfg=common.config("c3",wx)
d=common.tick_inputs(wx,cfg,B,seed=91)
ref=oracle.tick([wx],[cfg],d,DT,B,nthreads=8)
ok=ref["status"]==0
bt=WbcBatch(wx,B); bt.configure(cfg)
rng=np.random.default_rng(3)
junk=rng.integers(-2**62,2**62,(B,2),dtype=np.int64)
for path,(s3,pre) in {"sim3":(1,1),"general+presolve":(0,1),"general":(0,0)}.items():
    bt.set_option("sim3_kernel",s3); bt.set_option("presolve",pre)
    for name,ws in {"junk":junk,"junk bounds only":np.stack([junk[:,0],0*junk[:,1]],1),"junk rows only":np.stack([0*junk[:,0],junk[:,1]],1)}.items():
        got=bt.tick(dict(d,working_set=ws),DT,want_working_set=True)
        badm=got["status"]!=ref["status"]
        err=np.abs(got["qdot"]-ref["qdot"])[ok&~badm].max()
        print(path,name,"status mismatches",int(badm.sum()),"ref",np.bincount(ref["status"][badm],minlength=4),"got",np.bincount(got["status"][badm],minlength=4),"err %.2e"%err,"iters mean %.1f max %d"%(got["iters"].mean(),got["iters"].max()))
        if badm.any():
            i=np.nonzero(badm)[0][0]
            print("   first bad",i,"ref st",ref["status"][i],"got",got["status"][i],"iters",got["iters"][i],"ws %x %x"%(int(ws[i,0])&(2**64-1),int(ws[i,1])&(2**64-1)))
        e=np.abs(got["qdot"]-ref["qdot"]).max(axis=1); e[~ok|badm]=0
        j=int(e.argmax()); print("   worst err inst",j,"err %.2e"%e[j],"iters",got["iters"][j],"ws %x %x"%(int(ws[j,0])&(2**64-1),int(ws[j,1])&(2**64-1)), "n bad(>1e-5)", int((e>1e-5).sum()))
