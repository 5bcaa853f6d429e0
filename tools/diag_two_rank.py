"""Diagnosis aid (GPU box): tests/dist_iter_worker.py with 2 ranks and with 1, after 1, 2 and 3 iterations: which elements of
the state differ, in which records, and which rank owns them.  usage: python tools/diag_two_rank.py"""
import os, sys, subprocess, socket
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import dist_iter_worker
s=socket.socket(); s.bind(("127.0.0.1",0)); port=s.getsockname()[1]; s.close()
worker=ROOT+"/tests/dist_iter_worker.py"
two,one="/tmp/two","/tmp/one"
env=dict(os.environ)
for it in (1,2,3):
    r=subprocess.run([sys.executable,"-m","torch.distributed.run","--nnodes=1","--nproc-per-node=2","--master-addr","127.0.0.1","--master-port",str(port),worker,two,"gloo",str(it)],env=env,capture_output=True,text=True)
    assert r.returncode==0, r.stderr[-3000:]
    r=subprocess.run([sys.executable,worker,one,"gloo",str(it)],env=env,capture_output=True,text=True)
    assert r.returncode==0, r.stderr[-3000:]
    z0,z1,z=np.load(two+"_rank0.npz"),np.load(two+"_rank1.npz"),np.load(one+"_rank0.npz")
    ped=dist_iter_worker.make_ped()
    own0=set(z0["owned"].tolist())
    print("iter",it,"block",z0["block"],z1["block"],"hits",z0["hits"].tolist(),z["hits"].tolist(),"sf",z0["scalefactor"],z["scalefactor"])
    for k in ("allele","sure","hw"):
        d=np.argwhere(~np.isclose(z0[k].astype(float),z[k].astype(float),rtol=1e-9,atol=1e-12))
        recs=sorted(set(d[:,0].tolist()))
        print("  ",k,len(d),"diffs in records",recs)
        for idx in d[:8]:
            idx=tuple(idx); print("      ",idx,z0[k][idx],z[k][idx], "owned0" if idx[0] in own0 else "owned1", "par",ped.par[idx[0]].tolist(), "gen", ped.gen[idx[0]])
