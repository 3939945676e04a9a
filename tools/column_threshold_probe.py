import sys, os, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, interpn_amd
dev=torch.device("cuda:0")
for n in (32, 24):
    rng=np.random.default_rng(4); g=np.linspace(-1,1,n); vals=rng.uniform(-1,1,n**4)
    it=interpn_amd.Interpolator.regular("cubic",[n]*4,np.full(4,-1.0),np.full(4,g[1]-g[0]),vals,False,0,np.float64)
    for P in (500_000, 1_000_000, 1_500_000, 2_000_000, 3_000_000, 5_000_000):
        gen=torch.Generator(device=dev); gen.manual_seed(5)
        obs=[torch.rand(P,dtype=torch.float64,device=dev,generator=gen)*2-1 for _ in range(4)]
        out=torch.empty(P,dtype=torch.float64,device=dev)
        row={"n":n,"points":P}
        for name,opts in (("in_place",{"binned":0}),("tiled_sorted",{"binned":1,"column":0}),("column",{"binned":1,"column":1}),("auto",{"binned":-1,"column":-1})):
            for k,v in opts.items(): it.set_option(k,v)
            for _ in range(3): it.eval_tensors(obs,out)
            it.finish()
            ms=[]
            for _ in range(15):
                a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
                a.record(); it.eval_tensors(obs,out); b.record(); it.finish(); ms.append(a.elapsed_time(b))
            row[name]=round(float(np.median(ms)),4)
            if name=="auto": row["auto_kernel"]=it.kernel_name().split("<")[0].replace("interpn::","")
        print(json.dumps(row),flush=True)
    it.close()
