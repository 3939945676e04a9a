import sys, os, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, interpn_amd
n=32; P=10_000_000
dev=torch.device("cuda:0")
rng=np.random.default_rng(4); g=np.linspace(-1,1,n); vals=rng.uniform(-1,1,n**4)
it=interpn_amd.Interpolator.regular("cubic",[n]*4,np.full(4,-1.0),np.full(4,g[1]-g[0]),vals,False,0,np.float64)
gen=torch.Generator(device=dev); gen.manual_seed(5)
obs=[torch.rand(P,dtype=torch.float64,device=dev,generator=gen)*2-1 for _ in range(4)]
out=torch.empty(P,dtype=torch.float64,device=dev)
it.set_option("stage_timing",1)
ref=None
for rep in range(2):
    for w in (0,1,2,4,8):
        it.set_option("hist_wgs_per_cu", w)
        for _ in range(3): it.eval_tensors(obs,out); it.finish()
        st=[]; ms=[]
        for _ in range(15):
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record(); it.eval_tensors(obs,out); b.record(); it.finish(); ms.append(a.elapsed_time(b)); st.append(it.stage_ms()["hist"])
        if ref is None: ref=out.clone()
        print(json.dumps({"hist_wgs_per_cu":w,"hist_ms":round(float(np.median(st)),4),"ms":round(float(np.median(ms)),4),"same":bool(torch.equal(out,ref))}),flush=True)
