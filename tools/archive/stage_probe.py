import sys; sys.path.insert(0,".")
import numpy as np, torch, interpn_amd, json
dev=torch.device("cuda:0"); n=32; P=10_000_000
rng=np.random.default_rng(4); g=np.linspace(-1,1,n); vals=rng.uniform(-1,1,n**4)
it=interpn_amd.Interpolator.regular("cubic",[n]*4,np.full(4,-1.0),np.full(4,g[1]-g[0]),vals,False,0,np.float64)
gen=torch.Generator(device=dev); gen.manual_seed(5)
obs=[torch.rand(P,dtype=torch.float64,device=dev,generator=gen)*2-1 for _ in range(4)]
out=torch.empty(P,dtype=torch.float64,device=dev)
it.set_option("stage_timing",1)
st=[]
for _ in range(30):
    it.eval_tensors(obs,out); it.finish(); st.append(it.stage_ms())
print({k: round(float(np.median([s[k] for s in st])),4) for k in st[0]})
