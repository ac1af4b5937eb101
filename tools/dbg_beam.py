import numpy as np, sys
sys.path.insert(0,'/root/repo')
import chiron_amd as ca
from oracle import c_oracle
spec=ca.dna_default_spec(); w=ca.synthetic_weights(spec,seed=1)
rng=np.random.RandomState(3)
B=16
with ca.Engine(spec,w,max_batch=B,segment_len=400,max_beam=256) as eng:
    for T_eff in (20,100,400):
        lg=(rng.randn(B,400,5)*2.3).astype(np.float32)
        sl=np.full(B,T_eff,dtype=np.int32)
        for W in (60,64,65,100,128,129,200,256):
            res=eng.decode(lg,sl,beam_width=W)
            rows,lp=c_oracle.beam(lg,sl,W)
            got=[[] for _ in range(B)]
            for (r,_),v in zip(res.decoded.indices,res.decoded.values): got[r].append(int(v))
            nbad=sum(got[b]!=rows[b] for b in range(B))
            print(T_eff,W,'rows differ',nbad,'max lp diff %.4g'%np.abs(res.log_prob-lp).max())
