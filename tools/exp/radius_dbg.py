import numpy as np, torch
from tests import oracle_lib as O
from contrastboundary_amd import synthetic as S, tf_ops
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
xyz,_=S.s_room(15000,seed=5); lens=np.int32([6000,9000])
sub=np.concatenate([xyz[:6000:3],xyz[6000::3]]); sl=np.int32([len(xyz[:6000:3]),len(xyz[6000::3])])
r,limit=0.1,26
for name,(q,ql,s,slen) in {"self":(xyz,lens,xyz,lens),"sub->xyz":(sub,sl,xyz,lens),"xyz->sub":(xyz,lens,sub,sl)}.items():
    got=tf_ops.tf_batch_neighbors(dev(q),dev(s),dev(ql),dev(slen),r,limit,exact_shape=False).cpu().numpy()
    ref,counts,mc=O.radius_neighbors(q,s,ql,slen,r,limit)
    bad=np.nonzero((got!=ref).any(1))[0]
    print(name,"bad rows",len(bad),"of",len(q), "first",bad[:10])
    for i in bad[:3]:
        print(" row",i,"cloud",0 if i<ql[0] else 1,"count",counts[i]); print("  got",got[i]); print("  ref",ref[i])
