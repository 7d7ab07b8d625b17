import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import manticoresearch_amd as m
from oracle import oracle as orc
from test_gpu_parity import to_orc, orc_index_of, kw
def hp(f,p): return (f<<24)|p
cases = []
# doc0: multi-hit both terms; doc1: lone hits in field 0 / field 1; doc2: lone in field 1 both
hits = [(1,0,hp(0,1)),(1,0,hp(1,5)),(1,0,hp(2,9)), (2,0,hp(0,3)),(2,0,hp(1,6)),(2,0,hp(2,20)),
        (1,1,hp(1,4)), (2,1,hp(1,5)),
        (1,2,hp(0,4)),(1,2,hp(1,7)), (2,2,hp(1,8)),
        (1,3,hp(1,2)),(1,3,hp(2,3)), (2,3,hp(1,9)),(2,3,hp(2,4))]
hits.sort()
W=np.array([h[0] for h in hits],np.uint64); R=np.array([h[1] for h in hits],np.uint32); H=np.array([h[2] for h in hits],np.uint32)
hi = m.index_from_hits(W,R,H,n_terms=2,total_docs=4,n_fields=3)
ctx=m.Context(0); seg=m.Segment(ctx,hi); batch=m.Batch(ctx,8); oi=orc_index_of(orc,hi)
for masks in [(0xFFFFFFFF,0xFFFFFFFF),(2,2),(2,0xFFFFFFFF),(0xFFFFFFFF,2),(3,6)]:
    q=m.Query(m.XQNode.AND(kw(m,0,1,masks[0]),kw(m,1,2,masks[1])),ranker=m.SPH_RANK_PROXIMITY)
    g=batch.search(seg,[q])[0]; w=to_orc(orc,q).run(oi)
    print(masks, "dev", dict(zip(g.rowid.tolist(),g.weight.tolist())), "orc", dict(zip(w.rowid.tolist(),w.weight.tolist())))
