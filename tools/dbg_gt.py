import sys, torch
sys.path[:0]=["/root/repo","/root/repo/cugraph-gnn_amd"]
from wholegraph_amd import nn
for (n,F,H) in [(70001,128,4),(70001,128,2),(70001,128,3),(70001,256,4),(20000,128,4),(16384*2+5,128,4)]:
    g=torch.Generator().manual_seed(1)
    agg=(torch.rand((n,H*F),generator=g)-0.5); w=(torch.rand((F,H*64),generator=g)-0.5)*0.2
    got=nn.gat_transform_heads_fused(agg.cuda(),w.cuda(),H).cpu().double()
    ref=torch.einsum("nhf,fhc->nhc",agg.double().view(n,H,F),w.double().view(F,H,64)).reshape(n,H*64)
    err=(got-ref).abs().amax(dim=1)
    bad=(err>1e-4).nonzero().flatten()
    print(n,F,H,"bad rows:",bad.numel(), bad[:8].tolist(), bad[-8:].tolist(), "tiles:", sorted(set((bad//32).tolist()))[:12])
    if bad.numel():
        r=int(bad[0]); cols=((got[r]-ref[r]).abs()>1e-4).nonzero().flatten()
        print("  row",r,"bad cols",cols.numel(),cols[:6].tolist(),cols[-6:].tolist())
