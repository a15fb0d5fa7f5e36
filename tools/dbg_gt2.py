import sys, torch
sys.path[:0]=["/root/repo","/root/repo/cugraph-gnn_amd"]
from wholegraph_amd import nn
n,F,H=40000,128,4
agg=torch.zeros((n,H*F)); w=torch.zeros((F,H*64))
for h in range(H):
    agg[:,h*F+0]=torch.arange(n).float()          # k = 0 carries the row index
    agg[:,h*F+5]=1.0                               # k = 5 carries ones
    w[0,h*64+0]=1.0                                # col 0 = row index
    w[5,h*64+1]=float(h+1)                         # col 1 = head + 1
    w[:,h*64+2]=torch.arange(F).float()            # col 2 = sum_k agg*k = 5
got=nn.gat_transform_heads_fused(agg.cuda(),w.cuda(),H).cpu()
for r in [0,1,31,32,16383,16384,16385,16415,16416,32768,32769,39999]:
    print(r,[got[r,h*64:h*64+3].tolist() for h in range(H)])
