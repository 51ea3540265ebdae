import cProfile, pstats, sys, os, io
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"graph-hscn_amd")]
import torch, bench
from graph_hscn.config.config import ACT_DICT
from graph_hscn.model.hscn import HSCN
from graph_hscn.loss import criterion
dev=torch.device("cuda:0")
hb_host, graphs, _ = bench.build_hetero_batch("peptides_func", 128, 16, 0, dev)
hb = hb_host.to(dev)
y = hb["local"].y
model = HSCN("GAT","GCN","GCN",ACT_DICT["relu"],9,16,10,3).to(dev)
def step():
    for p in model.parameters(): p.grad=None
    pred = model(hb.x_dict, hb.edge_index_dict, hb)
    loss,_ = criterion("cross_entropy", pred, y)
    loss.backward()
for _ in range(20): step()
torch.cuda.synchronize()
import time
t0=time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter()-t0)/200*1e3)
pr=cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize()
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
