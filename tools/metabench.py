"""Developer tool: MetadataTrainer step (BASELINE configs[3]) stand-alone, for kernel traces: python tools/metabench.py [steps] [graph 0|1]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd.engine import MetadataTrainer
from deeptreeattention_amd.metadata import metadata_sensor_fusion
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
graph = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
dev = torch.device("cuda", 0)
torch.manual_seed(1)
m = metadata_sensor_fusion(bands=369, sites=23, classes=200, precision="bf16").to(dev).train()
tr = MetadataTrainer(m, lr=1e-4, graph_head=graph)
x = torch.rand(1024, 369, 11, 11, device=dev); site = torch.randint(0, 23, (1024,), device=dev); y = torch.randint(0, 200, (1024,), device=dev)
for _ in range(10): tr.train_step(x, site, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): tr.train_step(x, site, y)
torch.cuda.synchronize(); print("graph", graph, "ms/step %.4f" % ((time.perf_counter() - t0) / steps * 1e3))
