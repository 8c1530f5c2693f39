"""Two training forwards, ONE backward of the summed losses (head graph in mode fwd): reproduction script."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import rslo_amd  # noqa
from rslo_amd import workload, headgraph, streams
from test_gpu_model import reduced_pair
pool = [list(reduced_pair(i)[:2]) for i in range(4)]
torch.manual_seed(7)
net, _ = workload.build_network()
net.train()
ex = [workload.make_example(net, [pool[0], pool[1]]), workload.make_example(net, [pool[2], pool[3]])]
for k in range(3):
    net.zero_grad(set_to_none=True)
    net(ex[0])["loss"].mean().backward()
torch.cuda.synchronize()
print("graph:", headgraph._STATE[net.odom_predictor].graph is not None, flush=True)
net.zero_grad(set_to_none=True)
l1 = net(ex[0])["loss"].mean()
l2 = net(ex[1])["loss"].mean()
print("forwards done", flush=True)
(l1 + l2).backward()
torch.cuda.synchronize()
print("ok", flush=True)
