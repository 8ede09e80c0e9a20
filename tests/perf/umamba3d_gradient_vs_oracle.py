"""Debug aid: per-tensor elementwise gradient deviation of the 3-D product network from the CPU oracle (same weights, inputs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import mlagg_unet_amd
from mlagg_unet_amd import model3d, trainer
from oracle import mlagg_oracle as O, umamba3d_oracle as U
import test_umamba3d_gpu as T
torch.set_num_threads(16)
dev = torch.device("cuda:0")
net = T._net(dev)
CFG = T.CFG
data, target = model3d.synthetic_batch_3d(CFG["batch"], CFG["in_ch"], CFG["size"], CFG["strides"], CFG["n_cls"], seed=77, device=dev)
out = net(data)
loss = trainer.deep_supervision_loss(out, target, batch_dice=False)
loss.backward()
grads = {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None}
n = len(CFG["strides"])
ref = U.build_reference_3d_model(CFG["in_ch"], CFG["n_cls"], U.features_for(n), CFG["strides"]).eval()
O.deterministic_fill_(ref.state_dict(), seed=21)
dbl = "--double" in sys.argv
if dbl:
    ref = ref.double()
d2 = data.cpu().double() if dbl else data.cpu()
o2 = ref(d2)
l2 = O.deep_supervision_loss(o2, [t.cpu() for t in target], batch_dice=False)
l2.backward()
print("loss", float(loss.detach()), float(l2.detach()))
for name, p in ref.named_parameters():
    if p.grad is None or name not in grads: continue
    g, r = grads[name].double(), p.grad.double()
    if float(r.norm()) < 1e-5: continue
    print(f"{name:75s} relL2 {float((g-r).norm()/r.norm()):.2e}")
