"""How far the reference arithmetic (fp32, CPU) of the 3-D network is from float64 on the golden case: the spread that bounds what an
elementwise gradient comparison of any fp32 implementation with the golden can show (LeakyReLU sign flips behind every InstanceNorm:
a fraction f of flipped elements moves a gradient by ~sqrt(f) in relative L2).  Measured: 1.5e-5 (last norm) ... 8.5e-3."""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from oracle import mlagg_oracle as O, umamba3d_oracle as U
import test_umamba3d_gpu as T
CFG=T.CFG; n=len(CFG["strides"])
def run(dbl):
    ref = U.build_reference_3d_model(CFG["in_ch"], CFG["n_cls"], U.features_for(n), CFG["strides"]).eval()
    O.deterministic_fill_(ref.state_dict(), seed=21)
    data, target = U.synthetic_batch_3d(CFG["batch"], CFG["in_ch"], CFG["size"], CFG["strides"], CFG["n_cls"], seed=77)
    if dbl: ref=ref.double(); data=data.double()
    l=O.deep_supervision_loss(ref(data), target, batch_dice=False); l.backward()
    return {k:p.grad.double() for k,p in ref.named_parameters() if p.grad is not None}
a=run(False); b=run(True)
for k in ["decoder.stages.4.0.norm2.weight","decoder.stages.4.0.conv2.weight","decoder.stages.4.0.conv1.weight","decoder.stages.3.0.conv1.weight","decoder.stages.2.1.conv1.conv.weight","encoder.stages.0.0.conv1.weight","encoder.stem.0.conv1.weight","encoder.mamba_layers.0.blocks.0.self_attention.in_proj.weight","encoder.mamba_layers.4.blocks.0.mlp.linear1.weight"]:
    print(f"{k:70s} fp32-CPU vs fp64: relL2 {float((a[k]-b[k]).norm()/b[k].norm()):.2e}")
