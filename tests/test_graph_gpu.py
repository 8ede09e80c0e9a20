"""GPU: the whole train step captured into one hipGraph (trainer.GraphedTrainStep, `bench.py --graph`) replays the same
optimisation trajectory as eager steps: every C-ABI entry point launches on the capturing stream, takes its workspaces from
torch's graph-private pool and neither allocates nor synchronises (include/mlagg_hip.h)."""
import copy

import pytest
import torch

import mlagg_unet_amd  # noqa: F401
from mlagg_unet_amd import model, trainer
from _trajectory import assert_same_trajectory

pytestmark = pytest.mark.gpu


def test_graph_replay_follows_the_eager_trajectory():
    torch.manual_seed(0)
    net = model.build_network_architecture((64, 64), 1, 14, True, "B").cuda().eval()      # eval: no DropPath draws to align
    twin = copy.deepcopy(net)
    opt, _ = trainer.configure_optimizers(net, capturable=True)
    opt_t, _ = trainer.configure_optimizers(twin)                                          # the eager, launch-argument form of K11
    batches = [trainer.synthetic_batch(2, 1, 64, 64, 14, seed=40 + i, device="cuda") for i in range(3)]
    # the graph's own warm-up steps (3, on its first batch) are part of the trajectory: mirror them on the twin
    graphed = trainer.GraphedTrainStep(net, opt, *batches[0], batch_dice=True, warmup=3)
    for _ in range(3):
        trainer.train_step(twin, opt_t, *batches[0])
    for data, target in batches:
        got = float(graphed(data, target))
        want = float(trainer.train_step(twin, opt_t, data, target))
        assert abs(got - want) < 2e-4 * max(1.0, abs(want)), (got, want)
    assert_same_trajectory(net, twin, steps=6, lr=5e-4)
