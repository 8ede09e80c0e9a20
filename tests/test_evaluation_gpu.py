"""End-to-end DSC on the MI355X: product network -> sliding-window inference -> argmax ->
per-organ DSC, against the same chain on the CPU oracle network; and the in-loop validation step's hard counts."""
import numpy as np
import pytest
import torch

import mlagg_unet_amd  # noqa: F401
from mlagg_unet_amd import evaluation as EV
from mlagg_unet_amd import inference as PI
from mlagg_unet_amd import model as PM
from oracle import evaluation_oracle as EO
from oracle import inference_oracle as IO
from oracle import mlagg_oracle as O

gpu = pytest.mark.gpu


@gpu
def test_segmentation_dsc_matches_oracle_chain():
    tile, ncls = (64, 64), 14
    net = PM.build_network_architecture(tile, 1, ncls, False, "B")
    O.deterministic_fill_(net.state_dict())
    ref_net = O.build_reference_config_model(tile, 1, ncls, deep_supervision=False, variant="B")
    ref_net.load_state_dict(net.state_dict())
    net = net.to("cuda:0")
    g = torch.Generator().manual_seed(8)
    image = torch.rand(1, 3, 96, 72, generator=g)
    gt = torch.round(torch.rand(3, 96, 72, generator=g) * (ncls - 1)).long()
    got = PI.predict_sliding_window_return_logits(net, image, ncls, tile, mirror_axes=(0, 1), tile_batch=4)
    want = IO.predict_sliding_window(ref_net, image, ncls, tile, mirror_axes=(0, 1), accum_dtype=torch.float32)
    seg_dev = got.argmax(0)                                        # (D, X, Y) on the device
    seg_ref = want.argmax(0)
    agree = float((seg_dev.cpu() == seg_ref).float().mean())
    assert agree > 0.999                                           # ties within the 1e-3 logit tolerance only
    # per-organ DSC of the device segmentation, computed ON the device, vs the script restatement on the host
    vol = lambda t: t.permute(1, 2, 0)                             # [x, y, z] as the evaluation script indexes
    dsc_dev = EV.abdomen_case_dsc(vol(gt).to("cuda:0"), vol(seg_dev))
    dsc_host = EO.abdomen_case_dsc(vol(gt).numpy(), vol(seg_dev.cpu()).numpy())
    assert np.array_equal(np.asarray(list(dsc_dev.values()), dtype=float), np.asarray(dsc_host), equal_nan=True)
    dsc_ref = EO.abdomen_case_dsc(vol(gt).numpy(), vol(seg_ref).numpy())
    assert np.nanmax(np.abs(np.asarray(list(dsc_dev.values()), dtype=float) - np.asarray(dsc_ref))) < 5e-3
    # the product segmentation scored against the oracle's: mean DSC ~ 1
    same = EV.abdomen_case_dsc(vol(seg_ref).to("cuda:0"), vol(seg_dev))
    present = [v for k, v in same.items() if not np.isnan(v)]
    assert min(present) > 0.99


@gpu
def test_validation_step_counts_on_device():
    tile, ncls = (64, 64), 5
    net = PM.build_network_architecture(tile, 1, ncls, True, "B")
    O.deterministic_fill_(net.state_dict())
    ref_net = O.build_reference_config_model(tile, 1, ncls, deep_supervision=True, variant="B")
    ref_net.load_state_dict(net.state_dict())
    net, ref_net = net.to("cuda:0").eval(), ref_net.eval()
    data, target = O.synthetic_batch(2, 1, 64, 64, ncls, seed=3)
    out = EV.validation_step(net, data.to("cuda:0"), [t.to("cuda:0") for t in target])
    assert out["tp_hard"].is_cuda and out["tp_hard"].dtype == torch.int64
    with torch.no_grad():
        o = ref_net(data)
    tp, fp, fn = EO.hard_tp_fp_fn(o[0], target[0])
    assert abs(float(out["loss"]) - float(O.deep_supervision_loss(o, target))) < 1e-4
    n = data.shape[0] * 64 * 64
    for a, b in ((out["tp_hard"], tp), (out["fp_hard"], fp), (out["fn_hard"], fn)):
        assert np.abs(a.cpu().numpy() - b).sum() <= 1e-3 * n          # argmax flips only at near-ties
    res = EV.validation_epoch_end([out])
    assert 0 <= res["mean_fg_dice"] <= 1 and len(res["dice_per_class_or_region"]) == ncls - 1
