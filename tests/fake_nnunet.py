"""TEST INFRASTRUCTURE: a stand-in for the reference's ``nnUNetTrainer`` base class (nnunetv2 cannot be imported in the
build container or on the GPU box: batchgenerators, SimpleITK, ... are absent offline).

Only the members the MLAgg trainer plugin touches exist, and each restates what the reference does, in the same order
(B = mlagg/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py):
  * ``__init__``     B:64-185  is_ddp / local_rank / device selection, plans + configuration + label managers,
                               ``GradScaler`` when the device is a GPU (B:152), ``_set_batch_size_and_oversample``
  * ``initialize``   B:193-215 network -> device, optimizer, plain ``DDP(network, device_ids=[local_rank])``, loss
  * ``train_step``   B:833-863 H2D, zero_grad, autocast forward + loss, scaled backward, unscale, clip 12, step, update
Every inherited hot-path method counts its calls in ``self.base_calls`` so that a test can assert which body ran.
"""
import contextlib
from collections import Counter

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP


class _LabelManager:
    def __init__(self, n_classes, has_regions=False, ignore_label=None):
        self.num_segmentation_heads = n_classes
        self.all_labels = list(range(n_classes))
        self.has_regions, self.ignore_label = has_regions, ignore_label
        self.has_ignore_label = ignore_label is not None


class _ConfigurationManager:
    def __init__(self, cfg):
        self.patch_size = list(cfg["patch_size"])
        self.batch_size = int(cfg["batch_size"])
        self.batch_dice = bool(cfg.get("batch_dice", True))
        self.previous_stage_name = None
        self.data_identifier = cfg.get("data_identifier", "nnUNetPlans_2d")
        # the architecture entries of a plans file (plans_handler.py:94-120); absent from the 2-D MLAgg test plans
        for key in ("conv_kernel_sizes", "pool_op_kernel_sizes", "n_conv_per_stage_encoder", "n_conv_per_stage_decoder",
                    "UNet_base_num_features", "unet_max_num_features"):
            if key in cfg:
                setattr(self, key, cfg[key])


class _PlansManager:
    def __init__(self, plans):
        self.plans = plans
        self.dataset_name, self.plans_name = plans["dataset_name"], plans["plans_name"]

    def get_configuration(self, name):
        return _ConfigurationManager(self.plans["configurations"][name])

    def get_label_manager(self, dataset_json):
        return _LabelManager(len(dataset_json["labels"]), dataset_json.get("regions", False),
                             dataset_json.get("ignore_label"))


class _Logger:
    def __init__(self):
        self.records = []

    def log(self, key, value, epoch):
        self.records.append((key, value, epoch))


def make_plans(patch_size, batch_size, batch_dice=True):
    return {"dataset_name": "Dataset702_AbdomenMR", "plans_name": "nnUNetPlans",
            "configurations": {"2d_bs10": {"patch_size": list(patch_size), "batch_size": batch_size,
                                           "batch_dice": batch_dice}}}


def make_plans_3d(patch_size, batch_size, strides, base=32, cap=320):
    """A 3d_fullres configuration as the experiment planner writes it: 3x3x3 kernels, 2 convolutions per stage, no batch dice."""
    n = len(strides)
    return {"dataset_name": "Dataset703_BTCV", "plans_name": "nnUNetPlans",
            "configurations": {"3d_fullres": {"patch_size": list(patch_size), "batch_size": batch_size, "batch_dice": False,
                                              "conv_kernel_sizes": [[3, 3, 3]] * n, "pool_op_kernel_sizes": [list(s) for s in strides],
                                              "n_conv_per_stage_encoder": [2] * n, "n_conv_per_stage_decoder": [2] * (n - 1),
                                              "UNet_base_num_features": base, "unet_max_num_features": cap,
                                              "data_identifier": "nnUNetPlans_3d_fullres"}}}


def make_dataset_json(n_classes, in_channels=1):
    return {"labels": {str(i): i for i in range(n_classes)}, "channel_names": {str(c): "MR" for c in range(in_channels)}}


class nnUNetTrainer:
    def __init__(self, plans, configuration, fold, dataset_json, unpack_dataset=True, device=torch.device("cuda")):
        self.base_calls = Counter()
        self.is_ddp = dist.is_available() and dist.is_initialized()                 # B:81
        self.local_rank = 0 if not self.is_ddp else dist.get_rank()
        self.device = device
        if self.is_ddp and device.type == "cuda":
            self.device = torch.device(type="cuda", index=self.local_rank)         # B:91
        elif self.device.type == "cuda":
            self.device = torch.device(type="cuda", index=0)                       # B:95
        self.plans_manager = _PlansManager(plans)
        self.configuration_manager = self.plans_manager.get_configuration(configuration)
        self.configuration_name, self.dataset_json, self.fold = configuration, dataset_json, fold
        self.unpack_dataset = unpack_dataset
        self.initial_lr, self.weight_decay = 1e-2, 3e-5                             # B:136-141
        self.oversample_foreground_percent = 0.33
        self.num_iterations_per_epoch, self.num_val_iterations_per_epoch, self.num_epochs = 250, 50, 500
        self.current_epoch = 0
        self.label_manager = self.plans_manager.get_label_manager(dataset_json)
        self.num_input_channels = None
        self.network = None
        self.optimizer = self.lr_scheduler = None
        self.grad_scaler = torch.amp.GradScaler("cuda") if self.device.type == "cuda" else None   # B:152
        self.loss = None
        self.logger = _Logger()
        self.batch_size = self.configuration_manager.batch_size                     # B:283-287 (non-DDP branch)
        self.was_initialized = False

    def print_to_log_file(self, *args, **kwargs):
        pass

    @staticmethod
    def build_network_architecture(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                   enable_deep_supervision=True):
        raise NotImplementedError

    def configure_optimizers(self):                                                 # B:448-452 (PolyLRScheduler restated)
        optimizer = torch.optim.SGD(self.network.parameters(), self.initial_lr, weight_decay=self.weight_decay, momentum=0.99,
                                    nesterov=True)

        class _PolyLR:
            def __init__(self, opt, initial_lr, max_steps, exponent=0.9):
                self.opt, self.initial_lr, self.max_steps, self.exponent, self.ctr = opt, initial_lr, max_steps, exponent, 0

            def step(self, current_step=None):
                if current_step is None or current_step == -1:
                    current_step = self.ctr
                    self.ctr += 1
                for group in self.opt.param_groups:
                    group["lr"] = self.initial_lr * (1 - current_step / self.max_steps) ** self.exponent

        return optimizer, _PolyLR(optimizer, self.initial_lr, self.num_epochs)

    def _get_deep_supervision_scales(self):                                         # B:278-281
        import numpy as np
        return list(list(i) for i in 1 / np.cumprod(np.vstack(self.configuration_manager.pool_op_kernel_sizes), axis=0))[:-1]

    def _build_loss(self):
        """B:330-352 builds DeepSupervisionWrapper(DC_and_CE_loss | DC_and_BCE_loss); the loss classes are not importable
        next to this stand-in, so the base body only records that it was asked."""
        self.base_calls["_build_loss"] += 1
        return "reference-loss-classes"

    def initialize(self):                                                           # B:193-215
        self.base_calls["initialize"] += 1
        if self.was_initialized:
            raise RuntimeError("You have called self.initialize even though the trainer was already initialized.")
        self.num_input_channels = len(self.dataset_json["channel_names"])          # determine_num_input_channels
        self.network = self.build_network_architecture(self.plans_manager, self.dataset_json, self.configuration_manager,
                                                       self.num_input_channels, enable_deep_supervision=True
                                                       ).to(self.device)
        self.optimizer, self.lr_scheduler = self.configure_optimizers()
        if self.is_ddp:
            self.base_calls["plain_ddp_wrap"] += 1
            self.network = DDP(self.network, device_ids=[self.local_rank] if self.device.type == "cuda" else None)
        self.loss = self._build_loss()
        self.was_initialized = True

    def train_step(self, batch):                                                    # B:833-863
        self.base_calls["train_step"] += 1
        data, target = batch["data"], batch["target"]
        data = data.to(self.device, non_blocking=True)
        target = [t.to(self.device, non_blocking=True) for t in target] if isinstance(target, list) else \
            target.to(self.device, non_blocking=True)
        self.optimizer.zero_grad()
        ctx = torch.autocast(self.device.type, enabled=True) if self.device.type == "cuda" else contextlib.nullcontext()
        with ctx:
            output = self.network(data)
            loss = self.loss(output, target)
        if self.grad_scaler is not None:
            self.grad_scaler.scale(loss).backward()
            self.grad_scaler.unscale_(self.optimizer)
            torch.nn.utils.clip_grad_norm_(self.network.parameters(), 12)
            self.grad_scaler.step(self.optimizer)
            self.grad_scaler.update()
        else:
            loss.backward()
            torch.nn.utils.clip_grad_norm_(self.network.parameters(), 12)
            self.optimizer.step()
        return {"loss": loss.detach().cpu().numpy()}
