"""Plugin boundary #1: the trainer class nnU-Net discovers BY NAME
(reference utilities/find_class_by_name.py:7-24 walks nnunetv2/training/nnUNetTrainer/**).

Drop this file's class into that tree and ``nnUNetv2_train ... -tr
nnUNetTrainer_MLAgg_2D_dt_MS`` trains the MI355X network with unchanged plans, epoch loop and checkpoints.
nnunetv2 is not importable in the build container (its dependencies are absent offline), so the class
is produced by a factory that receives the base class.

What the class overrides, and why the inherited method cannot stay (B = nnUNetTrainer.py, T = the reference trainer):
  * ``train_step``  B:833-863 runs the network under ``autocast('cuda')`` with a ``GradScaler``.  The MI355X network
                    computes in the precision it was built with (``precision``: "fp32"; "bf16" / "fp16" = 16-bit
                    operands of every Linear and convolution with fp32 sums, i.e. what autocast does to those layers),
                    whatever the ambient autocast state, so the step is ``trainer.train_step`` (zero_grad, forward, loss,
                    backward, clip 12, AdamW).  The ``GradScaler`` of B:152 is kept for "fp16" only (as in the reference,
                    small gradients would otherwise flush to zero in the 16-bit operands) and dropped otherwise; the
                    returned dictionary and the per-step host copy of the loss are the reference's (B:863).
  * ``initialize``  B:193-215 wraps with a plain ``DDP(...)``; here ``trainer.wrap_ddp`` (bucket views, no buffer
                    broadcast) and the loss is rebuilt so that it knows about DDP.
  * ``_build_loss`` T:106-129: the fused Dice + CE deep-supervision loss (K9) with the batch-dice exchange as one
                    all-reduce; the ignore label of partially annotated datasets is handled in K9; region datasets keep the
                    reference's own loss classes.
"""
import os

import torch

from . import evaluation, miopen_tuning, model, trainer

PRECISIONS = ("fp32", "bf16", "fp16")
# The reference loop reads the loss back every step (B:863), so the host cannot run ahead of the device: an eagerly enqueued step then
# costs its ~30 ms of Python / launch time ON TOP of a part of its GPU time (41.6 against 37.7 ms per step at 256 x 256, batch 10:
# profiles/round4_e_bench_with_mfma_roofline.json.log).  After GRAPH_AFTER eager steps on one batch geometry the plugin therefore
# captures the whole step (zero_grad, forward, loss, backward, clip, AdamW: trainer.GraphedTrainStep) and replays it: 38.4 ms with
# the read-back.  Single-process fp32 / bf16 training only; MLAGG_PLUGIN_GRAPH=0 keeps every step eager.
PLUGIN_GRAPH = os.environ.get("MLAGG_PLUGIN_GRAPH", "1") == "1"
GRAPH_AFTER = 3


def make_trainer_class(nnUNetTrainer, variant="B", precision="fp32"):
    if precision not in PRECISIONS:
        raise RuntimeError(f"precision {precision!r}: this build of the MI355X path offers {PRECISIONS}")

    class nnUNetTrainer_MLAgg_2D_dt_MS(nnUNetTrainer):
        def __init__(self, plans, configuration, fold, dataset_json, unpack_dataset=True, device=None):
            super().__init__(plans, configuration, fold, dataset_json, unpack_dataset,
                             device if device is not None else torch.device("cuda"))
            # reference T:52-59
            self.initial_lr = 5e-4
            self.weight_decay = 3e-5
            self.oversample_foreground_percent = 0.33
            self.num_iterations_per_epoch = 250
            self.num_val_iterations_per_epoch = 50
            self.num_epochs = 500
            self.current_epoch = 0
            # B:152 creates a GradScaler for the fp16 autocast of B:848.  fp32 / bf16 do not scale the loss: checkpoints
            # then carry ``grad_scaler_state: None`` exactly as the reference's CPU runs do (B:1018, 1047-1049).
            if precision != "fp16":
                self.grad_scaler = None
            self.mlagg_precision = precision
            self._graphed, self._graph_key, self._graph_eager, self._graph_failed = None, None, 0, None
            # run_training.py:123-125 sets cudnn.benchmark (MIOpen's exhaustive find); here: the committed find-db, which
            # holds the fp32 convolutions of the 256 x 256 step only.  Any other patch size or precision takes MIOpen's
            # immediate-mode choice (no find, naive fallback solvers left available): a find-db miss in FAST mode with the
            # naive solvers switched off can stall or end in "no solver found"
            patch = tuple(int(v) for v in self.configuration_manager.patch_size)
            miopen_tuning.use_tuned_convolutions(enabled=(precision == "fp32" and patch == miopen_tuning.TUNED_PATCH))

        @staticmethod
        def build_network_architecture(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                       enable_deep_supervision=True):
            label_manager = plans_manager.get_label_manager(dataset_json)          # reference T:68
            return model.build_network_architecture(configuration_manager.patch_size, num_input_channels,
                                                    label_manager.num_segmentation_heads, enable_deep_supervision,
                                                    variant, precision)

        def initialize(self):                                                       # reference B:193-215
            ddp = self.is_ddp
            self.is_ddp = False           # keeps the inherited body from wrapping with the plain DDP(...) of B:205-207
            try:
                super().initialize()
            finally:
                self.is_ddp = ddp
            if ddp:
                self.network = trainer.wrap_ddp(self.network, self.device.index if self.device.type == "cuda" else None)
                self.loss = self._build_loss()

        def set_deep_supervision_enabled(self, enabled):
            trainer.set_deep_supervision_enabled(self.network, enabled)             # fixes SURVEY finding 7b

        def _get_deep_supervision_scales(self):                                    # reference T:101-104
            return [[1.0 / 2 ** i] * 2 for i in range(5)]

        def _build_loss(self):                                                      # reference T:106-129
            lm = self.label_manager
            if getattr(lm, "has_regions", False):
                # DC_and_BCE_loss on region targets (T:107-112): not on the MLAgg-UNet benchmark path; the reference's own
                # torch loss classes run unchanged on the device logits
                return super()._build_loss()
            batch_dice, ddp = bool(self.configuration_manager.batch_dice), bool(self.is_ddp)
            ignore = getattr(lm, "ignore_label", None)                              # T:116: partially annotated datasets

            def loss(output, target):
                if not isinstance(output, (list, tuple)):                           # deep supervision off (validation of
                    output, target = [output], [target if torch.is_tensor(target) else target[0]]   # a no-DS network)
                return trainer.deep_supervision_loss(list(output), list(target[:len(output)]), batch_dice, ddp,
                                                     ignore_label=ignore)

            return loss

        def _graph_ok(self):
            """The step may be replayed as a hipGraph: one process, no GradScaler, the fused device loss, a device network."""
            return bool(PLUGIN_GRAPH and self._graph_failed is None and not self.is_ddp and self.grad_scaler is None
                        and self.device.type == "cuda" and not getattr(self.label_manager, "has_regions", False))

        def configure_optimizers(self):                                             # reference T:137-147
            # capturable: learning rate and step counter live on the device, so the same optimizer serves eager and replayed steps
            return trainer.configure_optimizers(self.network, self.initial_lr, self.weight_decay, capturable=self._graph_ok())

        def _to_device(self, batch):                                                # reference B:834-841
            data = batch["data"].to(self.device, non_blocking=True)
            target = batch["target"]
            target = [t.to(self.device, non_blocking=True) for t in target] if isinstance(target, list) else \
                target.to(self.device, non_blocking=True)
            return data, target

        def train_step(self, batch):                                                # reference B:833-863
            data, target = self._to_device(batch)
            if not isinstance(target, list):
                target = [target]
            data, target = data.float(), [t.float() for t in target]
            if self._graph_ok() and isinstance(self.optimizer, trainer.ClipAdamW) and self.optimizer.capturable:
                key = (tuple(data.shape),) + tuple(tuple(t.shape) for t in target)
                if self._graphed is not None and key == self._graph_key:
                    return {"loss": self._graphed(data, target).cpu().numpy()}      # replay + the reference's per-step host copy
                if self._graphed is None:
                    self._graph_eager = self._graph_eager + 1 if key == self._graph_key else 1
                    self._graph_key = key
                    if self._graph_eager > GRAPH_AFTER:
                        try:
                            # warmup=0: the eager steps above were the warm-up; the capture itself executes nothing, so the first
                            # replay IS this step (no batch is trained on twice)
                            self._graphed = trainer.GraphedTrainStep(self.network, self.optimizer, data, target, clip=12.0,
                                                                     warmup=0, loss_fn=self.loss)
                            return {"loss": self._graphed().cpu().numpy()}
                        except Exception as e:                                      # noqa: BLE001  (stay eager, say why once)
                            self._graphed, self._graph_failed = None, repr(e)
                            self.print_to_log_file(f"MLAgg: hipGraph capture of the train step failed ({e!r}); steps stay eager")
            loss = trainer.train_step(self.network, self.optimizer, data, target, clip=12.0, loss_fn=self.loss,
                                      grad_scaler=self.grad_scaler)
            return {"loss": loss.cpu().numpy()}                                     # the reference's per-step host copy

        def validation_step(self, batch):                                           # reference B:880-942
            if getattr(self.label_manager, "has_regions", False):
                return super().validation_step(batch)        # sigmoid regions (B:899-905): the reference's own body
            data, target = self._to_device(batch)
            # B:897 evaluates self.loss; B:917-929 masks the ignore label out of tp / fp / fn
            return evaluation.validation_step(self.network, data, target, self.configuration_manager.batch_dice,
                                              self.is_ddp, getattr(self.label_manager, "ignore_label", None), self.loss)

        def on_validation_epoch_end(self, val_outputs):                             # reference B:944-978
            res = evaluation.validation_epoch_end(val_outputs)
            for key in ("mean_fg_dice", "dice_per_class_or_region", "val_losses"):
                self.logger.log(key, res[key], self.current_epoch)

        def plot_network_architecture(self):
            pass

    return nnUNetTrainer_MLAgg_2D_dt_MS


def make_umamba_enc_ss3d_trainer_class(nnUNetTrainer):
    """``nnUNetTrainerUMambaEnc_SS3D`` (reference variants/mamba/nnUNetTrainerUMambaEnc_SS3D.py:8-31) on the MI355X 3-D network
    (model3d.UMambaEnc; BASELINE configs[3]).  The reference class overrides ``build_network_architecture`` only and inherits the
    base trainer's recipe (SGD momentum 0.99 Nesterov + PolyLR, nnUNetTrainer.py:448-452; DC_and_CE deep-supervision loss,
    :330-352); so does this one, plus what the device network needs from the loop: the fp32 step without autocast / GradScaler
    (B:848-858), the fused K9 loss, ``trainer.wrap_ddp``, and the deep-supervision switch on the unwrapped module."""
    from . import model3d

    class nnUNetTrainerUMambaEnc_SS3D(nnUNetTrainer):
        def __init__(self, plans, configuration, fold, dataset_json, unpack_dataset=True, device=None):
            super().__init__(plans, configuration, fold, dataset_json, unpack_dataset,
                             device if device is not None else torch.device("cuda"))
            self.grad_scaler = None                                                  # fp32 step: no loss scaling (B:152)
            miopen_tuning.use_tuned_convolutions(enabled=False)                      # no committed records for 3-D convolutions

        @staticmethod
        def build_network_architecture(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                       enable_deep_supervision=True):
            cm = configuration_manager                                               # get_umamba_enc_3d_from_plans, S:890-942
            if len(cm.conv_kernel_sizes[0]) != 3:
                raise RuntimeError("nnUNetTrainerUMambaEnc_SS3D on MI355X: 3-D configurations only")
            label_manager = plans_manager.get_label_manager(dataset_json)
            return model3d.build_network_architecture_3d(
                num_input_channels, label_manager.num_segmentation_heads, cm.conv_kernel_sizes, cm.pool_op_kernel_sizes,
                cm.n_conv_per_stage_encoder, cm.n_conv_per_stage_decoder, cm.UNet_base_num_features, cm.unet_max_num_features,
                enable_deep_supervision)

        def initialize(self):                                                        # reference B:193-215
            ddp = self.is_ddp
            self.is_ddp = False
            try:
                super().initialize()
            finally:
                self.is_ddp = ddp
            if ddp:
                self.network = trainer.wrap_ddp(self.network, self.device.index if self.device.type == "cuda" else None)
                self.loss = self._build_loss()

        def set_deep_supervision_enabled(self, enabled):                             # B: self.network.decoder.deep_supervision
            trainer.set_deep_supervision_enabled(self.network, enabled)

        def _build_loss(self):                                                       # reference B:330-352
            lm = self.label_manager
            if getattr(lm, "has_regions", False):
                return super()._build_loss()
            batch_dice, ddp = bool(self.configuration_manager.batch_dice), bool(self.is_ddp)
            ignore = getattr(lm, "ignore_label", None)

            def loss(output, target):
                if not isinstance(output, (list, tuple)):
                    output, target = [output], [target if torch.is_tensor(target) else target[0]]
                return trainer.deep_supervision_loss(list(output), list(target[:len(output)]), batch_dice, ddp, ignore_label=ignore)

            return loss

        def train_step(self, batch):                                                 # reference B:833-863
            data = batch["data"].to(self.device, non_blocking=True)
            target = batch["target"]
            target = [t.to(self.device, non_blocking=True) for t in target] if isinstance(target, list) else \
                [target.to(self.device, non_blocking=True)]
            loss = trainer.train_step(self.network, self.optimizer, data.float(), [t.float() for t in target], clip=12.0,
                                      loss_fn=self.loss)
            return {"loss": loss.cpu().numpy()}

        def plot_network_architecture(self):
            pass

    return nnUNetTrainerUMambaEnc_SS3D
