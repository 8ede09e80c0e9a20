"""Plugin boundary #1: the trainer class nnU-Net discovers BY NAME
(reference utilities/find_class_by_name.py:7-24 walks nnunetv2/training/nnUNetTrainer/**).

Drop this file's class into that tree and ``nnUNetv2_train ... -tr
nnUNetTrainer_MLAgg_2D_dt_MS`` trains the MI355X network with unchanged plans, loop, checkpoints.
nnunetv2 is not importable in the build container (its dependencies are absent offline), so the class
is produced by a factory that receives the base class.
"""
from . import evaluation, miopen_tuning, model, trainer


def make_trainer_class(nnUNetTrainer, variant="B"):
    class nnUNetTrainer_MLAgg_2D_dt_MS(nnUNetTrainer):
        def __init__(self, plans, configuration, fold, dataset_json, unpack_dataset=True, device=None):
            import torch
            super().__init__(plans, configuration, fold, dataset_json, unpack_dataset,
                             device if device is not None else torch.device("cuda"))
            # reference T:52-59
            self.initial_lr = 5e-4
            self.weight_decay = 3e-5
            self.oversample_foreground_percent = 0.33
            self.num_iterations_per_epoch = 250
            self.num_val_iterations_per_epoch = 50
            self.num_epochs = 500
            # run_training.py:123-125 sets cudnn.benchmark (MIOpen's exhaustive find); here: the committed find-db
            miopen_tuning.use_tuned_convolutions()

        @staticmethod
        def build_network_architecture(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                       enable_deep_supervision=True):
            label_manager = plans_manager.get_label_manager(dataset_json)          # reference T:68
            return model.build_network_architecture(configuration_manager.patch_size, num_input_channels,
                                                    label_manager.num_segmentation_heads, enable_deep_supervision,
                                                    variant)

        def set_deep_supervision_enabled(self, enabled):
            trainer.set_deep_supervision_enabled(self.network, enabled)             # fixes SURVEY finding 7b

        def _get_deep_supervision_scales(self):                                    # reference T:101-104
            return [[1.0 / 2 ** i] * 2 for i in range(5)]

        def configure_optimizers(self):                                             # reference T:137-147
            return trainer.configure_optimizers(self.network, self.initial_lr, self.weight_decay)

        def validation_step(self, batch):                                           # reference B:880-942
            data = batch["data"].to(self.device, non_blocking=True)
            target = batch["target"]
            target = [t.to(self.device, non_blocking=True) for t in target] if isinstance(target, list) else \
                target.to(self.device, non_blocking=True)
            return evaluation.validation_step(self.network, data, target, self.configuration_manager.batch_dice,
                                              self.is_ddp)

        def on_validation_epoch_end(self, val_outputs):                             # reference B:944-978
            res = evaluation.validation_epoch_end(val_outputs)
            for key in ("mean_fg_dice", "dice_per_class_or_region", "val_losses"):
                self.logger.log(key, res[key], self.current_epoch)

        def plot_network_architecture(self):
            pass

    return nnUNetTrainer_MLAgg_2D_dt_MS
