"""torchplus.train: checkpoint index + fastai-style optimizer wrapper and schedules
(reference: rslo/torchplus/train/__init__.py:1-8)."""
from torchplus.train.checkpoint import (latest_checkpoint, restore, restore_latest_checkpoints, restore_models,
                                        save, save_models, save_models_cpu, try_restore_latest_checkpoints)
from torchplus.train.common import create_folder
from torchplus.train import fastai_optim, learning_schedules_fastai  # noqa: F401
