"""Checkpoint I/O with the reference's file format (utils.py:7-28): a `torch.save`d dict
{'optimizer', 'model', 'ema', 'step'} of state dicts.

Differences a user meets: no tensorflow (`tf.io.gfile` becomes `os`), and `restore_checkpoint`
accepts checkpoints written by the reference, whose model keys carry the `module.` prefix of
`torch.nn.DataParallel` (models/utils.py:93) -- `save_checkpoint(..., data_parallel_prefix=True)` writes that
form back for the reference to read.  Loading copies IN PLACE: parameters stay views of the flat
buffer the fused training step uses, Adam moments and the EMA shadow copy are re-homed on the next
step, and every engine lowered from the model re-packs its weights (four launches) on its next use.
"""
import logging
import os

import torch

from .models.utils import strip_data_parallel_prefix


def load_checkpoint_file(path, device):
    """torch.load of a reference-format checkpoint.  A checkpoint the reference writes holds one non-tensor numpy object:
    optimization_manager sets lr = lr * np.minimum(step / warmup, 1.0) (losses.py:46), so Adam's param_groups carry a
    numpy.float64, which torch >= 2.6's default weights-only unpickler refuses.  The numpy scalar constructors are
    allow-listed for this load (and nothing else: no arbitrary pickle execution)."""
    import numpy as np
    allow = [np._core.multiarray.scalar, np.dtype] + [type(np.dtype(t)) for t in (np.float64, np.float32, np.int64, np.int32)]
    with torch.serialization.safe_globals(allow):
        return torch.load(path, map_location=device)


def restore_checkpoint(ckpt_dir, state, device):
    if not os.path.exists(ckpt_dir):
        os.makedirs(os.path.dirname(ckpt_dir) or ".", exist_ok=True)
        logging.warning(f"No checkpoint found at {ckpt_dir}. Returned the same state as input")
        return state
    loaded_state = load_checkpoint_file(ckpt_dir, device)
    state['optimizer'].load_state_dict(loaded_state['optimizer'])
    state['model'].load_state_dict(strip_data_parallel_prefix(loaded_state['model']), strict=False)
    state['ema'].load_state_dict(loaded_state['ema'])
    state['step'] = loaded_state['step']
    return state


def save_checkpoint(ckpt_dir, state, data_parallel_prefix=False):
    model_sd = state['model'].state_dict()
    if data_parallel_prefix:
        model_sd = {"module." + k: v for k, v in model_sd.items()}
    saved_state = {
        'optimizer': state['optimizer'].state_dict(),
        'model': model_sd,
        'ema': state['ema'].state_dict(),
        'step': state['step'],
    }
    torch.save(saved_state, ckpt_dir)
