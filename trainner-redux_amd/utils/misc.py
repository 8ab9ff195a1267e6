"""Small helpers shared with the reference's naming (log keys are part of the contract)."""
from __future__ import annotations

import random

import numpy as np
import torch


def loss_type_to_label(loss_type: str, network_label: str = "g") -> str:
    """`L1Loss` -> `l_g_l1`; r3ganloss / multiscaleganloss -> `l_g_gan`
    (/root/reference/traiNNer/utils/misc.py:272-280; keys asserted by tests/test_models/test_sr_model.py:66-80)."""
    low = loss_type.lower()
    if low in ("r3ganloss", "multiscaleganloss"):
        return f"l_{network_label}_gan"
    return f"l_{network_label}_{low.replace('loss', '')}"


def set_random_seed(seed: int) -> None:
    """/root/reference/traiNNer/utils/misc.py:19-22 (+ numpy / python for the data side)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
