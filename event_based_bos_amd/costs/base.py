"""``CostBase`` -- base class of the cost plugins (reference: src/costs/base.py:11-77).

Contract kept from the reference: ``direction`` in {minimize, maximize, natural} (else ValueError);
``calculate(arg: dict)`` is wrapped so that a missing dict key logs ``required_keys`` and re-raises
``KeyError``, and so that the loss history records ``loss.item()`` when ``store_history`` is on.
"""
import functools
import logging
from typing import Dict, List

import torch

from ..types import FLOAT_TORCH

logger = logging.getLogger(__name__)

_DIRECTIONS = ("minimize", "maximize", "natural")


def _catch_key_error(func):
    """Log the keys a cost needs when the argument dict lacks one, then re-raise."""

    @functools.wraps(func)
    def wrapper(self, arg: dict):
        try:
            return func(self, arg)
        except KeyError as err:
            logger.error("Input for the cost needs keys of:")
            logger.error(self.required_keys)
            raise err

    return wrapper


def _register_history(func):
    """Append the scalar loss to ``self.history['loss']`` when history is enabled."""

    @functools.wraps(func)
    def wrapper(self, arg: dict):
        loss = func(self, arg)
        if self.store_history:
            self.history["loss"].append(self.get_item(loss))
        return loss

    return wrapper


class CostBase(object):
    """Base of the cost classes.

    Args:
        direction (str) ... 'minimize', 'maximize' or 'natural' (a more interpretable value).
        store_history (bool) ... record every evaluated loss.
    """

    required_keys: List[str] = []

    def __init__(self, direction="minimize", store_history: bool = False, *args, **kwargs):
        if direction not in _DIRECTIONS:
            e = f"direction should be minimize, maximize, and natural. Got {direction}."
            logger.error(e)
            raise ValueError(e)
        self.direction = direction
        self.store_history = store_history
        self.clear_history()

    # decorators, reachable as CostBase.catch_key_error / CostBase.register_history by subclasses
    catch_key_error = staticmethod(_catch_key_error)
    register_history = staticmethod(_register_history)

    def get_item(self, loss: FLOAT_TORCH) -> float:
        return loss.item() if isinstance(loss, torch.Tensor) else loss

    def clear_history(self) -> None:
        self.history: Dict[str, list] = {"loss": []}

    def get_history(self) -> dict:
        return self.history.copy()

    def enable_history_register(self) -> None:
        self.store_history = True

    def disable_history_register(self) -> None:
        self.store_history = False

    @_register_history
    @_catch_key_error
    def calculate(self, arg: dict) -> FLOAT_TORCH:
        raise NotImplementedError
