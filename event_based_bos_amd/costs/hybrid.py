"""``HybridCost`` -- weighted sum of registered costs (reference: src/costs/hybrid.py:12-79).
A weight may be a number or the string "inv" (the term becomes 1 / cost)."""
import logging
from typing import Union

import torch

from . import CostBase, functions

logger = logging.getLogger(__name__)


class HybridCost(CostBase):
    """Args:
        direction (str) ... forwarded to every member cost.
        cost_with_weight (dict) ... {cost name: weight | "inv"}.
    """

    name = "hybrid"

    def __init__(self, direction: str, cost_with_weight: dict, store_history: bool = False, *args, **kwargs):
        logger.info(f"Log functions are mix of {cost_with_weight}")
        self.cost_func = {}
        for key, weight in cost_with_weight.items():
            member = functions[key](direction=direction, store_history=store_history, *args, **kwargs)
            self.cost_func[key] = {"func": member, "weight": weight}
        super().__init__(direction=direction, store_history=store_history)
        self.required_keys = [k for entry in self.cost_func.values() for k in entry["func"].required_keys]

    def update_weight(self, cost_with_weight):
        assert set(self.cost_func.keys()) == set(cost_with_weight.keys())
        for key, weight in cost_with_weight.items():
            self.cost_func[key]["weight"] = weight

    @CostBase.register_history
    @CostBase.catch_key_error
    def calculate(self, arg: dict) -> Union[float, torch.Tensor]:
        loss = 0.0
        for entry in self.cost_func.values():
            value = entry["func"].calculate(arg)
            if entry["weight"] == "inv":
                loss += 1.0 / value
            else:
                loss += entry["weight"] * value
        return loss

    # histories are kept per member cost as well (:59-79)
    def clear_history(self) -> None:
        self.history = {"loss": []}
        for entry in self.cost_func.values():
            entry["func"].clear_history()

    def get_history(self) -> dict:
        out = self.history.copy()
        for name, entry in self.cost_func.items():
            out[name] = entry["func"].get_history()["loss"]
        return out

    def enable_history_register(self) -> None:
        self.store_history = True
        for entry in self.cost_func.values():
            entry["func"].store_history = True

    def disable_history_register(self) -> None:
        self.store_history = False
        for entry in self.cost_func.values():
            entry["func"].store_history = False
