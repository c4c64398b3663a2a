from typing import List, Union

from ..layers import PlannedModule


class Neck(PlannedModule):
    """Abstract class for all necks (reference: mindpose/models/necks/neck.py:6-21; no concrete neck exists).
    Child classes implement ``emit`` and ``out_channels``."""

    @property
    def out_channels(self) -> Union[List[int], int]:
        raise NotImplementedError("Child class must implement this method.")
