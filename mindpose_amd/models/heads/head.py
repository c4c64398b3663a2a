from ..layers import PlannedModule


class Head(PlannedModule):
    """Abstract class for all heads (reference: mindpose/models/heads/head.py:4)."""
