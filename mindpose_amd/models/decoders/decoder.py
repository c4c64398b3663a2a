import torch.nn as nn


class Decoder(nn.Module):
    """Abstract class for all decoders (reference: mindpose/models/decoders/decoder.py:4)."""
