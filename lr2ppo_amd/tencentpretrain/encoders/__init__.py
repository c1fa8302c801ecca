"""Encoder registry (encoders/__init__.py:12-14 of the reference).  Only the transformer family and the dual
wrapper are on the LR2PPO path; the RNN/CNN encoders of the reference are out of scope."""
from .dual_encoder import DualEncoder
from .transformer_encoder import TransformerEncoder

str2encoder = {"transformer": TransformerEncoder, "dual": DualEncoder}
__all__ = ["TransformerEncoder", "DualEncoder", "str2encoder"]
