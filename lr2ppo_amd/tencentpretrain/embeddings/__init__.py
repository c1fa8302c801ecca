"""Embedding registry (embeddings/__init__.py:13-16 of the reference), restricted to the sub-embeddings that the
ViT-B/16 (patch + pos) and RoBERTa-base (word + pos + seg) configs use."""
from .embedding import (DualEmbedding, Embedding, PatchEmbedding, PosEmbedding, SegEmbedding, WordEmbedding)

str2embedding = {"word": WordEmbedding, "pos": PosEmbedding, "seg": SegEmbedding, "patch": PatchEmbedding,
                 "dual": DualEmbedding}
__all__ = ["Embedding", "WordEmbedding", "PosEmbedding", "SegEmbedding", "PatchEmbedding", "DualEmbedding", "str2embedding"]
