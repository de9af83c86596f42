"""vit-vs-raw-iq_amd: MI355X-native (gfx950) training path for the ViT and raw-IQ modulation
classifiers of aliftffd/ViT-vs-Raw-IQ.

Import name: `vit_vs_raw_iq_amd` (the directory name contains hyphens; the repo-root module
`vit_vs_raw_iq_amd.py` registers this package under that name).

  modules   -- the reference's nn.Module surface (AMCTransformer x2, Encoder x2, layer shells)
  trainer   -- fused native training step (CE + backward + clip + AdamW) and data-parallel driver
  data      -- seeded synthetic IQ frames (the reference ships no data)
  _native   -- ctypes binding of include/iqvit.h  (libiqvit.so; no fallback)
  ViT.models.amc_transformer / transformer_rawIQ.models.transformer_rawIQ
            -- import paths used by the reference's scripts (hyperparameter_tuning.py:19,37)
"""
from .modules import (AMCTransformerViT, AMCTransformerRawIQ, EncoderViT, EncoderRawIQ, EncoderLayer, LayerNorm,
                      MultiHeadAttention, PositionwiseFeedForward, ScaleDotProductAttention, PatchEmbedding,
                      SequenceEmbedding, NativePlan)
from ._native import IqError, LIB_PATH

__all__ = ["AMCTransformerViT", "AMCTransformerRawIQ", "EncoderViT", "EncoderRawIQ", "EncoderLayer", "LayerNorm",
           "MultiHeadAttention", "PositionwiseFeedForward", "ScaleDotProductAttention", "PatchEmbedding",
           "SequenceEmbedding", "NativePlan", "IqError", "LIB_PATH"]
