"""`from ...ViT.models.amc_transformer import AMCTransformer` (hyperparameter_tuning.py:19) resolves to the
MI355X-native ViT classifier; the implementation lives in vit_vs_raw_iq_amd.modules."""
from ...modules import AMCTransformerViT as AMCTransformer  # noqa: F401
