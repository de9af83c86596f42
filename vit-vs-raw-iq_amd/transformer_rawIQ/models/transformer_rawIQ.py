"""`from ...transformer_rawIQ.models.transformer_rawIQ import AMCTransformer` (hyperparameter_tuning.py:37)
resolves to the MI355X-native raw-IQ classifier; the implementation lives in vit_vs_raw_iq_amd.modules."""
from ...modules import AMCTransformerRawIQ as AMCTransformer  # noqa: F401
