"""Import-path mirror: ViT encoder."""
from ...modules import EncoderViT as Encoder  # noqa: F401
