"""Import-path mirror: raw-IQ encoder."""
from ...modules import EncoderRawIQ as Encoder  # noqa: F401
