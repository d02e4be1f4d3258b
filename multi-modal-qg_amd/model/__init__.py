"""Drop-in counterparts of the reference's ``model`` package (model/encoder.py, model/decoder.py)."""
from .decoder import AttnDecoder, Decoder  # noqa: F401
from .encoder import (AudioEncoder, AudioVideoEncoder, TextEncoder, VideoConvLstmEncoder,  # noqa: F401
                      VideoResnetEncoder)
