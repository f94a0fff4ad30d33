"""translation-transformer_amd: MI355X-native speculative-decoding inference path of
Academich/translation-transformer.  Import as ``translation_transformer_amd`` through the
shim at the repository root (the directory name carries a hyphen)."""
from ._native import build, lib, TtxError, ReferenceError_  # noqa: F401
from .model import NativeTransformer, reference_pe_table  # noqa: F401
from .decoding import (TranslationInferenceGreedySpeculative, TranslationInferenceGreedy,  # noqa: F401
                       TranslationInferenceBeamSearch, TranslationInferenceBeamSearchSpeculative)
from .lightning_model import VanillaEncoderDecoderTransformerLightning, run_predict  # noqa: F401,E402
from . import dist  # noqa: F401,E402
from .tokenizer import NativeSmilesTokenizer  # noqa: F401,E402
from . import scheduling, scoring  # noqa: F401,E402
