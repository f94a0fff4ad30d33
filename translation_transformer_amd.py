"""Import shim: the package directory is named ``translation-transformer_amd`` (with a hyphen), which the
import system cannot spell.  ``import translation_transformer_amd`` loads that directory as a package."""
import importlib.util
import sys
from pathlib import Path

_dir = Path(__file__).resolve().parent / "translation-transformer_amd"
_spec = importlib.util.spec_from_file_location(__name__, _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
