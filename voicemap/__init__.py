"""Import-compatibility alias: ``from voicemap.models import ...`` / ``voicemap.utils`` / ``voicemap.librispeech`` resolve
to the MI355X implementation in ``voicemap_amd`` so the reference's experiment scripts keep their import lines."""
import importlib
import sys

for _name in ("models", "utils", "librispeech"):
    _mod = importlib.import_module("voicemap_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
