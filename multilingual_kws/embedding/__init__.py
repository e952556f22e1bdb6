"""`multilingual_kws.embedding.<module>` = `multilingual_kws_amd.embedding.<module>` (the same module object): the drop-in modules under
the reference's own import path."""
import importlib
import sys

_MODULES = ("input_data", "transfer_learning", "batch_streaming_analysis", "single_target_recognize_commands", "distance_filtering", "tpr_fpr")

for _m in _MODULES:
    _mod = importlib.import_module("multilingual_kws_amd.embedding." + _m)
    sys.modules[__name__ + "." + _m] = _mod
    globals()[_m] = _mod
del _m, _mod
