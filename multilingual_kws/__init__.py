"""The reference's import path: `from multilingual_kws.embedding import input_data, transfer_learning,
batch_streaming_analysis` and `from multilingual_kws import run` resolve to the MI355X build in multilingual_kws_amd -- the SAME module
objects, so a caller of harvard-edge/multilingual_kws (run.py:15-18) imports unchanged.  Nothing lives here but the aliases."""
import importlib
import sys

_ALIASES = ("run", "train_multilingual_embedding")


def __getattr__(name):          # lazy: `import multilingual_kws` must not pull in torch
    if name in _ALIASES:
        mod = importlib.import_module("multilingual_kws_amd." + name)
        sys.modules[__name__ + "." + name] = mod
        return mod
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


class _AliasFinder:
    """`import multilingual_kws.run` / `from multilingual_kws import run` -> multilingual_kws_amd.run."""

    @staticmethod
    def find_spec(fullname, path=None, target=None):
        head, _, tail = fullname.partition(".")
        if head != __name__ or tail not in _ALIASES:
            return None
        import importlib.util
        mod = importlib.import_module("multilingual_kws_amd." + tail)
        sys.modules[fullname] = mod
        return importlib.util.spec_from_loader(fullname, _Loader(mod))


class _Loader:
    def __init__(self, mod):
        self.mod = mod

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder)
