"""TEST INFRASTRUCTURE ONLY -- import shim for the *unmodified* reference modules.

Only usable where /root/reference exists (the build container).  It is used by
tests/golden/make_golden.py to (a) validate oracle/gigaam_oracle.py against the
reference's own code and (b) generate the committed golden vectors.  Nothing in
the product path (gigaam_amd/), the `-m gpu` tests, smoke() or bench.py imports it.

The reference's hot-path modules (gigaam/encoder.py, decoder.py, decoding.py,
utils.py) import third-party packages that are absent here (torchaudio, hydra,
omegaconf, soundfile); they are only needed by code paths outside SURVEY.md §8a,
so empty stub modules are registered in sys.modules before the import
(SURVEY.md §8c).
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GIGAAM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "gigaam"))


def _stub(name: str, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def import_reference():
    """Return a namespace with the reference's hot-path modules."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _stub("soundfile")
    ta = _stub("torchaudio")
    ta.transforms = _stub("torchaudio.transforms")
    ta.functional = _stub("torchaudio.functional")
    hy = _stub("hydra")
    hy.utils = _stub("hydra.utils")
    _stub("omegaconf", DictConfig=dict, OmegaConf=object)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    ns.encoder = importlib.import_module("gigaam.encoder")
    ns.decoder = importlib.import_module("gigaam.decoder")
    ns.decoding = importlib.import_module("gigaam.decoding")
    ns.utils = importlib.import_module("gigaam.utils")
    return ns
