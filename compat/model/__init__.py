"""Top-level `model` package for an UNCHANGED reference trainer: put `<repo>/compat` (and `<repo>`) ahead of the
reference's own tree on sys.path and `from model.ple import PLE` (run.py:15-26) resolves to the HIP mirror."""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)
import cdcmdr_amd  # noqa: E402,F401
