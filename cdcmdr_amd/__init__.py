"""Importable alias of the package directory `causal-domain-clustering-for-multi-domain-recommendation_amd/`
(whose name, fixed by the project layout, is not a valid Python identifier).  Submodules resolve from
that directory: `import cdcmdr_amd.model.ple`, `from cdcmdr_amd import plan`, ...
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "causal-domain-clustering-for-multi-domain-recommendation_amd")
__path__.insert(0, _REAL)

with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
