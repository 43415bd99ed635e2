"""Import alias for the package in `dinov2.cpp_amd/` (a directory name with a dot is not a Python identifier).

`import dinov2_cpp_amd`, `python -m dinov2_cpp_amd.inference ...`, `python -m dinov2_cpp_amd.quantize ...` and
`python -m dinov2_cpp_amd.convert ...` work from the repository root (or with it on PYTHONPATH); all the code lives in
`dinov2.cpp_amd/`, this directory holds nothing else.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dinov2.cpp_amd")
__path__.insert(0, _REAL)  # sub-modules (api, synth, inference, ...) are found in the real directory
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _f
