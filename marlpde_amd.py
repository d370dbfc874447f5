"""Import alias: ``marlpde_amd`` -> ``integrating-diagenetic-equations-using-python_amd/``.

The product directory keeps the upstream repository's (hyphenated) name; giving this module that
directory as its ``__path__`` makes it importable as a regular package.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "integrating-diagenetic-equations-using-python_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _fh:
    exec(compile(_fh.read(), _fh.name, "exec"))
del _os, _fh
