"""Importable alias of the package directory ``language-transformer-vae_amd/`` (a hyphen cannot be
imported by name): ``import lvae`` loads that directory as this package."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "language-transformer-vae_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
