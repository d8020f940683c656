"""Import shim: the package lives in ``audio-flamingo_amd/`` (hyphenated, as the repo layout requires);
this module makes it importable as ``audio_flamingo_amd`` by pointing ``__path__`` there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "audio-flamingo_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
