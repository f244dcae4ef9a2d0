"""Import alias for the package directory ``multiplexed-image-annotator_amd/``.

The directory name required by the repo layout is not a valid Python identifier, so this
module turns itself into that package: it points ``__path__`` at the directory and runs the
directory's ``__init__.py`` in its own namespace.  ``import multiplexed_image_annotator_amd``
and ``from multiplexed_image_annotator_amd.annotator import Annotator`` then work as usual.
"""
import os as _os

__package__ = __name__
__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "multiplexed-image-annotator_amd")]
_init = _os.path.join(__path__[0], "__init__.py")
with open(_init, "r") as _f:
    exec(compile(_f.read(), _init, "exec"))
del _f, _init
