"""`plyfile` resolver for the drop-in directory.

The reference imports the third-party `plyfile` package (scene/gaussian_model.py:20, scene/dataset_readers.py).  With this
directory on PYTHONPATH a module of that name must exist even where the package is not installed (this image has no
network), but it must NOT shadow a real installation: the stand-in (_gsr_ply_standin.py) implements only what the
reference uses (scalar properties, binary LE/BE and ASCII), e.g. it raises on the list properties of mesh files.
So: if another `plyfile` is importable from the rest of sys.path, this module replaces itself with it; otherwise it
re-exports the stand-in."""
import importlib.machinery
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.machinery.PathFinder.find_spec(
    "plyfile", [p for p in sys.path if os.path.abspath(p or os.getcwd()) != _here])
if _spec is not None and _spec.origin and os.path.abspath(_spec.origin) != os.path.abspath(__file__):
    _real = importlib.util.module_from_spec(_spec)
    sys.modules[__name__] = _real            # the real package wins, for this and every later import
    _spec.loader.exec_module(_real)
else:
    from _gsr_ply_standin import *     # noqa: F401,F403
    from _gsr_ply_standin import PlyData, PlyElement, PlyProperty, PlyParseError   # noqa: F401
    IS_STANDIN = True
