"""Import alias for the ``l-step_amd/`` source tree.

The project directory is named ``l-step_amd`` (a hyphen is not a legal Python
identifier), so this one-file package forwards ``import lstep_amd.<x>`` to the
modules that live in ``../l-step_amd/``.  Nothing else lives here.
"""
import os as _os

_SRC = _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "l-step_amd"))
__path__.insert(0, _SRC)  # noqa: F821  (package attribute)
SRC_DIR = _SRC
