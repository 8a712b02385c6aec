"""Import alias: the product package lives in the directory `epa-ng_amd/` (not a valid Python
identifier), this module makes it importable as `epa_ng_amd`."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 "epa-ng_amd"))
from .api import *  # noqa: F401,F403,E402
