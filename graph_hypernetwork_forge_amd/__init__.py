"""Import alias for the ``graph-hypernetwork-forge_amd/`` source directory.

The product lives in ``graph-hypernetwork-forge_amd/`` (a name Python cannot
import); this package points its search path there so that
``import graph_hypernetwork_forge_amd`` and its submodules resolve to it.
"""

import os as _os

_SRC = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                     "graph-hypernetwork-forge_amd")
__path__.insert(0, _SRC)

from ._api import *          # noqa: E402,F401,F403
from ._api import __all__, __version__   # noqa: E402,F401
