"""MI355X-native match -> rank -> top-K path of Manticore Search.

Importing the package needs csrc/libmrk.so (build: ``__graft_entry__.build()``); there is
no CPU fallback.
"""
from .api import *  # noqa: F401,F403
from .api import __all__  # noqa: F401
