# the reference's fireflies/utils/__init__.py is empty; submodules are imported explicitly
# (`import fireflies.utils.math`).  Importing them here only saves the user that line.
from . import math as math  # noqa: F401
from . import transforms as transforms  # noqa: F401
