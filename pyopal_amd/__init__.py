"""MI355X-native implementation of PyOpal's database-search hot path.

Drop-in for the names of ``pyopal`` (``src/pyopal/__init__.py:4-13``)::

    import pyopal_amd as pyopal
"""

from . import lib
from ._align import align
from .lib import (Aligner, Alphabet, BaseDatabase, Database, EndResult, FullResult, ScoreResult,
                  __version__)
from .matrices import ScoringMatrix

__all__ = [
    "Alphabet",
    "Aligner",
    "BaseDatabase",
    "Database",
    "ScoreResult",
    "EndResult",
    "FullResult",
    "align",
]
