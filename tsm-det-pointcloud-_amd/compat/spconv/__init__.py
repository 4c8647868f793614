"""`import spconv` compatibility alias: the MI355X-native operators of `spx` under the names the reference imports
(pcdet/utils/spconv_utils.py:3-6, pcdet/datasets/processor/data_processor.py:19-26).  Put this directory on PYTHONPATH
together with its parent (which holds `spx`)."""
from spx import *  # noqa: F401,F403
from spx import conv, ops  # noqa: F401

__version__ = "2.3.8+spx"
