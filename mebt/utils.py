"""import-name alias of reference mebt/utils.py for the helpers the hot path's callers use (shift_dim :30-53, view_range
:61-78, accuracy :80-94); implemented in mebt_amd/utils.py"""
from mebt_amd.utils import shift_dim, view_range, accuracy  # noqa: F401
