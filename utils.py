"""Top-level `utils.instantiate_from_config`, the import the reference's scripts use
(reference utils.py:3-7); the implementation lives in mebt_amd.config."""
from mebt_amd.config import instantiate_from_config  # noqa: F401
