"""The reference keeps its classes in the extension submodule `ppca_rs.ppca_rs` (python/ppca_rs/__init__.py:3) and some
user code imports them from there (examples/ppca_mixture.py:4); same names here."""
from ppca_rs_amd import *  # noqa: F401,F403
