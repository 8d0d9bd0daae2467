"""`import ppca_rs` for existing users of viodotcom/ppca_rs: the same names (python/ppca_rs/__init__.py:3 re-exports
the extension module's classes, src/python_bindings.rs:15-26), served by the MI355X-native engine in `ppca_rs_amd`."""
from ppca_rs_amd import *  # noqa: F401,F403
from ppca_rs_amd import __all__, __version__  # noqa: F401
