"""Re-export of oracle/oracle_lib.py so tests can `import oracle_lib`."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import importlib.util as _u

_spec = _u.spec_from_file_location("_photon_oracle_lib", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "oracle_lib.py"))
_mod = _u.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
