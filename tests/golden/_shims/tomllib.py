"""Import shim (fixture generation only): Python 3.10 has no tomllib; re-export tomli."""
from tomli import load, loads  # noqa: F401
