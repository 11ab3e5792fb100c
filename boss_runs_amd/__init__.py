"""Import alias: the product lives in the directory `boss-runs_amd/` (a name Python cannot
import directly because of the hyphen).  This stub package points its search path there, so
`import boss_runs_amd.runs` resolves to `boss-runs_amd/runs.py`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "boss-runs_amd")]

from ._version import __version__  # noqa: E402,F401
