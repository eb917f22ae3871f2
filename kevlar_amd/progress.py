"""kevlar.progress of the reference: the class lives in kevlar_amd.reporting."""
from kevlar_amd.reporting import ProgressIndicator  # noqa: F401
