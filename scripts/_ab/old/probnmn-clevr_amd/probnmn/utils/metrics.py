"""Stand-alone layout only: the running metrics live in ``probnmn.running_metrics`` (a name the reference's
``probnmn.utils`` package does not have, so that the models still import it when only
``probnmn.models`` / ``probnmn.modules`` are grafted onto the reference's package -- see probnmn_graft.py)."""
from probnmn.running_metrics import BLEU, Average, BooleanAccuracy  # noqa: F401
