"""One rank's supervisor for tests/test_launch_guard.py: probnmn.launch_guard.supervise around tests/helpers/guard_worker.py."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "probnmn-clevr_amd"))
from probnmn import launch_guard  # noqa: E402

sys.exit(launch_guard.supervise([sys.executable, os.path.join(HERE, "guard_worker.py"), sys.argv[1]], watchdog_s=4.0,
                                first_beat_s=60.0,
                                last_resort=lambda info: {"value": None, "hung": True, "launch_guard": info}))
