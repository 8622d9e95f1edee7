"""Observed parity margins, written next to the pass/fail verdicts.

The parity tests bound outlier FRACTIONS (discontinuous decisions flip single pixels); `pytest -q` prints nothing about
how far inside the bound a run was.  Every comparison therefore also records what it measured - max abs error of a map,
fraction of pixels over 1e-4, max relative gradient error, rows over 1e-3 - into

    gpurun_out/parity_margins.json      (merged back from the GPU box; the round's copy is committed as
                                         profiles/rNN_parity_margins.json)

keyed by the pytest node id.  Recording never decides a verdict."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "gpurun_out", "parity_margins.json")


def record(section, **values):
    """Merge `values` (numbers / short lists) under the current test's node id and `section`."""
    node = os.environ.get("PYTEST_CURRENT_TEST", "unknown").split(" (")[0]
    try:
        import fcntl
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        with open(PATH + ".lock", "w") as lock:              # safe if the suite is run in several worker processes (-n)
            fcntl.flock(lock, fcntl.LOCK_EX)
            data = {}
            if os.path.exists(PATH):
                with open(PATH) as f:
                    data = json.load(f)
            data.setdefault(node, {}).setdefault(section, {}).update(values)
            tmp = PATH + ".tmp%d" % os.getpid()
            with open(tmp, "w") as f:
                json.dump(data, f, indent=1, sort_keys=True)
            os.replace(tmp, PATH)
    except (OSError, ValueError):
        pass
