"""Tolerances of the GPU parity tests: at most 10 x what was MEASURED on the MI355X.

Every image / film comparison of the `-m gpu` tests goes through check(label, measured): the tolerance is 10 x the value committed for that label in
tests/golden/parity_measured.json (what the comparison gave on the GPU box when the file was last recorded; floor 5e-8: f32 sums in another order),
and the assertion message carries the measured value.  PARITY_RECORD=1 records instead of asserting against the table (the values go to
gpurun_out/parity_measured.json; a wide sanity bound still applies): that is how the table is (re)made after a change that legitimately moves a
number — copy the file to tests/golden/ and commit it with the change."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_TABLE = os.path.join(ROOT, "tests", "golden", "parity_measured.json")
_OUT = os.path.join(ROOT, "gpurun_out", "parity_measured.json")
_table = None
FLOOR = 5e-8
FACTOR = 10.0


def _load():
    global _table
    if _table is None:
        try:
            with open(_TABLE) as f:
                _table = json.load(f)
        except OSError:
            _table = {}
    return _table


def tolerance(label, sanity):
    t = _load()
    return FACTOR * max(float(t[label]), FLOOR) if label in t else None


def check(label, measured, sanity=2e-2):
    """Asserts measured <= 10 x the committed measurement of `label` (and, always, < `sanity`: the bound the comparison had before there was a table)."""
    measured = float(measured)
    if os.environ.get("PARITY_RECORD"):
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        try:
            with open(_OUT) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            rec = {}
        rec[label] = max(measured, rec.get(label, 0.0))
        with open(_OUT, "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)
        assert measured < sanity, f"{label}: measured {measured:.3e} (sanity bound {sanity:.1e})"
        return
    tol = tolerance(label, sanity)
    assert tol is not None, f"{label}: no committed measurement in tests/golden/parity_measured.json (record one with PARITY_RECORD=1); measured {measured:.3e}"
    assert measured <= min(tol, sanity), f"{label}: measured {measured:.3e}, tolerance {min(tol, sanity):.1e} = 10 x the committed measurement {float(_load()[label]):.3e}"
