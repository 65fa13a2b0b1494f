"""Parity numbers measured by the whole-path tests, printed as one table by the
`pytest_terminal_summary` hook of conftest.py so that the measured errors (and which reference
each full-size case was compared with) reach the driver's test record."""
ROWS = []        # (case, conv_mode, gain, max_rel, mean_l1_rel, ref)


def record(case: str, mode: str, gain, max_rel: float, mean_l1_rel: float, ref: str) -> None:
    ROWS.append((case, mode, float(gain), float(max_rel), float(mean_l1_rel), ref))
