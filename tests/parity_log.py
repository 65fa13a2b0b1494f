"""Parity numbers measured by the whole-path tests, printed as one table by the
`pytest_terminal_summary` hook of conftest.py so that the measured errors (and which reference
each full-size case was compared with) reach the driver's test record."""
ROWS = []        # (case, conv_mode, gain, max_rel, mean_l1_rel, ref, max_pixel_rel)


def record(case: str, mode: str, gain, max_rel: float, mean_l1_rel: float, ref: str, max_pixel_rel: float = float("nan")) -> None:
    """max_rel = max |d| / max |ref| (SURVEY.md §8(d)'s definition, the north star's bar); max_pixel_rel = max over pixels of
    |d| / |ref| -- what a depth consumer sees (distance = bf / inv_dist; inv_dist >= 0.96 everywhere, so it is well defined)."""
    ROWS.append((case, mode, float(gain), float(max_rel), float(mean_l1_rel), ref, float(max_pixel_rel)))
