"""Import shim (fixture generation only): the system Python of this image has no Bottleneck.

`move_sum` below restates the algorithm of Bottleneck 1.3.x (`move_sum` in
bottleneck/src/move_template.c, pinned `bottleneck~=1.3.7` by the reference's
pyproject.toml:19) for the only case the reference uses: 1-D float64 input without NaNs.
A single running accumulator: the first `window` outputs add a[i]; afterwards
`asum += a[i] - a[i-window]`.  PINNED: `tests/golden/g_movesum.npz` holds the outputs of the
real compiled Bottleneck 1.3.2 (/opt/conda/bin/python3.9, `make_movesum_golden.py`) on every
`scores_ds` column the reference passed through this shim in the golden runs, plus random arrays
over forty decades; `tests/test_oracle_golden.py` holds this function to them bit for bit.
"""
import numpy as np


def move_sum(a, window, min_count=None, axis=-1):
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    if window < 1:
        raise ValueError("Moving window (=%d) must between 1 and %d, inclusive" % (window, n))
    if window > n:
        raise ValueError("Moving window (=%d) must between 1 and %d, inclusive" % (window, n))
    if min_count is None:
        min_count = window
    x = a.tolist()
    y = [0.0] * n
    asum = 0.0
    for i in range(min(window, n)):
        asum += x[i]
        y[i] = asum if (i + 1) >= min_count else float("nan")
    for i in range(window, n):
        asum += x[i] - x[i - window]
        y[i] = asum
    return np.array(y, dtype=np.float64)
