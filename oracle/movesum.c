/* Oracle (test infrastructure): restatement of Bottleneck 1.3.x `move_sum` for the one case
 * the reference uses (1-D float64, no NaNs, min_count=1), call sites
 * /root/reference/boss/runs/reference.py:233-234 and :259-260.
 *
 * Published algorithm (bottleneck/src/move_template.c, MOVE(move_sum)): one running
 * accumulator; for i < window: asum += a[i]; for i >= window: asum += a[i] - a[i-window]
 * (the difference is formed first, then added).  PINNED to the library itself: tests/golden/g_movesum.npz
 * holds outputs of the real compiled Bottleneck 1.3.2 (tests/golden/make_movesum_golden.py, run with
 * /opt/conda/bin/python3.9) on every bin-sum column of the golden runs and on random arrays over forty
 * decades; tests/test_oracle_golden.py::test_move_sum_equals_bottleneck holds this file to them bit for bit.
 *
 * `stride` is in elements and may be negative (the reference passes reversed views
 * `scores_ds[::-1, b]`); the output is written contiguously in iteration order.
 * Returns 0, or -1 when window is outside [1, n] (Bottleneck raises ValueError there).
 */
#include <stddef.h>

int oracle_move_sum(const double *a, ptrdiff_t stride, long n, long window, double *y)
{
    if (window < 1 || window > n) return -1;
    double asum = 0.0;
    long i = 0;
    for (; i < window; ++i) {
        asum += a[i * stride];
        y[i] = asum;
    }
    for (; i < n; ++i) {
        double ai = a[i * stride];
        double aold = a[(i - window) * stride];
        asum += ai - aold;
        y[i] = asum;
    }
    return 0;
}

/* In-order 100-site bin accumulation, the float64 `np.add.at(ds, arange(L)//w, scores)` of
 * reference.py:229-231 (sequential, unbuffered): ds[i / w] += s[i * stride]. */
void oracle_bin_add(const double *s, ptrdiff_t stride, long n, long w, double *ds)
{
    for (long i = 0; i < n; ++i) ds[i / w] += s[i * stride];
}
