// Host-side read-length distribution update (include/bossx.h: bossx_rl_update).
//
// Replaces, for the decision-update path,
//   ReadlengthDist.update / ccl_approx_constant     /root/reference/boss/readlengthdist.py:36-97
// It sits on the critical path of an update — the move_sum windows come out of it — so it is
// native: one pass over the 4000 new lengths, one over the histogram up to the longest read,
// and the float cumsum of the pmf only as far as the last level crossing.  Every floating-point
// operation is the one numpy performs in the reference, in the same order (sequential cumsum,
// one division per pmf entry, `1 - cs > prob` comparisons), so approx_ccl / lam / time_cost are
// bit-identical; compiled without fast-math and without FMA contraction.
#include <cstdint>
#include <vector>

#include "bossx.h"

extern "C" int bossx_rl_update(uint16_t *hist, int64_t hist_len, const int64_t *lens, int64_t n_lens,
                               int64_t min_len_exclusive, int32_t eta, int64_t *hi_inout,
                               double *lam, int64_t *longest_read, int32_t *approx_ccl, int32_t *observed) {
    if (!hist || hist_len < 2 || (n_lens > 0 && !lens) || eta < 2 || !hi_inout || !lam || !longest_read ||
        !approx_ccl || !observed)
        return BOSSX_E_INVALID;
    int64_t hi = *hi_inout;
    for (int64_t i = 0; i < n_lens; ++i) {
        int64_t l = lens[i];
        if (l <= min_len_exclusive) continue;               // rejected / short reads (readlengthdist.py:44)
        if (l >= hist_len) l = hist_len - 1;                // reads longer than 1M count as 1M
        ++hist[l];                                          // uint16 wrap-around like the reference's `+= 1`
        if (l > hi) hi = l;
    }
    *hi_inout = hi;
    // wrap-around can zero the top counters: the reference takes the last non-zero one
    int64_t top = hi;
    while (top > 0 && hist[top] == 0) --top;
    uint64_t total = 0, dot = 0;
    for (int64_t i = 0; i <= top; ++i) { total += hist[i]; dot += uint64_t(i) * hist[i]; }
    if (total == 0) { *observed = 0; return BOSSX_OK; }
    *observed = 1;
    *longest_read = top;
    const double tot = double(total);
    *lam = double(dot) / tot;                               // int64 / uint64 true division of numpy
    // ccl[i + 1] = 1 - cumsum(L[1:])[i]; level k is crossed at the first entry <= prob_k; the
    // appended final zero (index len(L)) always qualifies
    const int nlev = eta - 1;
    const int64_t n = top;                                  // len(L[1:])
    double cs = 0.0;
    int64_t i = 0;
    for (int k = 0; k < nlev; ++k) {
        const double prob = 1 - (k + 0.5) / (eta - 1);
        // entries already passed have 1 - cs_i > prob_{k-1} > prob_k?  No: the levels decrease, so a
        // later level is crossed at the same or a later entry — continue from the current one
        while (i < n) {
            const double next = cs + double(hist[i + 1]) / tot;     // cs_i = L[1] + ... + L[i + 1]
            if (!(1.0 - next > prob)) break;
            cs = next;
            ++i;
        }
        approx_ccl[k] = int32_t(i + 1);
    }
    return BOSSX_OK;
}
