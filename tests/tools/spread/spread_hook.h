/* tests/tools/spread/spread_hook.h -- measurement hook compiled into a private copy of oracle/yama_profile_oracle.c (-DMZO_SPREAD_STATS):
 * per band row, how far apart the REACHABLE states (above MININT / 2) lie -- what a 16-bit state would have to hold (HISTORY section 10). */
#include <stdint.h>
typedef struct { int32_t C, D, I; } tri_;
extern long long sp_rows, sp_cells;
extern long long sp_hist_row[40], sp_hist_nb[40], sp_hist_cell[40];    /* histograms over bit lengths of: a row's spread, |state(c) - state(c-1)|, a cell's max - min */
extern int32_t sp_row_lo, sp_row_hi, sp_prev[3]; extern int sp_have_prev;
static inline int sp_bits(long long v) { int b = 0; while (v > 0) { ++b; v >>= 1; } return b; }
#define SP_LIVE(x) ((x) > -(1 << 29))
#define MZO_SPREAD_CELL(r, c, lo, now) do { \
    const int32_t v_[3] = { (now).C, (now).D, (now).I }; int32_t mn_ = 0x7fffffff, mx_ = -0x7fffffff; int k_; \
    if ((c) == (lo)) sp_have_prev = 0; \
    for (k_ = 0; k_ < 3; ++k_) if (SP_LIVE(v_[k_])) { \
        if (v_[k_] < mn_) mn_ = v_[k_]; if (v_[k_] > mx_) mx_ = v_[k_]; \
        if (v_[k_] < sp_row_lo) sp_row_lo = v_[k_]; if (v_[k_] > sp_row_hi) sp_row_hi = v_[k_]; \
        if (sp_have_prev && SP_LIVE(sp_prev[k_])) { long long d_ = (long long)v_[k_] - sp_prev[k_]; sp_hist_nb[sp_bits(d_ < 0 ? -d_ : d_)]++; } \
    } \
    if (mx_ >= mn_) { sp_hist_cell[sp_bits((long long)mx_ - mn_)]++; ++sp_cells; } \
    sp_prev[0] = v_[0]; sp_prev[1] = v_[1]; sp_prev[2] = v_[2]; sp_have_prev = 1; } while (0)
#define MZO_SPREAD_ROW(r) do { if (sp_row_hi >= sp_row_lo) { sp_hist_row[sp_bits((long long)sp_row_hi - sp_row_lo)]++; ++sp_rows; } sp_row_lo = 0x7fffffff; sp_row_hi = -0x7fffffff; } while (0)
