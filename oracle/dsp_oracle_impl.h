/* dsp_oracle_impl.h -- type-generic bodies of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Included twice by dsp_oracle.c with
 *     T   = float  / double      (the gufunc loop's array type)
 *     SFX = f32    / f64
 * Every function restates, operation for operation, one numba kernel body of the reference
 * (file:line given per function, paths relative to /root/reference/src/dspeed/processors/).
 * Arithmetic contract: `T op T` stays in T; anything that numba mixes
 * with an int literal / int32 argument / float64 value is evaluated in double and only the store
 * into a T array rounds.  Compiled with -fno-fast-math -ffp-contract=off: one IEEE rounding per
 * written operation, no FMA, no reassociation -- like the numba loops (no fastmath flag,
 * utils.py:215-218).
 *
 * Return value of each row function: 0, or an ORC_E_* code when the reference raises DSPFatal.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

static int FN(row_has_nan)(const T* w, int n) {
    for (int i = 0; i < n; ++i)
        if (isnan(w[i])) return 1;
    return 0;
}

static void FN(fill_nan)(T* w, int n) {
    for (int i = 0; i < n; ++i) w[i] = (T)NAN;
}

/* bl_subtract.py:11-46 */
static int FN(row_bl_subtract)(const T* w_in, int n, T baseline, T* w_out) {
    FN(fill_nan)(w_out, n);
    if (FN(row_has_nan)(w_in, n) || isnan(baseline)) return 0;
    for (int i = 0; i < n; ++i) w_out[i] = w_in[i] - baseline; /* T - T -> T */
    return 0;
}

/* pole_zero.py:24-77.  `-1 / t_tau` is int64/floatT -> float64 in numba, so `constant` is a double in
 * both loops; the recursion state w_tmp is float64 (:63); the store narrows to T (:72). */
static int FN(row_pole_zero)(const T* w_in, int n, T t_tau, T* w_out) {
    FN(fill_nan)(w_out, n);
    if (FN(row_has_nan)(w_in, n) || isnan(t_tau)) return 0;
    const double constant = exp(-1.0 / (double)t_tau);
    double acc;
    w_out[0] = w_in[0];
    acc = (double)w_in[0];
    for (int i = 1; i < n; ++i) {
        double nxt = (acc + (double)w_in[i]) - (double)w_in[i - 1] * constant;
        w_out[i] = (T)nxt;
        acc = nxt;
    }
    if (FN(row_has_nan)(w_out, n)) return ORC_E_PZ_NAN;
    return 0;
}

/* pole_zero.py:82-198.  a, b are float64 (same int64/floatT rule, :168-169); frac*b is T*float64 -> float64;
 * the expression at :187-193 is evaluated left to right in float64. */
static int FN(row_double_pole_zero)(const T* w_in, int n, T t_tau1, T t_tau2, T frac, T* w_out) {
    FN(fill_nan)(w_out, n);
    if (FN(row_has_nan)(w_in, n) || isnan(t_tau1) || isnan(t_tau2) || isnan(frac)) return 0;
    if (n <= 3) return ORC_E_DPZ_SHORT;
    const double a = exp(-1.0 / (double)t_tau1);
    const double b = exp(-1.0 / (double)t_tau2);
    const double fr = (double)frac;
    const double den1 = ((fr * b - fr * a) - b) - 1.0;
    const double den2 = -1.0 * ((fr * b - fr * a) - b);
    const double num1 = -1.0 * (a + b);
    const double num2 = a * b;
    double t0 = (double)w_in[0], t1 = (double)w_in[1];
    w_out[0] = w_in[0];
    w_out[1] = w_in[1];
    for (int i = 2; i < n; ++i) {
        double t2 = ((((double)w_in[i] + num1 * (double)w_in[i - 1]) + num2 * (double)w_in[i - 2]) - den1 * t1) - den2 * t0;
        w_out[i] = (T)t2;
        t0 = t1;
        t1 = t2;
    }
    return 0;
}

/* trap_filters.py:12-76.  Everything is T (op) T -> T with the feedback through the T output array. */
static int FN(row_trap_filter)(const T* w_in, int n, int rise, int flat, T* w_out) {
    FN(fill_nan)(w_out, n);
    if (FN(row_has_nan)(w_in, n)) return 0;
    if (rise < 0) return ORC_E_TRAP_RISE;
    if (flat < 0) return ORC_E_TRAP_FLAT;
    if (2 * (long)rise + flat > n) return ORC_E_TRAP_WIDE;
    if (n == 0) return 0;
    w_out[0] = w_in[0];
    for (int i = 1; i < rise; ++i) w_out[i] = w_out[i - 1] + w_in[i];
    for (int i = rise; i < rise + flat; ++i) {
        T prev = w_out[i > 0 ? i - 1 : n - 1]; /* python negative index wraps: rise==0 reads the NaN fill */
        w_out[i] = (prev + w_in[i]) - w_in[i - rise];
    }
    for (int i = rise + flat; i < 2 * rise + flat; ++i) {
        T prev = w_out[i > 0 ? i - 1 : n - 1];
        w_out[i] = ((prev + w_in[i]) - w_in[i - rise]) - w_in[i - rise - flat];
    }
    for (int i = 2 * rise + flat; i < n; ++i) {
        T prev = w_out[i > 0 ? i - 1 : n - 1];
        w_out[i] = (((prev + w_in[i]) - w_in[i - rise]) - w_in[i - rise - flat]) + w_in[i - 2 * rise - flat];
    }
    return 0;
}

/* trap_filters.py:79-149.  `expr / rise` is T/int32 -> float64, `w_out[i-1] + float64` -> float64, store rounds. */
static int FN(row_trap_norm)(const T* w_in, int n, int rise, int flat, T* w_out) {
    FN(fill_nan)(w_out, n);
    if (FN(row_has_nan)(w_in, n)) return 0;
    if (rise < 0) return ORC_E_TRAP_RISE;
    if (flat < 0) return ORC_E_TRAP_FLAT;
    if (2 * (long)rise + flat > n) return ORC_E_TRAP_WIDE;
    if (n == 0) return 0;
    if (rise == 0) return ORC_E_ZERODIV; /* numba (error_model='python') raises ZeroDivisionError at w_in[0] / rise */
    const double r = (double)rise;
    w_out[0] = (T)((double)w_in[0] / r);
    for (int i = 1; i < rise; ++i) w_out[i] = (T)((double)w_out[i - 1] + (double)w_in[i] / r);
    for (int i = rise; i < rise + flat; ++i) {
        T prev = w_out[i > 0 ? i - 1 : n - 1];
        T e = w_in[i] - w_in[i - rise];
        w_out[i] = (T)((double)prev + (double)e / r);
    }
    for (int i = rise + flat; i < 2 * rise + flat; ++i) {
        T prev = w_out[i > 0 ? i - 1 : n - 1];
        T e = (w_in[i] - w_in[i - rise]) - w_in[i - rise - flat];
        w_out[i] = (T)((double)prev + (double)e / r);
    }
    for (int i = 2 * rise + flat; i < n; ++i) {
        T prev = w_out[i > 0 ? i - 1 : n - 1];
        T e = ((w_in[i] - w_in[i - rise]) - w_in[i - rise - flat]) + w_in[i - 2 * rise - flat];
        w_out[i] = (T)((double)prev + (double)e / r);
    }
    return 0;
}

/* trap_filters.py:152-227 */
static int FN(row_asym_trap_filter)(const T* w_in, int n, int rise, int flat, int fall, T* w_out) {
    FN(fill_nan)(w_out, n);
    if (FN(row_has_nan)(w_in, n)) return 0;
    if (rise < 0) return ORC_E_TRAP_RISE;
    if (flat < 0) return ORC_E_TRAP_FLAT;
    if (fall < 0) return ORC_E_TRAP_FALL;
    if ((long)rise + flat + fall > n) return ORC_E_TRAP_WIDE;
    if (n == 0) return 0;
    if (rise == 0 || (fall == 0 && rise + flat < n)) return ORC_E_ZERODIV; /* ZeroDivisionError under numba */
    const double r = (double)rise, l = (double)fall;
    w_out[0] = (T)((double)w_in[0] / r);
    for (int i = 1; i < rise; ++i) w_out[i] = (T)((double)w_out[i - 1] + (double)w_in[i] / r);
    for (int i = rise; i < rise + flat; ++i) {
        T prev = w_out[i > 0 ? i - 1 : n - 1];
        w_out[i] = (T)((double)prev + (double)(T)(w_in[i] - w_in[i - rise]) / r);
    }
    for (int i = rise + flat; i < rise + flat + fall; ++i) {
        T prev = w_out[i > 0 ? i - 1 : n - 1];
        w_out[i] = (T)(((double)prev + (double)(T)(w_in[i] - w_in[i - rise]) / r) - (double)w_in[i - rise - flat] / l);
    }
    for (int i = rise + flat + fall; i < n; ++i) {
        T prev = w_out[i > 0 ? i - 1 : n - 1];
        w_out[i] = (T)(((double)prev + (double)(T)(w_in[i] - w_in[i - rise]) / r) -
                       (double)(T)(w_in[i - rise - flat] - w_in[i - rise - flat - fall]) / l);
    }
    return 0;
}

/* fixed_time_pickoff.py:12-125.  t0 = t_in - i_in is T - int64 -> float64; slopes m0/m1 are T differences
 * (halved exactly); x**3 with a literal exponent is numba's unrolled product x*(x*x). */
static int FN(row_fixed_time_pickoff)(const T* w_in, int n, T t_in, int mode, T* a_out) {
    *a_out = (T)NAN;
    if (FN(row_has_nan)(w_in, n) || isnan(t_in)) return 0;
    if (t_in < 0 || t_in > (T)(n - 1)) return 0;
    long i_in = (long)t_in;
    if ((T)i_in == t_in) {
        *a_out = w_in[i_in];
        return 0;
    }
    const double t0 = (double)t_in - (double)i_in;
    const double t1 = 1.0 - t0;
    switch (mode) {
        case 'i': return ORC_E_FTP_INT;
        case 'n': *a_out = (t0 < 0.5) ? w_in[i_in] : w_in[i_in + 1]; return 0;
        case 'f': *a_out = w_in[i_in]; return 0;
        case 'c': *a_out = w_in[i_in + 1]; return 0;
        case 'l': *a_out = (T)(t1 * (double)w_in[i_in] + t0 * (double)w_in[i_in + 1]); return 0;
        case 'h': {
            double m0 = (i_in == 0) ? (double)(T)(w_in[1] - w_in[0]) : (double)(T)(w_in[i_in + 1] - w_in[i_in - 1]) / 2.0;
            double m1 = (i_in == n - 2) ? (double)(T)(w_in[n - 1] - w_in[n - 2]) : (double)(T)(w_in[i_in + 2] - w_in[i_in]) / 2.0;
            double t1_2 = t1 * t1, t1_3 = t1 * t1_2, t0_2 = t0 * t0, t0_3 = t0 * t0_2;
            double v = (((-2.0 * t1_3 + 3.0 * t1_2) * (double)w_in[i_in] + (-2.0 * t0_3 + 3.0 * t0_2) * (double)w_in[i_in + 1]) -
                        (t1_3 - t1_2) * m0) + (t0_3 - t0_2) * m1;
            *a_out = (T)v;
            return 0;
        }
        case 's': {
            double* u = (double*)calloc((size_t)n, sizeof(double));
            double* w2 = (double*)calloc((size_t)n, sizeof(double));
            for (int i = 1; i < n - 1; ++i) {
                double p = 0.5 * w2[i - 1] + 2.0;
                w2[i] = -0.5 / p;
                u[i] = ((double)w_in[i + 1] - 2.0 * (double)w_in[i]) + (double)w_in[i - 1];
                u[i] = (3.0 * u[i] - 0.5 * u[i - 1]) / p;
            }
            for (long i = n - 2; i > i_in - 1; --i) w2[i] = w2[i] * w2[i + 1] + u[i];
            double t1_3 = t1 * (t1 * t1), t0_3 = t0 * (t0 * t0);
            double v = (t1 * (double)w_in[i_in] + t0 * (double)w_in[i_in + 1]) +
                       ((t1_3 - t1) * w2[i_in] + (t0_3 - t0) * w2[i_in + 1]) / 6.0;
            *a_out = (T)v;
            free(u);
            free(w2);
            return 0;
        }
        default: return ORC_E_FTP_MODE;
    }
}

/* time_point_thresh.py:12-92: comparisons only -> bit exact */
static int FN(row_time_point_thresh)(const T* w_in, int n, T a_threshold, T t_start, T walk_forward, T* t_out) {
    *t_out = (T)NAN;
    if (FN(row_has_nan)(w_in, n) || isnan(a_threshold) || isnan(t_start) || isnan(walk_forward)) return 0;
    if (floor((double)t_start) != (double)t_start) return ORC_E_TPT_START_INT;
    if (floor((double)walk_forward) != (double)walk_forward) return ORC_E_TPT_WALK_INT;
    long ts = (long)t_start;
    if (ts < 0 || ts >= n) return ORC_E_TPT_RANGE;
    if ((long)walk_forward == 1) {
        for (long i = ts; i < n - 1; ++i)
            if ((w_in[i] <= a_threshold && a_threshold < w_in[i + 1]) || (w_in[i] >= a_threshold && a_threshold > w_in[i + 1])) {
                *t_out = (T)i;
                return 0;
            }
    } else {
        for (long i = ts; i > 0; --i)
            if ((w_in[i - 1] < a_threshold && a_threshold <= w_in[i]) || (w_in[i - 1] > a_threshold && a_threshold >= w_in[i])) {
                *t_out = (T)i;
                return 0;
            }
    }
    return 0;
}

/* time_point_thresh.py:95-222 interpolated_time_point_thresh.  walk_forward is an int64 argument there (never NaN); t_start is truncated by
 * int(); a start outside the waveform gives NaN, not DSPFatal; the backward walk stops at sample 2 (range(int(t_start), 1, -1)).
 * Typing of mode 'l': the quotient is T / T -> T, i_cross (int64) + T -> float64, the store rounds to T; 'n': int64 + 0.5 -> float64. */
static int FN(row_interpolated_time_point_thresh)(const T* w_in, int n, T a_threshold, T t_start, long walk_forward, int mode, T* t_out) {
    *t_out = (T)NAN;
    if (FN(row_has_nan)(w_in, n) || isnan(a_threshold) || isnan(t_start)) return 0;
    if (t_start < 0 || t_start >= (T)n) return 0;
    long i_cross = -1;
    if (walk_forward > 0) {
        for (long i = (long)t_start; i < n - 1; ++i)
            if ((w_in[i] <= a_threshold && a_threshold < w_in[i + 1]) || (w_in[i] >= a_threshold && a_threshold > w_in[i + 1])) {
                i_cross = i;
                break;
            }
    } else {
        for (long i = (long)t_start; i > 1; --i)
            if ((w_in[i - 1] < a_threshold && a_threshold <= w_in[i]) || (w_in[i - 1] > a_threshold && a_threshold >= w_in[i])) {
                i_cross = i - 1;
                break;
            }
    }
    if (i_cross == -1) return 0;
    switch (mode) {
        case 'i':
        case 'b':
        case 'c': *t_out = (T)i_cross; return 0;
        case 'a':
        case 'f': *t_out = (T)(i_cross + 1); return 0;
        case 'r': {
            const T d0 = a_threshold - w_in[i_cross], d1 = a_threshold - w_in[i_cross + 1];
            *t_out = (T)((d0 < 0 ? -d0 : d0) < (d1 < 0 ? -d1 : d1) ? i_cross : i_cross + 1);
            return 0;
        }
        case 'n': *t_out = (T)((double)i_cross + 0.5); return 0;
        case 'l': {
            const T num = a_threshold - w_in[i_cross], den = w_in[i_cross + 1] - w_in[i_cross];
            const T q = num / den;
            *t_out = (T)((double)i_cross + (double)q);
            return 0;
        }
        default: return ORC_E_FTP_MODE; /* "Unrecognized interpolation mode", raised only once a crossing was found */
    }
}

/* min_max.py:85-140 min_max_norm: the waveform over the larger of |a_min|, |a_max|; unchanged if either is 0; NaN if the waveform has
 * a NaN or neither comparison holds (a NaN bound) */
static int FN(row_min_max_norm)(const T* w_in, int n, T a_min, T a_max, T* w_out) {
    for (int i = 0; i < n; ++i) w_out[i] = (T)NAN;
    if (FN(row_has_nan)(w_in, n)) return 0;
    const T amax = a_max < 0 ? -a_max : a_max, amin = a_min < 0 ? -a_min : a_min;
    if (amax == 0 || amin == 0) {
        for (int i = 0; i < n; ++i) w_out[i] = w_in[i];
    } else if (amax >= amin) {
        for (int i = 0; i < n; ++i) w_out[i] = w_in[i] / amax;
    } else if (amax < amin) {
        for (int i = 0; i < n; ++i) w_out[i] = w_in[i] / amin;
    }
    return 0;
}

/* min_max.py:11-82: strict comparisons, first occurrence wins */
static int FN(row_min_max)(const T* w_in, int n, T* t_min, T* t_max, T* a_min, T* a_max) {
    *t_min = *t_max = *a_min = *a_max = (T)NAN;
    if (FN(row_has_nan)(w_in, n)) return 0;
    int imin = 0, imax = 0;
    for (int i = 0; i < n; ++i) {
        if (w_in[i] < w_in[imin]) imin = i;
        if (w_in[i] > w_in[imax]) imax = i;
    }
    *a_min = w_in[imin];
    *a_max = w_in[imax];
    *t_min = (T)imin;
    *t_max = (T)imax;
    return 0;
}

/* windower.py:12-54: a window of len(w_out) samples starting at int(t0_in) (truncation toward zero); what falls outside the input is NaN */
static int FN(row_windower)(const T* w_in, int n, T t0_in, T* w_out, int m) {
    FN(fill_nan)(w_out, m);
    if (FN(row_has_nan)(w_in, n) || isnan(t0_in)) return 0;
    if (m >= n) return ORC_E_WINDOW_LONG;
    long beg = (long)t0_in;
    if (beg > n) beg = n;
    long end = beg + m;
    if (end < 0) end = 0;
    if (beg < 0) {
        for (long k = 0; k < end; ++k) w_out[m - end + k] = w_in[k];
    } else if (end < n) {
        for (long k = 0; k < m; ++k) w_out[k] = w_in[beg + k];
    } else {
        for (long k = 0; k < n - beg; ++k) w_out[k] = w_in[beg + k];
    }
    return 0;
}

/* moving_windows.py:206-249 avg_current: w_out = (w_in[L:] - w_in[:-L]) / length, all in T (array / T scalar) */
static int FN(row_avg_current)(const T* w_in, int n, T length, T* w_out, int m) {
    FN(fill_nan)(w_out, m);
    if (FN(row_has_nan)(w_in, n)) return 0;
    if (!(length >= 0) || !(length < (T)n)) return ORC_E_AVGCUR_RANGE;
    const int L = (int)length;
    if (L <= 0 || m != n - L) return ORC_E_AVGCUR_RANGE; /* L == 0: NumPy cannot broadcast w_in[0:] - w_in[:-0] */
    for (int k = 0; k < m; ++k) w_out[k] = (T)(w_in[k + L] - w_in[k]) / length;
    return 0;
}

/* trap_filters.py:230-293 trap_pickoff: float64 sums of the two rise-long windows ending at / rise+flat before the pick-off sample,
 * (i_1 - i_2) / rise (float64 / int32 -> float64), rounded by the store */
static int FN(row_trap_pickoff)(const T* w_in, int n, int rise, int flat, T t_pickoff, T* a_out) {
    *a_out = (T)NAN;
    if (FN(row_has_nan)(w_in, n) || isnan(t_pickoff)) return 0;
    if (floor((double)t_pickoff) != (double)t_pickoff) return ORC_E_TPO_INT;
    if (rise < 0) return ORC_E_TRAP_RISE;
    if (flat < 0) return ORC_E_TRAP_FLAT;
    if (2 * (long)rise + flat > n) return ORC_E_TRAP_WIDE;
    const long start = (long)((double)t_pickoff + 1.0);
    if (!(n >= start && start >= 2 * (long)rise + flat)) return 0;
    double i1 = 0.0, i2 = 0.0;
    for (long i = start - rise; i < start; ++i) i1 += (double)w_in[i];
    for (long i = start - 2 * (long)rise - flat; i < start - rise - flat; ++i) i2 += (double)w_in[i];
    if (rise == 0) return ORC_E_ZERODIV;
    *a_out = (T)((i1 - i2) / (double)rise);
    return 0;
}

/* upsampler.py:13-56: every input sample is written to int(upsample) consecutive outputs starting at
 * int(t_in * upsample - floor(upsample / 2)) (int64 * T -> float64; int() truncates toward zero); untouched outputs stay NaN */
static int FN(row_upsampler)(const T* w_in, int n, T upsample, T* w_out, int m) {
    FN(fill_nan)(w_out, m);
    if (FN(row_has_nan)(w_in, n)) return 0;
    if (!(upsample > 0)) return ORC_E_UPSAMPLE;
    const double half = floor((double)upsample / 2.0);
    const long cnt = (long)upsample;
    for (long t = 0; t < n; ++t) {
        long t_out = (long)((double)t * (double)upsample - half);
        for (long k = 0; k < cnt; ++k, ++t_out)
            if (t_out >= 0 && t_out < m) w_out[t_out] = w_in[t];
    }
    return 0;
}

/* moving_windows.py:117-204 moving_window_multi: num_mw moving averages of `length` samples, alternately from the left and from the
 * right (mw_type 0), only left (1) or only right (2).  All in T: w_out[i] = w_out[i-1] + (w_buf[i] - w_buf[i-L]) / length with one
 * rounding per operation and feedback through the output array; the first L samples subtract w_buf[0] instead. */
static int FN(row_moving_window_multi)(const T* w_in, int n, T length, T num_mw, int mw_type, T* w_out) {
    FN(fill_nan)(w_out, n);
    if (FN(row_has_nan)(w_in, n)) return 0;
    if (floor((double)length) != (double)length) return ORC_E_MW_LEN_INT;
    if (floor((double)num_mw) != (double)num_mw) return ORC_E_MW_NUM_INT;
    const long L = (long)length;
    if (L < 0 || L >= n) return ORC_E_MW_LEN_RANGE;
    if ((long)num_mw < 0) return ORC_E_MW_NUM_NEG;
    if ((long)num_mw > 0 && L == 0) return ORC_E_ZERODIV; /* numba error_model='python': (x - x) / 0.0 raises (L >= 1 below) */
    T* buf = (T*)malloc(sizeof(T) * (size_t)n);
    memcpy(buf, w_in, sizeof(T) * (size_t)n);
    for (long p = 0; p < (long)num_mw; ++p) {
        if (((p % 2 == 1) && mw_type == 0) || mw_type == 2) {
            w_out[n - 1] = buf[n - 1];
            for (long i = 1; i < L; ++i) w_out[n - 1 - i] = w_out[n - i] + (T)(buf[n - 1 - i] - w_out[n - 1]) / length;
            for (long i = L; i < n; ++i) w_out[n - 1 - i] = w_out[n - i] + (T)(buf[n - 1 - i] - buf[n - 1 - i + L]) / length;
        } else {
            w_out[0] = buf[0];
            for (long i = 1; i < L; ++i) w_out[i] = w_out[i - 1] + (T)(buf[i] - buf[0]) / length;
            for (long i = L; i < n; ++i) w_out[i] = w_out[i - 1] + (T)(buf[i] - buf[i - L]) / length;
        }
        memcpy(buf, w_out, sizeof(T) * (size_t)n);
    }
    free(buf);
    return 0;
}

/* linear_slope_fit.py:11-91.  PARITY UNPINNED: the reference has no test for it and its nopython typing cannot be executed here
 * (numba is not importable); this follows numba's documented rules as analysed in DESIGN.md: `mean`/`stdev` are 1-element T arrays
 * used as accumulators, `temp = w_in[i] - mean` is a T array expression, `temp / (i + 1)` is array(T) / int64 -> float64 and the
 * in-place add casts back to T; `stdev += temp * (w_in[i] - mean)` is all T; sum_xy, sum_y unify to float64, sum_x, sum_x2 stay
 * int64; `stdev /= isum - 1` divides in float64 and casts back, np.sqrt works in T. */
static int FN(row_linear_slope_fit)(const T* w_in, int n, T* mean, T* stdev, T* slope, T* intercept) {
    *mean = *stdev = *slope = *intercept = (T)NAN;
    if (FN(row_has_nan)(w_in, n)) return 0;
    if (n < 2) return ORC_E_ZERODIV; /* the regression denominator isum*sum_x2 - sum_x**2 is the int64 0: ZeroDivisionError */
    T m = (T)0, s = (T)0;
    double sum_xy = 0.0, sum_y = 0.0;
    long long sum_x = 0, sum_x2 = 0;
    for (long long i = 0; i < n; ++i) {
        const T temp = (T)(w_in[i] - m);
        m = (T)((double)m + (double)temp / (double)(i + 1));
        s = (T)(s + (T)(temp * (T)(w_in[i] - m)));
        sum_x += i;
        sum_x2 += i * i;
        sum_xy += (double)w_in[i] * (double)i;
        sum_y += (double)w_in[i];
    }
    s = (T)((double)s / (double)(n - 1));
    s = (T)sqrt((double)s); /* (correctly rounded either way: sqrtf(x) == (float)sqrt((double)x)) */
    *mean = m;
    *stdev = s;
    *slope = (T)(((double)n * sum_xy - (double)sum_x * sum_y) / (double)((long long)n * sum_x2 - sum_x * sum_x));
    *intercept = (T)((sum_y - (double)sum_x * (double)*slope) / (double)n);
    return 0;
}

/* arithmetic.py:9-62 mean_below_threshold: `total = 0.0` is float64 and stays float64 (float64 += T), `count` int64,
 * result = total / count (float64 / int64 -> float64) rounded by the store; NaN if nothing is below the threshold. */
static int FN(row_mean_below_threshold)(const T* w_in, int n, T threshold, T* result) {
    *result = (T)NAN;
    if (FN(row_has_nan)(w_in, n) || isnan(threshold)) return 0;
    double total = 0.0;
    long count = 0;
    for (int i = 0; i < n; ++i)
        if (w_in[i] < threshold) {
            total += (double)w_in[i];
            ++count;
        }
    if (count == 0) return 0;
    *result = (T)(total / (double)count);
    return 0;
}

/* convolutions.py:14-72 (and :75-119 for the FFT variant, which computes the same sums by another route).
 * np.convolve's float32 summation order is NumPy-internal and not part of the reference, so the
 * oracle accumulates each output in double and rounds once; parity vs the goldens is 1e-6 of max|out|. */
static int FN(row_convolve)(const T* w_in, int n, const T* kern, int m, int mode, T* w_out, int p) {
    FN(fill_nan)(w_out, p);
    if (FN(row_has_nan)(w_in, n)) return 0;
    if (FN(row_has_nan)(kern, m)) return 0;
    if (m > n) return ORC_E_CONV_LONG;
    int full = n + m - 1, start;
    if (mode == 'f') {
        if (p != full) return ORC_E_CONV_OUTLEN;
        start = 0;
    } else if (mode == 'v') {
        if (p != n - m + 1) return ORC_E_CONV_OUTLEN;
        start = m - 1;
    } else if (mode == 's') {
        if (p != n) return ORC_E_CONV_OUTLEN;
        start = (m - 1) / 2;
    } else
        return ORC_E_CONV_MODE;
    for (int o = 0; o < p; ++o) {
        int f = o + start; /* index in the 'full' convolution */
        int k0 = f - (n - 1) > 0 ? f - (n - 1) : 0, k1 = f < m - 1 ? f : m - 1;
        double acc = 0.0;
        for (int k = k0; k <= k1; ++k) acc += (double)w_in[f - k] * (double)kern[k];
        w_out[o] = (T)acc;
    }
    return 0;
}

/* dwt.py:13-81 -> pywt.downcoef(part, w, 'haar'|'db1', level), PyWavelets (third party, unpinned in
 * pyproject.toml:38): periodisation-free 'symmetric' mode, filters dec_lo = [c, c], dec_hi = [-c, c], c = 1/sqrt(2)
 * stored in the data type; each output is fl(fl(f0*x[2k+1]) + fl(f1*x[2k])), level by level, an odd-length level is
 * extended by repeating its last sample.  Checked bit-for-bit against PyWavelets 1.1.1 outputs (tests/golden/dwt.npz). */
static int FN(row_dwt_haar)(const T* w_in, int n, int level, int part, T* w_out, int p) {
    FN(fill_nan)(w_out, p);
    if (level <= 0) return ORC_E_DWT_LEVEL;
    if (FN(row_has_nan)(w_in, n)) return 0;
    const T c = (T)0.7071067811865476;
    T* cur = (T*)malloc(sizeof(T) * (size_t)(n + 2));
    T* nxt = (T*)malloc(sizeof(T) * (size_t)(n + 2));
    memcpy(cur, w_in, sizeof(T) * (size_t)n);
    int len = n, rc = 0;
    for (int l = 0; l < level; ++l) {
        if (len & 1) {
            cur[len] = cur[len - 1];
            ++len;
        }
        int half = len / 2, last = (l == level - 1);
        for (int k = 0; k < half; ++k) {
            T hi = cur[2 * k + 1], lo = cur[2 * k];
            if (last && part == 'd')
                nxt[k] = (T)((T)(-c * hi) + (T)(c * lo));
            else
                nxt[k] = (T)((T)(c * hi) + (T)(c * lo));
        }
        T* t = cur;
        cur = nxt;
        nxt = t;
        len = half;
    }
    if (len != p)
        rc = ORC_E_DWT_OUTLEN;
    else
        memcpy(w_out, cur, sizeof(T) * (size_t)p);
    free(cur);
    free(nxt);
    return rc;
}

/* ------------------------------------------------------------------ batch entry points (rows x len, C-contiguous) */
#define ROWLOOP(call)                       \
    int rc = 0;                             \
    for (long r = 0; r < n_wf; ++r) {       \
        int e = (call);                     \
        if (e && !rc) {                     \
            rc = e;                         \
            if (err_row) *err_row = r;      \
        }                                   \
    }                                       \
    return rc;

#define PV(p, r) ((p##_stride) ? (p)[(r)] : (p)[0]) /* per-row vector or broadcast constant */

int FN(orc_bl_subtract)(const T* in, long n_wf, int len, const T* bl, int bl_stride, T* out, long* err_row) {
    ROWLOOP(FN(row_bl_subtract)(in + r * len, len, PV(bl, r), out + r * len))
}
int FN(orc_pole_zero)(const T* in, long n_wf, int len, T tau, T* out, long* err_row) {
    ROWLOOP(FN(row_pole_zero)(in + r * len, len, tau, out + r * len))
}
int FN(orc_double_pole_zero)(const T* in, long n_wf, int len, T tau1, T tau2, T frac, T* out, long* err_row) {
    ROWLOOP(FN(row_double_pole_zero)(in + r * len, len, tau1, tau2, frac, out + r * len))
}
int FN(orc_trap_filter)(const T* in, long n_wf, int len, int rise, int flat, T* out, long* err_row) {
    ROWLOOP(FN(row_trap_filter)(in + r * len, len, rise, flat, out + r * len))
}
int FN(orc_trap_norm)(const T* in, long n_wf, int len, int rise, int flat, T* out, long* err_row) {
    ROWLOOP(FN(row_trap_norm)(in + r * len, len, rise, flat, out + r * len))
}
int FN(orc_asym_trap_filter)(const T* in, long n_wf, int len, int rise, int flat, int fall, T* out, long* err_row) {
    ROWLOOP(FN(row_asym_trap_filter)(in + r * len, len, rise, flat, fall, out + r * len))
}
int FN(orc_fixed_time_pickoff)(const T* in, long n_wf, int len, const T* t_in, int t_in_stride, int mode, T* out, long* err_row) {
    ROWLOOP(FN(row_fixed_time_pickoff)(in + r * len, len, PV(t_in, r), mode, out + r))
}
int FN(orc_time_point_thresh)(const T* in, long n_wf, int len, const T* thr, int thr_stride, const T* t_start, int t_start_stride,
                              T walk_forward, T* out, long* err_row) {
    ROWLOOP(FN(row_time_point_thresh)(in + r * len, len, PV(thr, r), PV(t_start, r), walk_forward, out + r))
}
int FN(orc_interpolated_time_point_thresh)(const T* in, long n_wf, int len, const T* thr, int thr_stride, const T* t_start, int t_start_stride,
                                           long walk_forward, int mode, T* out, long* err_row) {
    ROWLOOP(FN(row_interpolated_time_point_thresh)(in + r * len, len, PV(thr, r), PV(t_start, r), walk_forward, mode, out + r))
}
int FN(orc_min_max)(const T* in, long n_wf, int len, T* t_min, T* t_max, T* a_min, T* a_max, long* err_row) {
    ROWLOOP(FN(row_min_max)(in + r * len, len, t_min + r, t_max + r, a_min + r, a_max + r))
}
int FN(orc_min_max_norm)(const T* in, long n_wf, int len, const T* a_min, int a_min_stride, const T* a_max, int a_max_stride, T* out,
                         long* err_row) {
    ROWLOOP(FN(row_min_max_norm)(in + r * len, len, PV(a_min, r), PV(a_max, r), out + r * len))
}
int FN(orc_windower)(const T* in, long n_wf, int len, const T* t0, int t0_stride, T* out, int m, long* err_row) {
    ROWLOOP(FN(row_windower)(in + r * len, len, PV(t0, r), out + r * (long)m, m))
}
int FN(orc_avg_current)(const T* in, long n_wf, int len, T length, T* out, int m, long* err_row) {
    ROWLOOP(FN(row_avg_current)(in + r * len, len, length, out + r * (long)m, m))
}
int FN(orc_trap_pickoff)(const T* in, long n_wf, int len, int rise, int flat, const T* tp, int tp_stride, T* out, long* err_row) {
    ROWLOOP(FN(row_trap_pickoff)(in + r * len, len, rise, flat, PV(tp, r), out + r))
}
int FN(orc_upsampler)(const T* in, long n_wf, int len, T upsample, T* out, int m, long* err_row) {
    ROWLOOP(FN(row_upsampler)(in + r * len, len, upsample, out + r * (long)m, m))
}
int FN(orc_moving_window_multi)(const T* in, long n_wf, int len, T length, T num_mw, int mw_type, T* out, long* err_row) {
    ROWLOOP(FN(row_moving_window_multi)(in + r * len, len, length, num_mw, mw_type, out + r * len))
}
int FN(orc_linear_slope_fit)(const T* in, long n_wf, int len, T* mean, T* stdev, T* slope, T* intercept, long* err_row) {
    ROWLOOP(FN(row_linear_slope_fit)(in + r * len, len, mean + r, stdev + r, slope + r, intercept + r))
}
int FN(orc_mean_below_threshold)(const T* in, long n_wf, int len, const T* thr, int thr_stride, T* out, long* err_row) {
    ROWLOOP(FN(row_mean_below_threshold)(in + r * len, len, PV(thr, r), out + r))
}
int FN(orc_convolve)(const T* in, long n_wf, int len, long in_row_stride, const T* kern, int m, int mode, T* out, int p, long* err_row) {
    ROWLOOP(FN(row_convolve)(in + r * in_row_stride, len, kern, m, mode, out + r * (long)p, p))
}
int FN(orc_dwt_haar)(const T* in, long n_wf, int len, int level, int part, T* out, int p, long* err_row) {
    ROWLOOP(FN(row_dwt_haar)(in + r * len, len, level, part, out + r * (long)p, p))
}

#undef ROWLOOP
#undef PV
#undef FN
#undef CAT
#undef CAT_
