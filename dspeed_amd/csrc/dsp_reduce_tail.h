// dsp_reduce_tail.h -- what the kernels that read per-event values off rows share (dsp_reduce.hip: rows in HBM; dsp_fir_runs.hip: rows the
// kernel itself just filtered): first-occurrence extremes (min_max.py:73-77), and everything that follows once a wavefront has seen its
// whole row -- the exchange across the lanes, time_point_thresh walks from a constant sample or from an extreme (time_point_thresh.py:12-92),
// samples at constant indices (fixed_time_pickoff.py:68-80), the stores.
#pragma once
#include <hip/hip_runtime.h>

#include "dsp_program.h"
#include "dsp_wave.h"

namespace {

struct Extremes {
    float vmin, vmax;
    int imin, imax;
    bool nan;
};

// strict comparisons in ascending index order: the first occurrence stays (min_max.py:73-77)
__device__ __forceinline__ void take(Extremes& e, float v, int i, bool live) {
    const bool lt = live && v < e.vmin, gt = live && v > e.vmax;
    e.vmin = lt ? v : e.vmin;
    e.imin = lt ? i : e.imin;
    e.vmax = gt ? v : e.vmax;
    e.imax = gt ? i : e.imax;
    e.nan |= live && (v != v);
}

// every lane holds the extremes of the samples it saw; `w`: the row (readable by every lane: in HBM or the caches), n samples
// (ARGS: ReduceArgs wherever the kernel holds it -- a by-value argument or the kernel-argument segment)
template <typename IN, typename ARGS>
__device__ __forceinline__ void reduce_finish(ARGS& A, int64_t row, Extremes e, const IN* w, int n, int lane, int* err = nullptr) {
    // across the wavefront: smaller value, then smaller index
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float ovmin = __shfl_xor(e.vmin, m), ovmax = __shfl_xor(e.vmax, m);
        const int oimin = __shfl_xor(e.imin, m), oimax = __shfl_xor(e.imax, m);
        const bool tmin = ovmin < e.vmin || (ovmin == e.vmin && oimin < e.imin);
        const bool tmax = ovmax > e.vmax || (ovmax == e.vmax && oimax < e.imax);
        e.vmin = tmin ? ovmin : e.vmin;
        e.imin = tmin ? oimin : e.imin;
        e.vmax = tmax ? ovmax : e.vmax;
        e.imax = tmax ? oimax : e.imax;
    }
    const bool any_nan = wave_any(e.nan);
    // time_point_thresh (time_point_thresh.py:12-92) from a sample known by now: 64 consecutive samples per step, away from the start, until a
    // step holds a crossing -- the row was just read, the walk finds it in the caches.  Comparisons only.
    constexpr int WR = 8;  // steps per round of a long walk
    float walked[DSP_REDUCE_WALKS];
#pragma unroll
    for (int k = 0; k < DSP_REDUCE_WALKS; ++k) {
        walked[k] = quiet_nan<float>();
        if (k >= A.n_walks) continue;  // (uniform)
        float thr = A.walk_thr[k] ? A.walk_thr[k][row * A.walk_thr_stride[k]] : A.walk_thr_const[k];
        if (A.walk_thr_scaled[k]) thr = thr * A.walk_thr_factor[k];
        const int from = A.walk_from[k];
        int ts = from == 1 ? e.imin : (from == 2 ? e.imax : A.walk_start[k]);
        if (from >= 3) {  // a start that is a per-event value: the processor's own checks (time_point_thresh.py:60-74)
            float ts_f = from == 3 ? A.walk_ts[k][row * A.walk_ts_stride[k]] : quiet_nan<float>();
#pragma unroll
            for (int j = 0; j < k; ++j)
                if (from == 4 + j) ts_f = walked[j];
            if (any_nan || thr != thr || ts_f != ts_f) continue;
            int code = 0;
            if (floorf(ts_f) != ts_f) code = DSP_E_TPT_START_INT;
            else if (ts_f < 0.0f || ts_f >= (float)n) code = DSP_E_TPT_RANGE;
            if (code) {
                if (err && lane == 0 && atomicCAS(&err[0], 0, code) == 0) {
                    err[1] = (int)(row & 0xffffffffll);
                    err[2] = (int)(row >> 32);
                }
                continue;
            }
            ts = (int)ts_f;
        }
        if (any_nan || thr != thr) continue;
        int found = -1;
        // the first two steps one at a time (where the walks of real pulses end), then eight steps a round with their sixteen loads in flight
        if (A.walk_forward[k]) {  // smallest i in [ts, n - 2] with w[i] <= thr < w[i+1] or w[i] >= thr > w[i+1]
            int b = ts;
            for (int s = 0; s < 2 && b <= n - 2 && found < 0; ++s, b += 64) {
                const int i = b + lane;
                bool hit = false;
                if (i <= n - 2) {
                    const float cur = (float)w[i], nxt = (float)w[i + 1];
                    hit = (cur <= thr && thr < nxt) || (cur >= thr && thr > nxt);
                }
                const unsigned long long m = __ballot(hit);
                if (m) found = b + __builtin_ctzll(m);
            }
            for (; b <= n - 2 && found < 0; b += 64 * WR) {
                float cur[WR], nxt[WR];
#pragma unroll
                for (int q = 0; q < WR; ++q) {
                    const int i = b + 64 * q + lane, ic = i <= n - 2 ? i : n - 2;
                    cur[q] = (float)w[ic];
                    nxt[q] = (float)w[ic + 1];
                }
#pragma unroll
                for (int q = 0; q < WR; ++q) {
                    const int i = b + 64 * q + lane;
                    const bool hit = i <= n - 2 && ((cur[q] <= thr && thr < nxt[q]) || (cur[q] >= thr && thr > nxt[q]));
                    const unsigned long long m = __ballot(hit);
                    if (m && found < 0) found = b + 64 * q + __builtin_ctzll(m);
                }
            }
        } else {  // largest i in [1, ts] with w[i-1] < thr <= w[i] or w[i-1] > thr >= w[i]
            int b = ts;
            for (int s = 0; s < 2 && b >= 1 && found < 0; ++s, b -= 64) {
                const int i = b - lane;
                bool hit = false;
                if (i >= 1) {
                    const float cur = (float)w[i], prv = (float)w[i - 1];
                    hit = (prv < thr && thr <= cur) || (prv > thr && thr >= cur);
                }
                const unsigned long long m = __ballot(hit);
                if (m) found = b - __builtin_ctzll(m);
            }
            for (; b >= 1 && found < 0; b -= 64 * WR) {
                float cur[WR], prv[WR];
#pragma unroll
                for (int q = 0; q < WR; ++q) {
                    const int i = b - 64 * q - lane, ic = i >= 1 ? i : 1;
                    cur[q] = (float)w[ic];
                    prv[q] = (float)w[ic - 1];
                }
#pragma unroll
                for (int q = 0; q < WR; ++q) {
                    const int i = b - 64 * q - lane;
                    const bool hit = i >= 1 && ((prv[q] < thr && thr <= cur[q]) || (prv[q] > thr && thr >= cur[q]));
                    const unsigned long long m = __ballot(hit);
                    if (m && found < 0) found = b - 64 * q - __builtin_ctzll(m);
                }
            }
        }
        if (found >= 0) walked[k] = (float)found;
    }
    if (lane == 0) {
        const float nanv = quiet_nan<float>();
        // min_max: NaN anywhere -> four NaNs (min_max.py:62-68); numpy.amax of a row with a NaN is NaN
        const float v[5] = {any_nan ? nanv : (float)e.imin, any_nan ? nanv : (float)e.imax, any_nan ? nanv : e.vmin, any_nan ? nanv : e.vmax,
                            any_nan ? nanv : e.vmax};
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (A.out[k]) ((float*)A.out[k])[row * A.out_stride[k]] = v[k];
#pragma unroll
        for (int k = 0; k < DSP_REDUCE_WALKS; ++k)
            if (A.walk_out[k]) ((float*)A.walk_out[k])[row * A.walk_stride[k]] = walked[k];
#pragma unroll
        for (int k = 0; k < DSP_REDUCE_PICKS; ++k) {
            if (!A.pick_out[k]) continue;
            // a sample at a constant index: fixed_time_pickoff's NaN rule (a NaN anywhere, or a time outside the waveform) or the plain sample
            float s = nanv;
            if (A.pick_at[k] >= 0 && A.pick_at[k] < n && !(A.pick_rule[k] && any_nan)) s = (float)w[A.pick_at[k]];
            ((float*)A.pick_out[k])[row * A.pick_stride[k]] = s;
        }
    }
}

}  // namespace
