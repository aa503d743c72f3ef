// dsp_fir_runs.hip -- convolve_wf (processors/convolutions.py:14-72) with a PIECEWISE CONSTANT kernel, and what a recipe reads off the filtered
// waveform, in one pass over the rows.
//
// The t0 filter of the Ge recipes (processors/kernels.py t0_filter: a ramp of `rise` taps, then `fall` equal ones; 8 + 125 in
// icpc-dsp-config.json) is 133 taps on 8192 samples: 1.1 M multiply-adds per waveform, 3.4 ms per 131 072 rows on the matrix cores (the
// tile of dsp_fir_f16.hip is issue-bound at that length) plus 0.7 ms for min_max and the threshold walk that read the result back.  But
// a run of equal taps multiplies a *sum* of consecutive samples, and sums of consecutive samples are differences of prefix sums: with
// P[i] = x[0] + .. + x[i-1] and the kernel's breakpoints t[0] = 0 < t[1] < .. < t[B] = m (kernel[j] constant on [t[b], t[b+1])),
//
//     np.convolve(x, kernel)[f] = sum_j kernel[j] x[f - j] = sum_b (v[b] - v[b-1]) P[f + 1 - t[b]]        (v[-1] = v[B] = 0)
//
// -- B + 1 = 10 terms per output instead of 133, in float64 (P of 8192 float32 samples is exact or within 2^-53 of it; the float32
// np.convolve of the reference carries 133 roundings: this form sits closer to the oracle's float64 sums than any float32 product).
// A wavefront walks its row in memory order, 512 samples a step: prefix sums as in dsp_pz.hip (local prefix of 8 samples per lane,
// one scan across the wavefront, a carry from step to step), the sums of the last m + 512 samples in LDS, every lane 8 outputs per step
// (lane + 64 u: consecutive lanes read consecutive sums -- no bank conflict), first-occurrence extremes as the outputs appear, and at
// the end of the row the walks / pick-offs of dsp_reduce_tail.h on what was just written.  When nothing else reads the filtered waveform
// it never reaches HBM: the wavefront keeps it in a scratch row of its own that stays in the caches.  HBM-bound by the bytes of the rows
// read (and written, when kept).
//
// Rows with a NaN become NaN (convolutions.py:40-43); rows with an infinity (the prefix sums are useless from there on) and kernels that
// are not piecewise constant after all (the taps are a binding: dsp_fir_runs_prep_kernel looks at them ahead of every launch) are done
// tap by tap, as dsp_fir_fixup_kernel does them.
#include <hip/hip_runtime.h>

#include "dsp_program.h"
#include "dsp_reduce_tail.h"
#include "dsp_wave.h"

#define FR_GLOBAL __attribute__((address_space(1)))
#define FR_KARG __attribute__((address_space(4)))
#define FR_LDS __attribute__((address_space(3)))

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int STEP = 512;  // samples per step: 8 per lane

// One wavefront: the breakpoints of the kernel and their weights.  A breakpoint is a tap index t in [0, m] where kernel[t] differs from
// kernel[t - 1] (zeros in front of the kernel and behind it).
__global__ void __launch_bounds__(64) dsp_fir_runs_prep_kernel(const float* taps, int m, FirRunsTable* tab) {
    const int lane = lane_id();
    int n_break = 0;
    bool bad = false, nan = false;
    for (int t0 = 0; t0 <= m; t0 += 64) {
        const int t = t0 + lane;
        const float cur = t < m ? taps[t] : 0.0f, prev = (t > 0 && t <= m) ? taps[t - 1] : 0.0f;
        bad |= !(cur - cur == 0.0f);
        nan |= cur != cur;
        const bool brk = t <= m && cur != prev;
        const unsigned long long mask = __ballot(brk);
        const int rank = n_break + __popcll(mask & ((1ull << lane) - 1ull));
        if (brk && rank < DSP_FIR_RUNS_MAX + 1) {
            tab->t[rank] = t;
            tab->weight[rank] = (double)cur - (double)prev;
        }
        n_break += __popcll(mask);
    }
    bad = wave_any(bad);
    nan = wave_any(nan);
    if (lane == 0) {
        tab->n_break = (bad || n_break > DSP_FIR_RUNS_MAX + 1) ? 0 : n_break;
        tab->taps_nan = nan ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256) dsp_fir_runs_kernel(FirRunsArgs A_, int64_t n_wf) {
    const FR_KARG FirRunsArgs& A = *(const FR_KARG FirRunsArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    extern __shared__ __attribute__((aligned(16))) double fr_smem[];
    const int lane = lane_id();
    const int wib = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int n = A.n, m = A.m, p = A.p, start = A.start;
    const bool has_red = A.has_red != 0;
    const int mp = (m + 63) & ~63;  // the sums carried from step to step: positions [0, mp); the step's own: [mp, mp + 512)
    FR_LDS double* buf = (FR_LDS double*)fr_smem + wib * (mp + STEP);
    // the kernel's breakpoints: entry b in lane b's registers, read out with v_readlane where the loop over them needs it (a load per
    // breakpoint and step -- scalar or LDS -- would sit in front of the step's LDS reads and wait out its latency every time)
    const FR_KARG FirRunsTable& T = *(const FR_KARG FirRunsTable*)(unsigned long long)A.table;
    const int n_break = T.n_break;
    int tab_t = 0;
    double tab_w = 0.0;  // (lanes from n_break on: weight 0 at a position inside the window -- the loop takes the breakpoints two at a time)
    if (lane < n_break) {
        const FR_GLOBAL FirRunsTable* tg = (const FR_GLOBAL FirRunsTable*)A.table;
        tab_t = tg->t[lane];
        tab_w = tg->weight[lane];
    }
    const int64_t wave = (int64_t)blockIdx.x * 4 + wib, n_waves = (int64_t)gridDim.x * 4;
    const int n_steps = (p + start + STEP - 1) / STEP;  // output c of step g, position j: c = 512 g - start + j

    for (int64_t row = wave; row < n_wf; row += n_waves) {
        const FR_GLOBAL float* x = (const FR_GLOBAL float*)A.wf + row * A.wf_stride + A.wf_offset;
        FR_GLOBAL float* outp = (FR_GLOBAL float*)A.out + (A.keep ? row : wave) * A.out_stride;
        Extremes e;
        e.vmin = __builtin_inff();
        e.vmax = -__builtin_inff();
        e.imin = e.imax = 0;
        e.nan = false;
        double carry = 0.0;
        if (n_break > 0) {
            for (int q = lane; q < mp; q += 64) buf[q] = 0.0;  // P of the samples in front of the row
            auto fetch = [&](f4 (&dst)[2], int g) {
                int at = g * STEP + lane * 8;
                at = at < n ? at : 0;  // (a lane beyond the row's end asks for something inside it and ignores it)
                const FR_GLOBAL f4* src = (const FR_GLOBAL f4*)(x + at);
                dst[0] = __builtin_nontemporal_load(src);
                dst[1] = __builtin_nontemporal_load(src + 1);
            };
            f4 cur[2], nxt[2];
            fetch(cur, 0);
            for (int g = 0; g < n_steps; ++g) {
                fetch(nxt, g + 1);
                // ---- the step's prefix sums: P[512 g + 8 lane + u + 1] at position mp + 8 lane + u
                const bool live = g * STEP + lane * 8 < n;  // (n is a multiple of 8)
                double pl[8], run = 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float v = live ? (u < 4 ? cur[0][u] : cur[1][u - 4]) : 0.0f;
                    run += (double)v;
                    pl[u] = run;
                }
                const double inc = wave_scan_add(run);
                const double base = wave_prev(inc) + carry;
                typedef double d2 __attribute__((ext_vector_type(2)));
                FR_LDS d2* dst = (FR_LDS d2*)(buf + mp + lane * 8);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    d2 v;
                    v[0] = base + pl[2 * u];
                    v[1] = base + pl[2 * u + 1];
                    dst[u] = v;
                }
                carry += readlane(inc, 63);
                wave_sync();
                // ---- 8 outputs per lane: c = 512 g - start + lane + 64 u reads P[c + start + 1 - t] = position lane + 64 u + mp - t
                double acc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = 0.0;
                for (int b = 0; b < n_break; b += 2) {
                    const double w0 = readlane(tab_w, b), w1 = readlane(tab_w, b + 1);
                    const FR_LDS double* s0 = buf + (lane + mp - __builtin_amdgcn_readlane(tab_t, b));
                    const FR_LDS double* s1 = buf + (lane + mp - __builtin_amdgcn_readlane(tab_t, b + 1));
                    double p0[8], p1[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        p0[u] = s0[64 * u];
                        p1[u] = s1[64 * u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[u] = __builtin_fma(w1, p1[u], __builtin_fma(w0, p0[u], acc[u]));
                }
                const int c0 = g * STEP - start + lane;
                float y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) y[u] = (float)acc[u];
                if (g * STEP - start >= 0 && g * STEP - start + STEP <= p) {  // (uniform) a step inside the output: no bounds to look at
#pragma unroll
                    for (int u = 0; u < 8; ++u) outp[c0 + 64 * u] = y[u];
                    if (has_red) {
                        // (finite samples give finite sums: no NaN to look for here -- rows with a NaN or an infinity are redone below)
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const bool lt = y[u] < e.vmin, gt = y[u] > e.vmax;
                            e.vmin = lt ? y[u] : e.vmin;
                            e.imin = lt ? c0 + 64 * u : e.imin;
                            e.vmax = gt ? y[u] : e.vmax;
                            e.imax = gt ? c0 + 64 * u : e.imax;
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + 64 * u;
                        const bool valid = c >= 0 && c < p;
                        if (valid) outp[c] = y[u];
                        if (has_red) take(e, y[u], c, valid);
                    }
                }
                // ---- the last mp sums move to the front
                double keep[DSP_FIR_RUNS_MAX_TAPS / 64];
#pragma unroll
                for (int k = 0; k < DSP_FIR_RUNS_MAX_TAPS / 64; ++k)
                    if (k * 64 < mp) keep[k] = buf[STEP + k * 64 + lane];
                wave_sync();
#pragma unroll
                for (int k = 0; k < DSP_FIR_RUNS_MAX_TAPS / 64; ++k)
                    if (k * 64 < mp) buf[k * 64 + lane] = keep[k];
                wave_sync();
                cur[0] = nxt[0];
                cur[1] = nxt[1];
            }
        }
        // ---- the rows the sums cannot do: a NaN or an infinity among the samples (the row's total tells: once a sum is not finite it stays so)
        if (n_break == 0 || !(carry - carry == 0.0)) {
            bool has_nan = T.taps_nan != 0;
            for (int i = lane; i < n; i += 64) has_nan |= x[i] != x[i];
            has_nan = wave_any(has_nan);
            e.vmin = __builtin_inff();
            e.vmax = -__builtin_inff();
            e.imin = e.imax = 0;
            e.nan = has_nan;
            if (has_nan) {
                for (int c = lane; c < p; c += 64) outp[c] = quiet_nan<float>();
            } else {
                const FR_GLOBAL float* kp = (const FR_GLOBAL float*)A.taps;
                const int d = m - 1 - start;  // output c sums the samples c - d .. c - d + m - 1
                for (int c = lane; c < p; c += 64) {
                    float s = 0.0f;
                    for (int t = 0; t < m; ++t) {
                        const int i = c - d + t;
                        if (i >= 0 && i < n) s = __builtin_fmaf(x[i], kp[m - 1 - t], s);
                    }
                    outp[c] = s;
                    take(e, s, c, true);
                }
            }
        }
        if (has_red) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the walks read what other lanes of this wavefront just wrote (same CU, same L1: no write-back, no invalidate)
            reduce_finish<float>(A.red, row, e, (const float*)outp, p, lane);  // (its loads are back before the next row's stores are issued)
        }
    }
}

}  // namespace

extern "C" int dsp_internal_fir_runs_lds_bytes(int m) { return 4 * (((m + 63) & ~63) + STEP) * (int)sizeof(double); }

extern "C" int dsp_internal_launch_fir_runs(const FirRunsArgs* A, FirRunsTable* table, int64_t n_wf, int blocks, hipStream_t stream) {
    if (n_wf <= 0 || A->p <= 0) return 0;
    hipLaunchKernelGGL(dsp_fir_runs_prep_kernel, dim3(1), dim3(64), 0, stream, A->taps, A->m, table);
    hipLaunchKernelGGL(dsp_fir_runs_kernel, dim3((unsigned)blocks), dim3(256), (size_t)dsp_internal_fir_runs_lds_bytes(A->m), stream, *A, n_wf);
    return (int)hipGetLastError();
}

extern "C" const char* dsp_internal_fir_runs_kernel_name() { return "dsp_fir_runs_kernel"; }
