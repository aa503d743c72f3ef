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
// A wavefront walks its row in memory order, 512 samples a step: prefix sums as in dsp_pz.hip (local prefix of 8 samples per lane, one
// scan across the wavefront, a carry from step to step), the sums of the last m + 512 samples in LDS, every lane 8 CONSECUTIVE outputs per
// step.  Consecutive outputs at consecutive breakpoints (the ramp: a breakpoint at every tap) read overlapping sums: a group of G
// breakpoints needs 8 + G - 1 sums per lane, read once into registers, instead of 8 G -- LDS bandwidth is what bounds this kernel (a sum
// per output and breakpoint: 82 % LDS-busy, 2.3 ms per 131 072 rows of 8192; with the groups 16 + 8 reads per lane and step instead of
// 80).  Lanes 64 bytes apart would meet on the same banks, so sum i lives at element i + i / 8 (72 bytes from lane to lane: every bank
// once per half wavefront).  First-occurrence extremes as the outputs appear, and at the end of the row the walks / pick-offs of
// dsp_reduce_tail.h on what was just written.  When nothing else reads the filtered waveform it never reaches HBM: the wavefront keeps it
// in a scratch row of its own that stays in the caches.
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
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // lane 0 reads what the others wrote
    if (lane == 0) {
        const bool ok = !(bad || n_break > DSP_FIR_RUNS_MAX + 1);
        tab->n_break = ok ? n_break : 0;
        tab->taps_nan = nan ? 1 : 0;
        int n_groups = 0;
        for (int b = 0; ok && b < n_break;) {  // breakpoints at consecutive taps go together
            int cnt = 1;
            while (b + cnt < n_break && cnt < DSP_FIR_RUNS_GROUP && tab->t[b + cnt] == tab->t[b] + cnt) ++cnt;
            tab->first[n_groups] = b;
            tab->count[n_groups] = cnt;
            ++n_groups;
            b += cnt;
        }
        tab->n_groups = n_groups;
    }
}

// A group of G breakpoints at the taps t, t + 1, .. t + G - 1 (weights w[0 .. G-1]) on the lane's 8 consecutive outputs: output r and
// breakpoint k read window element 8 lane + r + mp - t - k -- 8 + G - 1 different ones, win[r + G - 1 - k].  `row9`: the window + 9 lane
// elements (the padded position of element 8 lane), s = mp - t - (G - 1) >= 0 the first one needed, relative to that (uniform).
template <int G>
__device__ __forceinline__ void group_of(double (&acc)[8], const FR_LDS double* row9, int s, double tab_w, int bi) {
    double win[8 + G - 1];
#pragma unroll
    for (int j = 0; j < 8 + G - 1; ++j) {
        const int q = s + j;
        win[j] = row9[q + (q >> 3)];
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const double w = readlane(tab_w, bi + k);
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = __builtin_fma(w, win[r + (G - 1) - k], acc[r]);
    }
}

__global__ void __launch_bounds__(256) dsp_fir_runs_kernel(FirRunsArgs A_, int64_t n_wf, int* err) {
    const FR_KARG FirRunsArgs& A = *(const FR_KARG FirRunsArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    extern __shared__ __attribute__((aligned(16))) double fr_smem[];
    const int lane = lane_id();
    const int wib = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int n = A.n, m = A.m, p = A.p, start = A.start;
    const bool has_red = A.has_red != 0;
    const int mp = (m + 63) & ~63;  // the sums carried from step to step: window elements [0, mp); the step's own: [mp, mp + 512)
    constexpr int PSTEP = STEP + STEP / 8;
    const int pmp = mp + mp / 8;    // (element i at position i + i / 8)
    FR_LDS double* buf = (FR_LDS double*)fr_smem + wib * (pmp + PSTEP);
    const FR_LDS double* row9 = buf + 9 * lane;
    // the kernel's breakpoints: entry b in lane b's registers, read out with v_readlane where the loop over them needs it (a load per
    // breakpoint and step -- scalar or LDS -- would sit in front of the step's LDS reads and wait out its latency every time)
    const FR_KARG FirRunsTable& T = *(const FR_KARG FirRunsTable*)(unsigned long long)A.table;
    const int n_break = T.n_break;
    const int n_groups = T.n_groups;
    int tab_t = 0, tab_first = 0, tab_count = 1;
    double tab_w = 0.0;
    if (lane < n_break) {
        const FR_GLOBAL FirRunsTable* tg = (const FR_GLOBAL FirRunsTable*)A.table;
        tab_t = tg->t[lane];
        tab_w = tg->weight[lane];
        if (lane < n_groups) {
            tab_first = tg->first[lane];
            tab_count = tg->count[lane];
        }
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
            for (int q = lane; q < pmp; q += 64) buf[q] = 0.0;  // P of the samples in front of the row
            auto fetch = [&](f4 (&dst)[2], int g) {
                int at = g * STEP + lane * 8;
                at = at < n ? at : 0;  // (a lane beyond the row's end asks for something inside it and ignores it)
                const FR_GLOBAL f4* src = (const FR_GLOBAL f4*)(x + at);
                dst[0] = __builtin_nontemporal_load(src);
                dst[1] = __builtin_nontemporal_load(src + 1);
            };
            // three sets of registers for the samples, filled in turn: the samples of step g + 2 are asked for when step g begins -- with one
            // step of lead the loop waited for memory (16 wavefronts a CU x 2 kB in flight carry about 2 TB/s at HBM's latency under load:
            // every form of this kernel took the same 2.1 ms per 131 072 rows, whatever its arithmetic)
            auto step = [&](int g, const f4 (&cur)[2]) __attribute__((always_inline)) {
                // ---- the step's prefix sums: P[512 g + 8 lane + u + 1] at position mp + 8 lane + u
                double pl[8], run = 0.0;
                if (g * STEP + STEP <= n) {  // (uniform) every lane's eight samples lie inside the row
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        run += (double)(u < 4 ? cur[0][u] : cur[1][u - 4]);
                        pl[u] = run;
                    }
                } else {
                    const bool live = g * STEP + lane * 8 < n;  // (n is a multiple of 8)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float v = live ? (u < 4 ? cur[0][u] : cur[1][u - 4]) : 0.0f;
                        run += (double)v;
                        pl[u] = run;
                    }
                }
                const double inc = wave_scan_add(run);
                const double base = wave_prev(inc) + carry;
                FR_LDS double* dst = buf + pmp + lane * 9;
#pragma unroll
                for (int u = 0; u < 8; ++u) dst[u] = base + pl[u];
                carry += readlane(inc, 63);
                wave_sync();
                // ---- 8 consecutive outputs per lane: c = 512 g - start + 8 lane + r reads P[c + start + 1 - t] = window element 8 lane + r + mp - t
                double acc[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = 0.0;
                for (int gi = 0; gi < n_groups; ++gi) {
                    const int bi = __builtin_amdgcn_readlane(tab_first, gi), G = __builtin_amdgcn_readlane(tab_count, gi);
                    const int s = mp - __builtin_amdgcn_readlane(tab_t, bi) - (G - 1);
                    switch (G) {
                        case 1: group_of<1>(acc, row9, s, tab_w, bi); break;
                        case 2: group_of<2>(acc, row9, s, tab_w, bi); break;
                        case 3: group_of<3>(acc, row9, s, tab_w, bi); break;
                        case 4: group_of<4>(acc, row9, s, tab_w, bi); break;
                        case 5: group_of<5>(acc, row9, s, tab_w, bi); break;
                        case 6: group_of<6>(acc, row9, s, tab_w, bi); break;
                        case 7: group_of<7>(acc, row9, s, tab_w, bi); break;
                        default: group_of<8>(acc, row9, s, tab_w, bi); break;
                    }
                }
                const int c0 = g * STEP - start + 8 * lane;
                float y[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) y[r] = (float)acc[r];
                if (g * STEP - start >= 0 && g * STEP - start + STEP <= p) {  // (uniform) a step inside the output: no bounds to look at
                    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // (rows of outputs start wherever `start` puts them)
                    f4u lo = {y[0], y[1], y[2], y[3]}, hi = {y[4], y[5], y[6], y[7]};
                    *(FR_GLOBAL f4u*)(outp + c0) = lo;
                    *(FR_GLOBAL f4u*)(outp + c0 + 4) = hi;
                    if (has_red) {
                        // (finite samples give finite sums: no NaN to look for here -- rows with a NaN or an infinity are redone below.)
                        // Most steps change no lane's extremes (the filtered pulse has one rise): the largest and smallest of the lane's
                        // eight first, the indices only in a step where some lane has a new extreme
                        const float hi8 = fmaxf(fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])), fmaxf(fmaxf(y[4], y[5]), fmaxf(y[6], y[7])));
                        const float lo8 = fminf(fminf(fminf(y[0], y[1]), fminf(y[2], y[3])), fminf(fminf(y[4], y[5]), fminf(y[6], y[7])));
                        if (__any(hi8 > e.vmax)) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                const bool gt = y[r] > e.vmax;
                                e.vmax = gt ? y[r] : e.vmax;
                                e.imax = gt ? c0 + r : e.imax;
                            }
                        }
                        if (__any(lo8 < e.vmin)) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                const bool lt = y[r] < e.vmin;
                                e.vmin = lt ? y[r] : e.vmin;
                                e.imin = lt ? c0 + r : e.imin;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int c = c0 + r;
                        const bool valid = c >= 0 && c < p;
                        if (valid) outp[c] = y[r];
                        if (has_red) take(e, y[r], c, valid);
                    }
                }
                // ---- the last mp sums move to the front
                constexpr int NK = (DSP_FIR_RUNS_MAX_TAPS + DSP_FIR_RUNS_MAX_TAPS / 8) / 64;
                double keep[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k)
                    if (k * 64 < pmp) keep[k] = buf[PSTEP + k * 64 + lane];  // (the last lanes of the last round read past the window: the block's LDS runs that far, and the value is dropped)
                wave_sync();
#pragma unroll
                for (int k = 0; k < NK; ++k)
                    if (k * 64 + lane < pmp) buf[k * 64 + lane] = keep[k];
                wave_sync();
            };
            f4 b0[2], b1[2], b2[2];
            fetch(b0, 0);
            fetch(b1, 1);
            for (int g = 0; g < n_steps; g += 3) {
                fetch(b2, g + 2);
                step(g, b0);
                if (g + 1 < n_steps) {  // (uniform)
                    fetch(b0, g + 3);
                    step(g + 1, b1);
                }
                if (g + 2 < n_steps) {
                    fetch(b1, g + 4);
                    step(g + 2, b2);
                }
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
            reduce_finish<float>(A.red, row, e, (const float*)outp, p, lane, err);  // (its loads are back before the next row's stores are issued)
        }
    }
}

}  // namespace

extern "C" int dsp_internal_fir_runs_lds_bytes(int m) {
    const int mp = (m + 63) & ~63;
    return (4 * (mp + mp / 8 + STEP + STEP / 8) + 64) * (int)sizeof(double);  // (+ the round of the window's copy that reads past the last window)
}

extern "C" int dsp_internal_launch_fir_runs(const FirRunsArgs* A, FirRunsTable* table, int64_t n_wf, int blocks, int* err, hipStream_t stream) {
    if (n_wf <= 0 || A->p <= 0) return 0;
    hipLaunchKernelGGL(dsp_fir_runs_prep_kernel, dim3(1), dim3(64), 0, stream, A->taps, A->m, table);
    hipLaunchKernelGGL(dsp_fir_runs_kernel, dim3((unsigned)blocks), dim3(256), (size_t)dsp_internal_fir_runs_lds_bytes(A->m), stream, *A, n_wf, err);
    return (int)hipGetLastError();
}

extern "C" const char* dsp_internal_fir_runs_kernel_name() { return "dsp_fir_runs_kernel"; }
