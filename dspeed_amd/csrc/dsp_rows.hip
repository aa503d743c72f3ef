// dsp_rows.hip -- one waveform per LANE: the chain  [bl_subtract ->] pole_zero | double_pole_zero -> short trapezoid -> min_max /
// time_point_thresh, plus the Haar DWT of the pole-zero corrected waveform  (BASELINE.json configs[4]; the t0 branch of the LEGEND
// recipes has the same shape).
//
// Why a second execution model.  The waveform VM gives a waveform to a wavefront and shares its samples out among the lanes; every
// recursion then needs carries across lanes, and the trapezoids -- whose reference implementation rounds to float32 after every
// step -- need a rounding replay on top (dsp_vm.hip, trap_core).  When the trapezoid is SHORT (rise + flat + fall of the asymmetric
// t0 trapezoid: 137 samples) the history a sequential evaluation needs fits LDS for 64 waveforms at once, so here every lane walks
// its own waveform from the first sample to the last, exactly in the reference's order: no scan, no replay, every output bit-identical
// to the numba loop (double_pole_zero included, which the VM evaluates as a 2x2 affine scan).
//
// A pair of wavefronts per 64 waveforms.  The history ring (R samples x 64 lanes x 4 bytes = 40 KB for the 8/4/125 trapezoid) allows four
// such groups per CU.  One wavefront per group would leave one wavefront per SIMD, and a lone wavefront issues one instruction per
// 4 cycles whatever its kind (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'), so the work of a group is split over TWO
// wavefronts that share the ring as a pipe:
//   * the producer reads the rows (16 bytes per lane and load, four blocks ahead; a 128-byte line is used up by consecutive loads of the
//     same lane and comes from L2 meanwhile), subtracts the baseline, runs the pole-zero recursion in float64 and writes w1 = its float32
//     output into the ring, 8 samples per block; it also owns the Haar tree of w1 (registers; 16-byte stores of 4 coefficients);
//   * the consumer, one block behind, reads the block and the three lagged blocks the trapezoid needs from the ring, runs the
//     trapezoid recurrence and keeps the running extremes and threshold crossings -- time_point_thresh is evaluated on the fly:
//     walking backward from the arg-maximum = "the last crossing seen when the maximum was last raised".
// One s_barrier per block orders the two.  The consumer reads the LAGGED samples of block n+1 while it works on block n (every lag is
// at least one block), so the producer's writes of block n+1 and those reads never meet for R > largest lag + 7: R = 152 for the 137 of
// the LEGEND t0 trapezoid.  Ring layout: sample-major, [k][lane]: every access is one dword per lane at consecutive addresses
// (conflict-free, any lag), the samples of a block at immediate offsets j * 256 bytes; entries R .. R+7 mirror entries 0 .. 7, so a lagged
// block never wraps: (R + 8) * 256 = 40 960 bytes per pair, four pairs per CU.
//
// Reference bodies: processors/bl_subtract.py:11-46, pole_zero.py:24-77 and :82-198, trap_filters.py:12-76, :79-149, :152-227,
// min_max.py:11-82, time_point_thresh.py:12-92, dwt.py:13-81.  Compiled with -ffp-contract=off: one rounding per written operation.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "dsp_program.h"
#include "dsp_wave.h"

#define ROWS_LDS __attribute__((address_space(3)))
#define ROWS_GLOBAL __attribute__((address_space(1)))
#define ROWS_KARG __attribute__((address_space(4)))  /* the kernel-argument segment: scalar loads, no private copy of the struct */

namespace {

constexpr int RB = 8;  // samples per block (= per barrier)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// both wavefronts of the pair: my LDS accesses are done, then wait for the partner
__device__ __forceinline__ void pair_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 8-sample blocks a group walks when RowsArgs.stop_at_start holds (the kernel's STOP build: the plain one keeps its loop bounds as they were,
// a run-time test of the flag cost the other shapes 3.5 %): up to the block behind the latest start among its 64 rows -- both wavefronts of
// the pair read the same column and agree
template <typename ArgsRef>
__device__ __forceinline__ int rows_blocks(const ArgsRef& A, int64_t rowc) {
    const int all = A.len / 8;
    const float ts_f = A.ts ? ((const __attribute__((address_space(1))) float*)A.ts)[rowc * A.ts_stride] : A.ts_const;
    const bool ok = !(ts_f != ts_f) && ts_f >= 0.0f && ts_f < (float)A.len;
    const int last = wave_max(ok ? (int)ts_f : 0) / 8 + 2;
    return last < all ? last : all;
}

__device__ __forceinline__ void rows_report(int* err, int code, int64_t row) {
    if (atomicCAS(&err[0], 0, code) == 0) {
        err[1] = (int)(row & 0xffffffffll);
        err[2] = (int)(row >> 32);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// producer: rows -> [bl_subtract] -> pole_zero / double_pole_zero -> ring, Haar tree
// IN: 0 float32, 1 int16, 2 uint16 rows;  PZ: 1 pole_zero, 2 double_pole_zero
// ------------------------------------------------------------------------------------------------------------------------------
template <int IN, int PZ, bool STOP>
__device__ __forceinline__ void rows_produce(const ROWS_KARG RowsArgs& A, ROWS_LDS float* ring, int64_t n_wf, int* err) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * 64 + lane;
    const bool live = row < n_wf;
    const int64_t rowc = live ? row : n_wf - 1;
    constexpr int ESZ = IN == 0 ? 4 : 2, NV = IN == 0 ? 2 : 1;
    const ROWS_GLOBAL char* rowp = (const ROWS_GLOBAL char*)A.wf + (rowc * A.wf_stride + A.wf_offset) * ESZ;
    const float bl = A.sub_mode ? (A.bl ? ((const ROWS_GLOBAL float*)A.bl)[rowc * A.bl_stride] : A.bl_const) : 0.0f;
    const bool sub = A.sub_mode != 0;
    const int R = A.ring_entries;
    const int nblk = STOP ? rows_blocks(A, rowc) : A.len / RB;

    u4 pf[4][NV];
    auto fetch = [&](u4 (&dst)[NV], int blk) {
        const ROWS_GLOBAL u4* g = (const ROWS_GLOBAL u4*)(rowp + (size_t)blk * RB * ESZ);
#pragma unroll
        for (int v = 0; v < NV; ++v) dst[v] = g[v];
    };
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (u < nblk) fetch(pf[u], u);

    // recursion state (float64, as the reference keeps it)
    double acc = 0.0, xp = 0.0;                    // pole_zero: w_tmp[0], (double)w_in[i-1]
    double t0 = 0.0, t1 = 0.0, x1 = 0.0, x2 = 0.0;  // double_pole_zero: w_tmp[0], w_tmp[1], (double)w_in[i-1], (double)w_in[i-2]
    const double c = A.pz_c, n1 = A.n1, n2 = A.n2, d1 = A.d1, d2 = A.d2;
    bool in_nan = false;

    // Haar tree: levels 1-3 inside a block, levels 4.. across blocks (one pending value per level), four finished coefficients per store
    const int L = A.dwt_level;
    const float hc = 0.70710678118654752440f;  // (float)(1/sqrt 2): PyWavelets keeps the filter in the data type
    const float hs = A.dwt_part == 'd' ? -hc : hc;  // sign of the high tap at the LAST level ('d': detail coefficients)
    float pend4 = 0.0f, pend5 = 0.0f, pend6 = 0.0f, pend7 = 0.0f, pend8 = 0.0f;
    f4 obuf = {0.0f, 0.0f, 0.0f, 0.0f};
    int ocnt = 0;
    ROWS_GLOBAL float* dwt_row = (ROWS_GLOBAL float*)A.dwt_out + row * A.dwt_stride;

    int pos = 0;
    for (int sb = 0; sb < nblk; sb += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int blk = sb + u;
            if (blk >= nblk) break;  // (uniform)
            // ---- samples of this block as float32 (the reference's ufunc casting picks the float32 loop for 16-bit rows)
            float x[RB];
            if (IN == 0) {
#pragma unroll
                for (int j = 0; j < RB; ++j) x[j] = __uint_as_float(pf[u][j >> 2][j & 3]);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const unsigned int wv = pf[u][0][m];
                    x[2 * m] = IN == 1 ? (float)(short)(wv & 0xffffu) : (float)(wv & 0xffffu);
                    x[2 * m + 1] = IN == 1 ? (float)(short)(wv >> 16) : (float)(wv >> 16);
                }
            }
            if (blk + 4 < nblk) fetch(pf[u], blk + 4);
            if (sub) {
#pragma unroll
                for (int j = 0; j < RB; ++j) x[j] = x[j] - bl;  // bl_subtract.py:45
            }
            if (IN == 0) {
#pragma unroll
                for (int j = 0; j < RB; ++j) in_nan |= (x[j] != x[j]);
            }
            // ---- stage 1
            float w[RB];
            auto steps = [&](auto first_tag) {
                constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const double xd = (double)x[j];
                    if (PZ == 0) {  // rows that are the corrected waveform already
                        w[j] = x[j];
                    } else if (PZ == 1) {
                        if (FIRST && j == 0) {  // pole_zero.py:66-67
                            w[j] = x[j];
                            acc = xd;
                        } else {                // :69-73, evaluated left to right
                            acc = (acc + xd) - xp * c;
                            w[j] = (float)acc;
                        }
                        xp = xd;
                    } else {
                        if (FIRST && j < 2) {   // pole_zero.py:180-184
                            w[j] = x[j];
                            t0 = t1;
                            t1 = xd;
                        } else {                // :187-193
                            const double t2 = ((((xd + n1 * x1) + n2 * x2) - d1 * t1) - d2 * t0);
                            w[j] = (float)t2;
                            t0 = t1;
                            t1 = t2;
                        }
                        x2 = x1;
                        x1 = xd;
                    }
                }
            };
            if (blk == 0)
                steps(std::true_type{});
            else
                steps(std::false_type{});
            // ---- into the ring
            ROWS_LDS float* wr = ring + pos * 64 + lane;
#pragma unroll
            for (int j = 0; j < RB; ++j) wr[j * 64] = w[j];
            if (pos == 0) {  // entries R .. R+7 mirror entries 0 .. 7: a lagged block that starts near the end of the ring reads on into them
#pragma unroll
                for (int j = 0; j < RB; ++j) wr[(R + j) * 64] = w[j];
            }
            pos += RB;
            if (pos == R) pos = 0;
            // ---- Haar tree (dwt.py:81 -> pywt.downcoef: each level fl(fl(hi_tap * x[2k+1]) + fl(lo_tap * x[2k])))
            if (L > 0) {
                float a1[4], a2[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) a1[k] = hc * w[2 * k + 1] + hc * w[2 * k];
#pragma unroll
                for (int k = 0; k < 2; ++k) a2[k] = hc * a1[2 * k + 1] + hc * a1[2 * k];
                float v = (L == 3 ? hs : hc) * a2[1] + hc * a2[0];
                // levels 4..8 (all conditions uniform): the value of the even block of a level waits for the odd one
                bool pending = false, done = (L == 3);
#define HAAR_LEVEL(LV, PEND)                                   \
    if (!pending && !done) {                                    \
        if (((blk >> (LV - 4)) & 1) == 0) {                     \
            PEND = v;                                           \
            pending = true;                                     \
        } else {                                                \
            v = (L == LV ? hs : hc) * v + hc * PEND;            \
            done = (L == LV);                                   \
        }                                                       \
    }
                HAAR_LEVEL(4, pend4)
                HAAR_LEVEL(5, pend5)
                HAAR_LEVEL(6, pend6)
                HAAR_LEVEL(7, pend7)
                HAAR_LEVEL(8, pend8)
#undef HAAR_LEVEL
                if (!pending) {  // a finished coefficient
                    if (ocnt == 0) obuf[0] = v;
                    else if (ocnt == 1) obuf[1] = v;
                    else if (ocnt == 2) obuf[2] = v;
                    else obuf[3] = v;
                    if (++ocnt == 4) {
                        const int k0 = ((blk * RB) >> L) - 3;  // index of obuf[0]
                        if (live) *(ROWS_GLOBAL f4*)(dwt_row + k0) = obuf;
                        ocnt = 0;
                    }
                }
            }
            pair_barrier();
        }
    }
    // ---- what only shows at the end
    const bool state_nan = PZ == 0 ? false : (PZ == 1 ? (acc != acc) : (t1 != t1 || t0 != t0));
    if (PZ == 1 && live && state_nan && !in_nan && !A.pz_param_nan && !(bl != bl)) rows_report(err, DSP_E_PZ_NAN, row);  // pole_zero.py:76-77
    if (L > 0 && live && (state_nan || in_nan || A.pz_param_nan)) {  // dwt.py:70-71: a NaN anywhere in w1 -> NaN coefficients
        const float nanv = quiet_nan<float>();
        for (int k = 0; k < (A.len >> L); ++k) dwt_row[k] = nanv;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// consumer: ring -> trapezoid -> running extremes, threshold crossings
// TPT: 0 none, 1 walk backward from a known start, 2 backward from the running arg-extremum, 3 forward from a known start,
//      4 forward from the running arg-extremum
// ------------------------------------------------------------------------------------------------------------------------------
template <int TRAP, bool RPOW2, int TPT, bool STOP>
__device__ __forceinline__ void rows_consume(const ROWS_KARG RowsArgs& A, ROWS_LDS float* ring, int64_t n_wf, int* err) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * 64 + lane;
    const bool live = row < n_wf;
    const int64_t rowc = live ? row : n_wf - 1;
    const int R = A.ring_entries, n = A.len;
    const int nblk = STOP ? rows_blocks(A, rowc) : A.len / RB;
    const float thr = A.thr ? ((const ROWS_GLOBAL float*)A.thr)[rowc * A.thr_stride] : A.thr_const;
    const float ts_f = A.ts ? ((const ROWS_GLOBAL float*)A.ts)[rowc * A.ts_stride] : A.ts_const;
    const bool ts_ok = (TPT == 1 || TPT == 3) && !(ts_f != ts_f) && floorf(ts_f) == ts_f && ts_f >= 0.0f && ts_f < (float)n;
    const int tsi = ts_ok ? (int)ts_f : 0;
    const bool use_min = A.tpt_use_min != 0;
    const double rr = A.rr, ll = A.ll, inv_rr = A.inv_rr, inv_ll = A.inv_ll;
    const int lag0 = A.lag[0], lag1 = A.lag[1], lag2 = A.lag[2];
    constexpr bool fwd = (TPT == 3 || TPT == 4);

    float y = -0.0f;  // (-0 + x == x for every x, -0 included: the first step reproduces w_out[0] = w_in[0] [/ rise])
    float vmin = __builtin_inff(), vmax = -__builtin_inff();
    int imin = 0, imax = 0;
    int last = -1, tp0 = -1;
    // sample before the block, for the crossing tests; the start value rules a crossing at sample 0 out: walking backward the test
    // is "not (thr <= w[i-1]) and ...", forward "(thr >= w[i-1]) and ..." -- thr itself / a NaN make both halves false
    float yprev = fwd ? quiet_nan<float>() : thr;

    // lagged samples of a block: every lag is at least one block, so they are in the ring a block early -- read them then, which also
    // takes 8 samples off the history the ring has to keep (R > largest lag + 7)
    int lpos = 0;  // ring position of the block whose lagged samples are read next
    auto load_lags = [&](float (&l)[3][RB]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            int p = lpos - (q == 0 ? lag0 : (q == 1 ? lag1 : lag2));
            if (p < 0) p += R;
            const ROWS_LDS float* b = ring + p * 64 + lane;  // (a block that starts in the last 7 entries runs on into the mirror of entries 0..7)
#pragma unroll
            for (int j = 0; j < RB; ++j) l[q][j] = b[j * 64];
        }
        lpos += RB;
        if (lpos == R) lpos = 0;
    };
    int pos = 0;
    auto block = [&](const float (&l)[3][RB], float (&lnext)[3][RB], int blk) {
        float cur[RB], ys[RB];
        {
            const ROWS_LDS float* b = ring + pos * 64 + lane;
#pragma unroll
            for (int j = 0; j < RB; ++j) cur[j] = b[j * 64];
        }
        pos += RB;
        if (pos == R) pos = 0;
        if (blk + 1 < nblk) load_lags(lnext);
        // ---- trapezoid steps in the reference's operation order; samples before the waveform read 0 from the ring, which turns the
        // start-up loops of the reference into the general step (x - 0 == x, y + 0/r == y)
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            if (TRAP == TRAP_FILTER) {
                y = (((y + cur[j]) - l[0][j]) - l[1][j]) + l[2][j];
            } else if (TRAP == TRAP_NORM) {
                const float e = ((cur[j] - l[0][j]) - l[1][j]) + l[2][j];
                y = (float)((double)y + (RPOW2 ? (double)e * inv_rr : div_by_count_fx((double)e, rr, inv_rr)));
            } else {
                const float e1 = cur[j] - l[0][j], e2 = l[1][j] - l[2][j];
                const double q1 = RPOW2 ? (double)e1 * inv_rr : div_by_count_fx((double)e1, rr, inv_rr);
                y = (float)(((double)y + q1) - div_by_count_fx((double)e2, ll, inv_ll));
            }
            ys[j] = y;
        }
        // ---- does anything happen in this block?  A new extreme needs the block's extreme beyond the running one; a crossing needs
        // the threshold between the block's extremes (the sample before the block included).  Mostly neither, for all 64 waveforms.
        const float bmax = fmaxf(fmaxf(fmaxf(ys[0], ys[1]), fmaxf(ys[2], ys[3])), fmaxf(fmaxf(ys[4], ys[5]), fmaxf(ys[6], ys[7])));
        const float bmin = fminf(fminf(fminf(ys[0], ys[1]), fminf(ys[2], ys[3])), fminf(fminf(ys[4], ys[5]), fminf(ys[6], ys[7])));
        bool look = (bmax > vmax) || (bmin < vmin);
        if (TPT != 0) look |= (thr <= fmaxf(bmax, yprev)) && (thr >= fminf(bmin, yprev));
        if (wave_any(look)) {
            const int i0 = blk * RB;
            bool c1p = thr <= yprev, c2p = thr >= yprev;
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int i = i0 + j;
                const float v = ys[j];
                // min_max.py:73-77: strict comparisons, the first occurrence stays
                const bool lt = v < vmin, gt = v > vmax;
                vmin = lt ? v : vmin;
                imin = lt ? i : imin;
                vmax = gt ? v : vmax;
                imax = gt ? i : imax;
                if (TPT != 0) {
                    const bool c1 = thr <= v, c2 = thr >= v;
                    const bool ev = use_min ? lt : gt;
                    if (!fwd) {
                        // time_point_thresh.py:85-92: i in [1, t_start] with (w[i-1] < thr <= w[i]) or (w[i-1] > thr >= w[i]); a NaN among
                        // the operands makes the whole output NaN anyway, so "w[i-1] < thr" may be taken as "not (thr <= w[i-1])"
                        const bool cross = (!c1p && c1) || (!c2p && c2);
                        if (TPT == 1) {
                            last = (cross && i <= tsi) ? i : last;
                        } else {
                            last = cross ? i : last;
                            tp0 = ev ? last : tp0;  // the walk from the (new) extremum looks at sample i itself first
                        }
                    } else {
                        // :77-84: k = i - 1 in [t_start, n - 2] with (w[k] <= thr < w[k+1]) or (w[k] >= thr > w[k+1])
                        const bool cross = (c2p && !c2) || (c1p && !c1);
                        if (TPT == 3) {
                            tp0 = (cross && tp0 < 0 && i - 1 >= tsi) ? i - 1 : tp0;
                        } else {
                            tp0 = (cross && tp0 < 0) ? i - 1 : tp0;
                            tp0 = ev ? -1 : tp0;  // a new extremum at i: the walk starts over from there
                        }
                    }
                    c1p = c1;
                    c2p = c2;
                }
            }
        }
        yprev = ys[RB - 1];
    };

    float la[3][RB], lb[3][RB];
    load_lags(la);   // block 0: zeros (the ring was cleared)
    pair_barrier();  // block 0 is in the ring
    for (int blk = 0; blk < nblk; blk += 2) {
        block(la, lb, blk);
        if (blk + 1 < nblk) {
            pair_barrier();
            block(lb, la, blk + 1);
            if (blk + 2 < nblk) pair_barrier();
        }
    }
    if (!live) return;
    // ---- results.  A NaN sample anywhere upstream is still in y: every recurrence here feeds its own output back
    const float nanv = quiet_nan<float>();
    const bool nan_all = (y != y) || A.pz_param_nan || A.trap_all_nan;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!A.out_mm[k]) continue;
        float v = k == 0 ? (float)imin : (k == 1 ? (float)imax : (k == 2 ? vmin : vmax));
        ((ROWS_GLOBAL float*)A.out_mm[k])[row * A.out_mm_stride[k]] = nan_all ? nanv : v;
    }
    if (TPT != 0 && A.out_tpt) {
        float out = nanv;
        const bool known = (TPT == 1 || TPT == 3);
        if (!nan_all && !(thr != thr) && !A.walk_nan && !(known && ts_f != ts_f)) {
            if (known && floorf(ts_f) != ts_f) {
                rows_report(err, DSP_E_TPT_START_INT, row);
            } else if (A.walk_frac) {
                rows_report(err, DSP_E_TPT_WALK_INT, row);
            } else if (known && !ts_ok) {
                rows_report(err, DSP_E_TPT_RANGE, row);
            } else {
                const int found = (TPT == 1) ? last : tp0;
                if (found >= 0) out = (float)found;
            }
        }
        ((ROWS_GLOBAL float*)A.out_tpt)[row * A.out_tpt_stride] = out;
    }
}

template <bool STOP>
__global__ void __launch_bounds__(128, 2) dsp_rows_kernel(RowsArgs A_, int64_t n_wf, int* err) {
    // (taking the address of the by-value argument would make the compiler copy it to scratch: read it where it lies)
    const ROWS_KARG RowsArgs& A = *(const ROWS_KARG RowsArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    extern __shared__ __attribute__((aligned(16))) float rows_smem[];
    ROWS_LDS float* ring = (ROWS_LDS float*)rows_smem;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int e = (int)threadIdx.x; e < (A.ring_entries + RB) * 64; e += 128) ring[e] = 0.0f;  // samples before the waveform are zeros
    __syncthreads();
    if (wave == 0) {
        switch (A.in_kind * 3 + A.pz_kind) {
            case 0: rows_produce<0, 0, STOP>(A, ring, n_wf, err); break;
            case 1: rows_produce<0, 1, STOP>(A, ring, n_wf, err); break;
            case 2: rows_produce<0, 2, STOP>(A, ring, n_wf, err); break;
            case 3: rows_produce<1, 0, STOP>(A, ring, n_wf, err); break;
            case 4: rows_produce<1, 1, STOP>(A, ring, n_wf, err); break;
            case 5: rows_produce<1, 2, STOP>(A, ring, n_wf, err); break;
            case 6: rows_produce<2, 0, STOP>(A, ring, n_wf, err); break;
            case 7: rows_produce<2, 1, STOP>(A, ring, n_wf, err); break;
            default: rows_produce<2, 2, STOP>(A, ring, n_wf, err); break;
        }
    } else {
#define ROWS_C(TRAP, P2)                                                      \
    switch (A.tpt_mode) {                                                     \
        case 0: rows_consume<TRAP, P2, 0, STOP && 0 == 1>(A, ring, n_wf, err); break;         \
        case 1: rows_consume<TRAP, P2, 1, STOP && 1 == 1>(A, ring, n_wf, err); break;         \
        case 2: rows_consume<TRAP, P2, 2, STOP && 2 == 1>(A, ring, n_wf, err); break;         \
        case 3: rows_consume<TRAP, P2, 3, STOP && 3 == 1>(A, ring, n_wf, err); break;         \
        default: rows_consume<TRAP, P2, 4, STOP && 4 == 1>(A, ring, n_wf, err); break;        \
    }
        if (A.trap_kind == TRAP_FILTER) {
            ROWS_C(TRAP_FILTER, false)
        } else if (A.trap_kind == TRAP_NORM) {
            if (A.rise_pow2) { ROWS_C(TRAP_NORM, true) } else { ROWS_C(TRAP_NORM, false) }
        } else {
            if (A.rise_pow2) { ROWS_C(TRAP_ASYM, true) } else { ROWS_C(TRAP_ASYM, false) }
        }
#undef ROWS_C
    }
}

}  // namespace

extern "C" int dsp_internal_launch_rows(const RowsArgs* A, int64_t n_wf, int* err, int lds_bytes, hipStream_t stream) {
    if (n_wf <= 0) return 0;
    const unsigned blocks = (unsigned)((n_wf + 63) / 64);
    if (A->stop_at_start)
        hipLaunchKernelGGL(dsp_rows_kernel<true>, dim3(blocks), dim3(128), lds_bytes, stream, *A, n_wf, err);
    else
        hipLaunchKernelGGL(dsp_rows_kernel<false>, dim3(blocks), dim3(128), lds_bytes, stream, *A, n_wf, err);
    return (int)hipGetLastError();
}

extern "C" int dsp_internal_set_rows_lds(int lds_bytes) {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dsp_rows_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (rc) return rc;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dsp_rows_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}

extern "C" const char* dsp_internal_rows_kernel_name() { return "dsp_rows_kernel"; }
