// dsp_current.hip -- one waveform per LANE: the current branch of the LEGEND Ge recipes
//   windower(t0) -> avg_current -> upsampler(x cnt) -> moving_window_multi(L, 3 passes, alternating) -> min_max      (A/E: A_max, tp_aoe_max)
// as ONE kernel on rows in HBM (what whole recipes run ahead of their program: the pole-zero corrected waveform is rows there already and
// the window start, tp_0_est, a column).
//
// Why.  Inside the program (one wavefront per waveform) the three moving averages are float32 recurrences that round after every
// operation: they need the rounding replay of the trapezoids (two runs per pass) and were 30 % of the LEGEND program.  One waveform per
// lane runs the reference's loops as they are written, 64 waveforms per instruction, every output bit-identical to the numba loop.
//
// The catch is the middle pass.  moving_window_multi with mw_type 0 goes left -> right, right -> left, left -> right, so a pass needs the
// WHOLE output of the pass before it, in the opposite order: 4784 floats per waveform, 1.2 MB per 64 lanes -- not LDS, and streaming it
// through HBM would cost 77 kB of traffic per waveform.  Instead only CHECKPOINTS are kept (the running value of a pass every 16 samples:
// 2 x 299 floats per waveform, in a scratch area that stays in L2) and the samples between two checkpoints are recomputed, in registers,
// when the next pass needs them -- a recurrence restarted from its own intermediate value reproduces its values bit for bit:
//   sweep A (blocks 0 .. nb-1):  pass 1 values, checkpoint P0[b] = value before block b
//   sweep B (blocks nb-1 .. 0):  pass 1 block b again from P0[b]; pass 2 steps on it (the lagged operand: block b + L/16, again from its
//                                checkpoint); checkpoint P1[b] = pass 2 value before block b in walking order
//   sweep C (blocks 0 .. nb-1):  pass 1 block b + L/16 from P0, pass 2 block b from P1, pass 3 steps on it + the running min / max
// 48 divisions per 16 samples instead of 32 (pass 1 adds one of n_c precomputed increments per sample: the upsampled waveform repeats each
// current sample cnt times), no sample array anywhere; the checkpoints and increments of the next block are loaded while a block is computed.
//
// Shape taken (anything else runs on the waveform VM): float32 rows; an upsampling factor in {1, 2, 4, 8, 16}; window length L a multiple
// of 16 with L / 16 <= 7; three alternating windows; an upsampled length that is a multiple of 16.
// Reference bodies: processors/windower.py:12-54, moving_windows.py:206-249 (avg_current) and :117-204 (moving_window_multi),
// upsampler.py:13-56, min_max.py:11-82.  Compiled with -ffp-contract=off: one rounding per written operation.
#include <hip/hip_runtime.h>

#include "dsp_program.h"
#include "dsp_wave.h"

#define CUR_LDS __attribute__((address_space(3)))
#define CUR_GLOBAL __attribute__((address_space(1)))
#define CUR_KARG __attribute__((address_space(4)))

namespace {

constexpr int CB = 16;  // samples per block (= per checkpoint)

typedef float f4 __attribute__((ext_vector_type(4)));

// e / d in float32, correctly rounded: the float64 quotient (div_by_count: correctly rounded) rounds to the float32 one without a
// double-rounding error (53 >= 2 * 24 + 2); infinities and NaN through the hardware's division fix-up
__device__ __forceinline__ float div_f32(float e, double d, double inv_d) {
    const double x = (double)e;
    const double q = x * inv_d;
    const double r = __builtin_fma(-q, d, x);
    return (float)__builtin_amdgcn_div_fixup(__builtin_fma(r, inv_d, q), d, x);
}

// SH: log2 of the upsampling factor (0 .. 4).  SCAN: screen the whole rows for NaN (off when the producer of the rows guarantees "a NaN
// anywhere means NaN everywhere" -- pole_zero does: the window's own samples are looked at anyway).
template <int SH, bool SCAN>
__global__ void __launch_bounds__(64, 1) dsp_current_kernel(CurrentArgs A_, int64_t n_wf) {
    const CUR_KARG CurrentArgs& A = *(const CUR_KARG CurrentArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    extern __shared__ __attribute__((aligned(16))) float cur_smem[];
    constexpr int CNT = 1 << SH, HALF = CNT / 2;
    constexpr int NR = (CB >> SH) + (SH > 0 ? 1 : 0);  // runs of equal upsampled samples that a block of 16 touches
    const int lane = lane_id();
    const int q = A.ma_len / CB, nring = q + 1;
    // pass-2 blocks: [nring][CB][64].  (The pass-1 blocks a pass-2 step lags behind had a ring of their own until round 4: 32 kB per wavefront,
    // four or five wavefronts per CU -- one per SIMD, and a lone wavefront waits out every dependent instruction.  A pass-1 block is 16
    // additions from its checkpoint: it is recomputed where it is needed, and eight wavefronts fit a CU.)
    CUR_LDS float* ring1 = (CUR_LDS float*)cur_smem;
    const int n_c = A.n_c, nb = A.n_up / CB, ql = A.ma_len >> SH;  // u[i] = c[(i + HALF) >> SH];  u[i - L] = c[((i + HALF) >> SH) - ql]
    const double len_d = (double)A.ma_length, inv_len = 1.0 / len_d;
    const double acl_d = (double)A.ac_length, inv_acl = 1.0 / acl_d;
    CUR_GLOBAL float* scr = (CUR_GLOBAL float*)A.scratch + (int64_t)blockIdx.x * A.scratch_per_wave + lane;
    CUR_GLOBAL float* D = scr;                        // pass 1's increment while the upsampled samples equal c[t]:  D[t] at D[t * 64]
    CUR_GLOBAL float* P0 = scr + (int64_t)n_c * 64;   // P0[b]  at P0[b * 64]
    CUR_GLOBAL float* P1 = P0 + (int64_t)nb * 64;
    const int64_t n_groups = (n_wf + 63) / 64;

    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t row = g * 64 + lane;
        const bool live = row < n_wf;
        const int64_t rowc = live ? row : n_wf - 1;
        // ---- NaN anywhere in the row -> NaN window (windower.py:36): the wavefront screens its 64 rows together, 1 KiB per load, the
        // loads of a row in flight while the row before it is looked at
        unsigned long long nan_rows = 0ull;
        if (SCAN) {
            const int nv = A.n_in >> 2;  // float4 per row (n_in is a multiple of 4)
            constexpr int U = 8;
            auto fetch = [&](f4 (&x)[U], int r, int v0) {
                int64_t rr = g * 64 + r;
                rr = rr < n_wf ? rr : n_wf - 1;
                const CUR_GLOBAL f4* p = (const CUR_GLOBAL f4*)((const CUR_GLOBAL float*)A.wf + rr * A.wf_stride + A.wf_offset);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int v = v0 + u * 64 + lane;
                    x[u] = p[v < nv ? v : nv - 1];
                }
            };
            auto any_nan = [&](const f4 (&x)[U]) {
                bool bad = false;
#pragma unroll
                for (int u = 0; u < U; ++u) bad |= (x[u][0] != x[u][0]) | (x[u][1] != x[u][1]) | (x[u][2] != x[u][2]) | (x[u][3] != x[u][3]);
                return bad;
            };
            const int per_row = (nv + U * 64 - 1) / (U * 64), total = 64 * per_row;
            f4 xa[U], xb[U];
            fetch(xa, 0, 0);
            for (int k = 0; k < total; k += 2) {  // batch k = (row k / per_row, part k % per_row)
                if (k + 1 < total) fetch(xb, (k + 1) / per_row, ((k + 1) % per_row) * U * 64);
                if (wave_any(any_nan(xa))) nan_rows |= 1ull << (k / per_row);
                if (k + 2 < total) fetch(xa, (k + 2) / per_row, ((k + 2) % per_row) * U * 64);
                if (k + 1 < total && wave_any(any_nan(xb))) nan_rows |= 1ull << ((k + 1) / per_row);
            }
        }
        // ---- windower + avg_current: c[k] = (w[k + La] - w[k]) / length, w[j] = x[beg + j]; and what pass 1 adds per sample while the
        // upsampled waveform repeats c[t]:  D[t] = (c[t] - c[max(t - ql, 0)]) / L  (the first L samples subtract w_buf[0] = c[0]; for the
        // run that straddles sample L both rules name c[0])
        const float t0 = A.t0 ? ((const CUR_GLOBAL float*)A.t0)[rowc * A.t0_stride] : A.t0_const;
        const int m = A.win_len, n_in = A.n_in, La = A.ac_lag;
        // int(t0) truncates toward zero; a start outside [0, n_in - m] leaves NaN samples in the window (windower.py:41-54), and one NaN in
        // the window makes every later waveform NaN (avg_current, upsampler, moving_window_multi, min_max each return NaN for a NaN input)
        bool valid = !((nan_rows >> lane) & 1ull) && !(t0 != t0) && t0 > -1.0f && t0 < (float)(n_in - m + 1);
        const int beg = valid ? (int)t0 : 0;
        valid = valid && beg >= 0 && beg + m <= n_in;
        const CUR_GLOBAL float* xw = (const CUR_GLOBAL float*)A.wf + rowc * A.wf_stride + A.wf_offset + (valid ? beg : 0);
        bool c_nan = false;
        const float c0 = div_f32(xw[La] - xw[0], acl_d, inv_acl);
        for (int t = 0; t < n_c; ++t) {
            const int tl = t >= ql ? t - ql : 0;
            const float ch = div_f32(xw[t + La] - xw[t], acl_d, inv_acl);
            const float cl = div_f32(xw[tl + La] - xw[tl], acl_d, inv_acl);
            c_nan |= (ch != ch);
            D[(int64_t)t * 64] = div_f32(ch - cl, len_d, inv_len);
        }

        // the increments of pass 1 in block b: run r of the block repeats c[(16 b + HALF >> SH) + r]
        auto load_d = [&](float (&d)[NR], int b) {
            const int bc = b < 0 ? 0 : (b < nb ? b : nb - 1);
            const CUR_GLOBAL float* p = D + (int64_t)((bc * CB + HALF) >> SH) * 64;
#pragma unroll
            for (int r = 0; r < NR; ++r) d[r] = p[r * 64];
        };
        auto load_cp = [&](const CUR_GLOBAL float* P, int b) {
            const int bc = b < 0 ? 0 : (b < nb ? b : nb - 1);
            return P[(int64_t)bc * 64];
        };
        // pass 1 (left -> right) on block b from the value before it: o[j] = out0[16 b + j]
        auto pass1_block = [&](int b, float acc, const float (&d)[NR], float (&o)[CB]) {
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                const float next = acc + d[((j + HALF) >> SH) - (HALF >> SH)];
                acc = (j == 0 && b == 0) ? c0 : next;  // w_out[0] = w_buf[0]
                o[j] = acc;
            }
            return acc;
        };
        // pass 2 (right -> left) on block b: out1[k] = out1[k + 1] + (out0[k] - out0[k + L]) / length, the last L samples subtract out1[n - 1]
        auto pass2_block = [&](int b, float acc, const float (&o0)[CB], const float (&lag)[CB], float first1, float (&o1)[CB]) {
            const bool tail = b >= nb - q;
#pragma unroll
            for (int j = CB - 1; j >= 0; --j) {
                const float e = o0[j] - (tail ? first1 : lag[j]);
                const float next = acc + div_f32(e, len_d, inv_len);
                acc = (j == CB - 1 && b == nb - 1) ? o0[j] : next;  // w_out[-1] = w_buf[-1]
                o1[j] = acc;
            }
            return acc;
        };

        // ---- sweep A: pass 1, checkpoints
        float acc0 = 0.0f;
        {
            float d[NR], dn[NR];
            load_d(d, 0);
            for (int b = 0; b < nb; ++b) {
                float o[CB];
                load_d(dn, b + 1);
                P0[(int64_t)b * 64] = acc0;
                acc0 = pass1_block(b, acc0, d, o);
#pragma unroll
                for (int r = 0; r < NR; ++r) d[r] = dn[r];
            }
        }
        const float first1 = acc0;  // out0[n - 1] = out1[n - 1]
        // ---- sweep B: pass 2 over recomputed pass-1 blocks, checkpoints
        float acc1 = 0.0f;
        {
            float d[NR], dn[NR], dq[NR], dqn[NR];
            load_d(d, nb - 1);
            load_d(dq, nb - 1 + q);
            float cp = load_cp(P0, nb - 1), cpq = load_cp(P0, nb - 1 + q);
            for (int b = nb - 1; b >= 0; --b) {
                float o0[CB], lag[CB], o1[CB];
                load_d(dn, b - 1);
                load_d(dqn, b - 1 + q);
                const float cpn = load_cp(P0, b - 1), cpqn = load_cp(P0, b - 1 + q);
                pass1_block(b, cp, d, o0);
                pass1_block(b + q, cpq, dq, lag);  // (the lagged block, again from its checkpoint; past the end in the tail, where nobody reads it)
                P1[(int64_t)b * 64] = acc1;
                acc1 = pass2_block(b, acc1, o0, lag, first1, o1);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    d[r] = dn[r];
                    dq[r] = dqn[r];
                }
                cp = cpn;
                cpq = cpqn;
            }
        }
        const float first2 = acc1;  // out1[0] = out2[0]
        // ---- sweep C: pass 3 over recomputed pass-2 blocks, running extremes
        float acc2 = 0.0f, vmin = __builtin_inff(), vmax = -__builtin_inff();
        int imin = 0, imax = 0;
        {
            float d[NR], dn[NR], db[NR], dbn[NR];
            load_d(d, q);
            load_d(db, 0);
            float cp0 = load_cp(P0, q), cpb = load_cp(P0, 0), cp1 = load_cp(P1, 0);
            for (int b = 0; b < nb; ++b) {
                float o0[CB], lag0[CB], o1[CB], lag1[CB];
                load_d(dn, b + q + 1);
                load_d(dbn, b + 1);
                const float cp0n = load_cp(P0, b + q + 1), cpbn = load_cp(P0, b + 1), cp1n = load_cp(P1, b + 1);
                pass1_block(b + q, cp0, d, lag0);  // (past the end: values nobody uses -- the tail subtracts first1)
                pass1_block(b, cpb, db, o0);       // (both pass-1 blocks from their checkpoints: 16 additions each, no ring)
                pass2_block(b, cp1, o0, lag0, first1, o1);
                CUR_LDS float* w1 = ring1 + (b % nring) * CB * 64 + lane;
#pragma unroll
                for (int j = 0; j < CB; ++j) w1[j * 64] = o1[j];
                {
                    const CUR_LDS float* r1 = ring1 + ((b + nring - q) % nring) * CB * 64 + lane;  // block b - q (unused for b < q)
#pragma unroll
                    for (int j = 0; j < CB; ++j) lag1[j] = r1[j * 64];
                }
                const bool head = b < q;
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    const int i = b * CB + j;
                    const float e = o1[j] - (head ? first2 : lag1[j]);
                    const float next = acc2 + div_f32(e, len_d, inv_len);
                    acc2 = (j == 0 && b == 0) ? o1[0] : next;
                    const bool lt = acc2 < vmin, gt = acc2 > vmax;  // min_max.py:73-77: strict, the first occurrence stays
                    vmin = lt ? acc2 : vmin;
                    imin = lt ? i : imin;
                    vmax = gt ? acc2 : vmax;
                    imax = gt ? i : imax;
                }
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    d[r] = dn[r];
                    db[r] = dbn[r];
                }
                cp0 = cp0n;
                cpb = cpbn;
                cp1 = cp1n;
            }
        }
        // ---- results: a NaN anywhere is still in the last value (every pass feeds its output back; pass 2 carries a NaN of pass 1 down to
        // sample 0, where pass 3 starts)
        if (live) {
            const bool nan_all = !valid || c_nan || (c0 != c0) || (acc2 != acc2) || (first1 != first1);
            const float nanv = quiet_nan<float>();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!A.out[k]) continue;
                const float v = k == 0 ? (float)imin : (k == 1 ? (float)imax : (k == 2 ? vmin : vmax));
                ((CUR_GLOBAL float*)A.out[k])[row * A.out_stride[k]] = nan_all ? nanv : v;
            }
        }
    }
}

template <int SH>
int launch_sh(const CurrentArgs* A, int64_t n_wf, int blocks, int lds_bytes, hipStream_t stream) {
    if (A->scan_rows)
        hipLaunchKernelGGL((dsp_current_kernel<SH, true>), dim3((unsigned)blocks), dim3(64), lds_bytes, stream, *A, n_wf);
    else
        hipLaunchKernelGGL((dsp_current_kernel<SH, false>), dim3((unsigned)blocks), dim3(64), lds_bytes, stream, *A, n_wf);
    return (int)hipGetLastError();
}

template <int SH>
int set_lds_sh(int lds_bytes) {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dsp_current_kernel<SH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (rc) return rc;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dsp_current_kernel<SH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}

}  // namespace

extern "C" int dsp_internal_current_lds_bytes(int ma_len) { return (ma_len / CB + 1) * CB * 64 * 4; }

extern "C" int dsp_internal_launch_current(const CurrentArgs* A, int64_t n_wf, int blocks, int lds_bytes, hipStream_t stream) {
    if (n_wf <= 0) return 0;
    switch (A->up_shift) {
        case 0: return launch_sh<0>(A, n_wf, blocks, lds_bytes, stream);
        case 1: return launch_sh<1>(A, n_wf, blocks, lds_bytes, stream);
        case 2: return launch_sh<2>(A, n_wf, blocks, lds_bytes, stream);
        case 3: return launch_sh<3>(A, n_wf, blocks, lds_bytes, stream);
        default: return launch_sh<4>(A, n_wf, blocks, lds_bytes, stream);
    }
}

extern "C" int dsp_internal_set_current_lds(int lds_bytes) {
    int rc = set_lds_sh<0>(lds_bytes);
    if (!rc) rc = set_lds_sh<1>(lds_bytes);
    if (!rc) rc = set_lds_sh<2>(lds_bytes);
    if (!rc) rc = set_lds_sh<3>(lds_bytes);
    if (!rc) rc = set_lds_sh<4>(lds_bytes);
    return rc;
}

extern "C" const char* dsp_internal_current_kernel_name() { return "dsp_current_kernel"; }
