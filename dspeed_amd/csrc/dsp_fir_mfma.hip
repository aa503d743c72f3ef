// dsp_fir_mfma.hip -- the long FIR of the energy filters (cusp / zac: 5792 taps, 301 'valid' outputs of a 6092-sample slice, followed by
// numpy.amax; BASELINE.json configs[2]) as a float32 matrix product on the matrix cores.
//
//   convolve_wf(w, k, 'v')[j] = sum_i  w[i] * kr[i - j],   kr[t] = k[m - 1 - t],   0 <= j <= n - m          (convolutions.py:14-72)
//
// For a batch that is  Out[W x p] = X[W x n] . T[n x p]  with the Toeplitz matrix T[i][j] = kr[i - j] of the taps -- the same for every
// waveform and 95 % dense (5792 of the 6092 rows of a column are inside the kernel).  SURVEY.md H4: this config is bound by FP32 multiply-adds
// (145 flop per byte), and MI355X_MICROARCH.md prices the two ways to issue them: 52 TFLOP/s for a tuned VALU kernel (where the waveform VM's
// direct-form op already is, 51 TFLOP/s), 155 for v_mfma_f32_16x16x4_f32 / 32x32x2_f32, which are exact float32 fused multiply-add chains.
//
// One workgroup (8 wavefronts) = 64 waveforms x one kernel's outputs (up to 320 columns):
//   * A = the rows, 32 samples per stage, global -> registers -> LDS (double buffered, pitch 36: conflict-free 16-byte fragment reads),
//     baseline subtracted and samples beyond the slice zeroed on the way;
//   * B is never materialised: T's fragment for (k, column) is the tap kr[k - column], read from ONE zero-margined copy of the reversed
//     kernel in LDS at a per-lane address (consecutive lanes, consecutive addresses: conflict-free);
//   * each wavefront owns 32 rows x 80 columns = 2 x 5 tiles of v_mfma_f32_16x16x4_f32, the k index inside a 16-sample group permuted
//     (lane group h takes k = 4h .. 4h+3) so that a lane's four A values are one ds_read_b128;
//   * ACCURACY: a 6092-term float32 chain is off by 1.2 - 5.5e-6 of the peak on this data (measured, tools/fir_accuracy.py) -- outside
//     the 1e-6 bar.  The matrix cores therefore accumulate 128 samples at a time and the partial sums are added in float64 (VALU, 3 % of
//     the matrix time): 1e-7 of the peak, like the VM's op (64-tap float32 blocks into float64 totals);
//   * epilogue: numpy.amax over the valid columns (NaN if any output is NaN), one float per waveform and kernel.
// Rows holding an infinity (but no NaN) are recomputed tap by tap: multiplied into the zeros outside its window an infinity would turn
// outputs into NaN that the reference leaves finite or infinite.
#include <hip/hip_runtime.h>

#include "dsp_program.h"
#include "dsp_wave.h"

#define FIR_LDS __attribute__((address_space(3)))
#define FIR_GLOBAL __attribute__((address_space(1)))
#define FIR_KARG __attribute__((address_space(4)))

namespace {

constexpr int BM = 64, BN = 320, BK = 32, APITCH = 36, KCHUNK = 128;
constexpr int MT = 2, NT = 5;  // 16 x 16 tiles per wavefront: 32 rows x 80 columns

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <int IN>
__global__ void __launch_bounds__(512, 2) dsp_fir_mfma_kernel(FirArgs A_, int64_t n_wf) {
    const FIR_KARG FirArgs& A = *(const FIR_KARG FirArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    extern __shared__ __attribute__((aligned(16))) float fir_smem[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;  // 2 x 4 wavefronts: rows 32 wm .., columns 80 wn ..
    const int q = (int)blockIdx.x;            // which kernel
    const int64_t row0 = (int64_t)blockIdx.y * BM;
    const int n = A.n, m = A.m[q], p = A.p[q], kend = A.kend;
    const int TZ = BN + kend;  // zero margin of BN below tap 0: index BN + k - column is never negative

    FIR_LDS float* tapz = (FIR_LDS float*)fir_smem;
    FIR_LDS float* As = tapz + ((TZ + 3) & ~3);
    FIR_LDS float* red = As + 2 * BM * APITCH;  // [BM][4][2]

    // ---- the reversed kernel with zero margins
    {
        const FIR_GLOBAL float* kp = (const FIR_GLOBAL float*)A.taps[q];
        for (int idx = tid; idx < TZ; idx += 512) {
            const int t = idx - BN;
            tapz[idx] = (t >= 0 && t < m) ? kp[m - 1 - t] : 0.0f;
        }
    }
    // ---- staging: thread t carries 4 consecutive samples of row t / 8
    const int srow = tid >> 3, skc = (tid & 7) * 4;
    const int64_t grow = row0 + srow < n_wf ? row0 + srow : n_wf - 1;
    constexpr int ESZ = IN == 0 ? 4 : 2;
    const FIR_GLOBAL char* rowp = (const FIR_GLOBAL char*)A.wf + (grow * A.wf_stride + A.wf_offset) * ESZ;
    const float bl = A.sub_mode ? (A.bl ? ((const FIR_GLOBAL float*)A.bl)[grow * A.bl_stride] : A.bl_const) : 0.0f;
    const bool sub = A.sub_mode != 0;
    bool bad = false, has_nan = false;
    f4 stage_v;
    auto fetch = [&](int k0) {
        const int k = k0 + skc;
        if (IN == 0) {
            stage_v = *(const FIR_GLOBAL f4*)(rowp + (size_t)k * 4);
        } else {
            const u2 raw = *(const FIR_GLOBAL u2*)(rowp + (size_t)k * 2);
            stage_v[0] = IN == 1 ? (float)(short)(raw[0] & 0xffffu) : (float)(raw[0] & 0xffffu);
            stage_v[1] = IN == 1 ? (float)(short)(raw[0] >> 16) : (float)(raw[0] >> 16);
            stage_v[2] = IN == 1 ? (float)(short)(raw[1] & 0xffffu) : (float)(raw[1] & 0xffffu);
            stage_v[3] = IN == 1 ? (float)(short)(raw[1] >> 16) : (float)(raw[1] >> 16);
        }
    };
    auto commit = [&](int k0, int buf) {  // baseline, screening, zeros beyond the slice, into LDS
        f4 v = stage_v;
        const int k = k0 + skc;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float x = sub ? v[u] - bl : v[u];
            if (k + u >= n) x = 0.0f;
            has_nan |= (x != x);
            bad |= !(__builtin_fabsf(x) <= 3.4028234663852886e38f);
            v[u] = x;
        }
        *(FIR_LDS f4*)(As + buf * BM * APITCH + srow * APITCH + skc) = v;
    };

    typedef float acc_t __attribute__((ext_vector_type(4)));
    acc_t acc[MT][NT];
    double tot[MT][NT][4];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            acc[a][b] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[a][b][r] = 0.0;
        }

    const int j = lane & 15, h4 = lane >> 4;
    // A fragment: rows 32 wm + 16 tm + j, samples 16 g + 4 h4 .. + 3 of the stage
    const int a_off = (wm * 32 + j) * APITCH + 4 * h4;
    // B fragment: tapz[BN + k - column], k = k0 + 16 g + 4 h4 + s, column = 80 wn + 16 tn + j
    int t_off = BN + 4 * h4 - (wn * 80 + j);

    fetch(0);
    commit(0, 0);
    __syncthreads();
    const int n_stage = kend / BK;
    // (two loops: the float64 flush sits between runs of KCHUNK / BK stages, outside the stage loop -- as a conditional inside it the compiler
    // copies the 120 accumulator registers around the branch in every stage: a vector move per matrix instruction)
    for (int st0 = 0; st0 < n_stage; st0 += KCHUNK / BK) {
        const int st1 = st0 + KCHUNK / BK < n_stage ? st0 + KCHUNK / BK : n_stage;
        for (int st = st0; st < st1; ++st) {
            const int buf = st & 1, k0 = st * BK;
            if (st + 1 < n_stage) fetch(k0 + BK);
            const FIR_LDS float* ab = As + buf * BM * APITCH + a_off;
#pragma unroll
            for (int g = 0; g < BK / 16; ++g) {
                f4 a[MT];
#pragma unroll
                for (int tm = 0; tm < MT; ++tm) a[tm] = *(const FIR_LDS f4*)(ab + tm * 16 * APITCH + g * 16);
                float b[NT][4];
                const FIR_LDS float* tb = tapz + t_off + k0 + g * 16;
#pragma unroll
                for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[tn][s] = tb[s - tn * 16];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
                        for (int tn = 0; tn < NT; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
            }
            if (st + 1 < n_stage) commit(k0 + BK, buf ^ 1);
            __syncthreads();
        }
        // partial sums of (up to) 128 samples leave float32 here
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tot[tm][tn][r] += (double)acc[tm][tn][r];
                acc[tm][tn] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
            }
    }
    // ---- numpy.amax over the valid columns of every row; C layout of the 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float vmax = -__builtin_inff();
            bool vnan = false;
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
                const float v = (float)(tot[tm][tn][r] + (double)acc[tm][tn][r]);
                const bool valid = wn * 80 + tn * 16 + j < p;
                vnan |= valid && (v != v);
                vmax = (valid && v > vmax) ? v : vmax;
            }
            float fn = vnan ? 1.0f : 0.0f;
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) {
                vmax = fmaxf(vmax, __shfl_xor(vmax, sft));
                fn = fmaxf(fn, __shfl_xor(fn, sft));
            }
            if (j == 0) {
                const int rl = wm * 32 + tm * 16 + h4 * 4 + r;
                red[(rl * 4 + wn) * 2] = vmax;
                red[(rl * 4 + wn) * 2 + 1] = fn;
            }
        }
    // samples of the waveform outside the slice: bl_subtract's "NaN anywhere" covers them (DSP_OP_LOAD ip[0..1])
    if (IN == 0 && sub) {
#pragma unroll 1
        for (int part = 0; part < 2; ++part) {
            const int cnt = part == 0 ? A.scan_before : A.scan_after;
            const FIR_GLOBAL float* sp = (const FIR_GLOBAL float*)rowp + (part == 0 ? -cnt : n);
            for (int e = tid & 7; e < cnt; e += 8) {
                const float x = sp[e];
                has_nan |= (x != x);
            }
        }
    }
    // row screening: the 8 threads of a row agree
    unsigned flags = (bad ? 1u : 0u) | (has_nan ? 2u : 0u);
#pragma unroll
    for (int sft = 1; sft < 8; sft <<= 1) flags |= (unsigned)__shfl_xor((int)flags, sft);
    __syncthreads();
    FIR_LDS unsigned* rowflag = (FIR_LDS unsigned*)(As);  // (the A buffers are free now)
    if ((tid & 7) == 0) rowflag[srow] = flags;
    __syncthreads();
    FIR_GLOBAL float* outp = (FIR_GLOBAL float*)A.out[q];
    if (tid < BM && row0 + tid < n_wf) {
        float vmax = -__builtin_inff(), fn = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            vmax = fmaxf(vmax, red[(tid * 4 + w) * 2]);
            fn = fmaxf(fn, red[(tid * 4 + w) * 2 + 1]);
        }
        const unsigned f = rowflag[tid];
        float res = (fn != 0.0f || (f & 2u)) ? quiet_nan<float>() : vmax;
        if ((f & 1u) && !(f & 2u)) res = 0.0f;  // an infinity in the row: decided below
        outp[(row0 + tid) * A.out_stride[q]] = res;
    }
    // ---- rows with an infinity (no NaN): tap by tap, the way np.convolve sees them
    for (int rl = 0; rl < BM; ++rl) {
        const unsigned f = rowflag[rl];  // (uniform)
        if (!(f & 1u) || (f & 2u) || row0 + rl >= n_wf) continue;
        __syncthreads();
        const FIR_GLOBAL char* rp = (const FIR_GLOBAL char*)A.wf + ((row0 + rl) * A.wf_stride + A.wf_offset) * ESZ;
        const float rbl = A.sub_mode ? (A.bl ? ((const FIR_GLOBAL float*)A.bl)[(row0 + rl) * A.bl_stride] : A.bl_const) : 0.0f;
        float vmax = -__builtin_inff();
        bool vnan = false;
        for (int jo = tid; jo < p; jo += 512) {
            float s = 0.0f;
            for (int t = 0; t < m; ++t) {
                float x = IN == 0 ? ((const FIR_GLOBAL float*)rp)[jo + t]
                                  : (IN == 1 ? (float)((const FIR_GLOBAL short*)rp)[jo + t] : (float)((const FIR_GLOBAL unsigned short*)rp)[jo + t]);
                if (sub) x = x - rbl;
                s = __builtin_fmaf(x, tapz[BN + t], s);
            }
            vnan |= (s != s);
            vmax = s > vmax ? s : vmax;
        }
        float fn = vnan ? 1.0f : 0.0f;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            vmax = fmaxf(vmax, __shfl_xor(vmax, sft));
            fn = fmaxf(fn, __shfl_xor(fn, sft));
        }
        if (lane == 0) {
            red[wave * 2] = vmax;
            red[wave * 2 + 1] = fn;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 8; ++w) {
                vmax = fmaxf(vmax, red[w * 2]);
                fn = fmaxf(fn, red[w * 2 + 1]);
            }
            outp[(row0 + rl) * A.out_stride[q]] = fn != 0.0f ? quiet_nan<float>() : vmax;
        }
    }
}

// ---- the same product with the outputs kept: convolve_wf in any mode whose result is a waveform other processors read (the t0 filter of
// the Ge recipes: 133 taps, 'same', 8192 outputs).  Output c sums the samples c - d .. c - d + m - 1 (d = 0 'valid', m / 2 'same', m - 1
// 'full'; samples outside the waveform are the zeros np.convolve pads with), so a tile of 320 columns is the product of the 64 x K window
// of the rows that starts at sample 320 ct - d (rounded down to a multiple of 4 for the 16-byte loads, the remainder e moves into the tap
// index) with the same zero-margined reversed kernel; K = 320 + m - 1 + e rounded up to 32.  A wavefront's 80 columns meet the band of a
// short kernel only in some of the stages: the others are skipped (the tile's barriers stay).  No screening here: dsp_fir_fixup_kernel
// looks at the rows afterwards and rewrites the ones holding a NaN (all NaN, convolutions.py:40-43) or an infinity (tap by tap).
constexpr int TB = BN + 4, SCHUNK = 64;

template <int IN>
__global__ void __launch_bounds__(512, 2) dsp_fir_store_kernel(FirArgs A_, int64_t n_wf) {
    const FIR_KARG FirArgs& A = *(const FIR_KARG FirArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    extern __shared__ __attribute__((aligned(16))) float fir_smem[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int c0 = (int)blockIdx.x * BN;
    const int64_t row0 = (int64_t)blockIdx.y * BM;
    const int n = A.n, m = A.m[0], P = A.p[0];
    const int s0 = c0 - A.dshift;
    const int ks = (s0 >> 2) << 2, e = s0 - ks;  // (arithmetic shift: rounds down for the negative start of the first tiles)
    const int cols = P - c0 < BN ? P - c0 : BN;
    const int kt = ((cols + m - 1 + e + BK - 1) / BK) * BK;

    FIR_LDS float* tapz = (FIR_LDS float*)fir_smem;
    FIR_LDS float* As = tapz + ((TB + A.kend + 3) & ~3);  // (kend: the longest window of any tile, the host sized the margins for it)
    {
        const FIR_GLOBAL float* kp = (const FIR_GLOBAL float*)A.taps[0];
        for (int idx = tid; idx < TB + kt; idx += 512) {
            const int t = idx - TB;
            tapz[idx] = (t >= 0 && t < m) ? kp[m - 1 - t] : 0.0f;
        }
    }
    const int srow = tid >> 3, skc = (tid & 7) * 4;
    const int64_t grow = row0 + srow < n_wf ? row0 + srow : n_wf - 1;
    constexpr int ESZ = IN == 0 ? 4 : 2;
    const FIR_GLOBAL char* rowp = (const FIR_GLOBAL char*)A.wf + (grow * A.wf_stride + A.wf_offset) * ESZ;
    const float bl = A.sub_mode ? (A.bl ? ((const FIR_GLOBAL float*)A.bl)[grow * A.bl_stride] : A.bl_const) : 0.0f;
    const bool sub = A.sub_mode != 0;
    f4 stage_v;
    auto one = [&](int i) -> float {
        if (i < 0 || i >= n) return 0.0f;
        return IN == 0 ? ((const FIR_GLOBAL float*)rowp)[i] : (IN == 1 ? (float)((const FIR_GLOBAL short*)rowp)[i] : (float)((const FIR_GLOBAL unsigned short*)rowp)[i]);
    };
    auto fetch = [&](int k0) {
        const int i = ks + k0 + skc;
        if (i >= 0 && i + 4 <= n) {
            if (IN == 0) {
                stage_v = *(const FIR_GLOBAL f4*)(rowp + (size_t)i * 4);
            } else {
                const u2 raw = *(const FIR_GLOBAL u2*)(rowp + (size_t)i * 2);
                stage_v[0] = IN == 1 ? (float)(short)(raw[0] & 0xffffu) : (float)(raw[0] & 0xffffu);
                stage_v[1] = IN == 1 ? (float)(short)(raw[0] >> 16) : (float)(raw[0] >> 16);
                stage_v[2] = IN == 1 ? (float)(short)(raw[1] & 0xffffu) : (float)(raw[1] & 0xffffu);
                stage_v[3] = IN == 1 ? (float)(short)(raw[1] >> 16) : (float)(raw[1] >> 16);
            }
        } else {  // a window end: sample by sample, zeros outside the waveform
#pragma unroll
            for (int u = 0; u < 4; ++u) stage_v[u] = one(i + u);
        }
    };
    auto commit = [&](int k0, int buf) {
        f4 v = stage_v;
        const int i = ks + k0 + skc;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float x = sub ? v[u] - bl : v[u];
            v[u] = (i + u >= 0 && i + u < n) ? x : 0.0f;
        }
        *(FIR_LDS f4*)(As + buf * BM * APITCH + srow * APITCH + skc) = v;
    };

    typedef float acc_t __attribute__((ext_vector_type(4)));
    acc_t acc[MT][NT];
    double tot[MT][NT][4];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            acc[a][b] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[a][b][r] = 0.0;
        }
    const int j = lane & 15, h4 = lane >> 4;
    const int a_off = (wm * 32 + j) * APITCH + 4 * h4;
    // Column tiles are dealt out to the four wavefronts of a row block in turn (tile tn of wavefront wn: columns 16 (wn + 4 tn) ..): the band
    // of a short kernel covers about half of a tile row's 20 column tiles in any stage -- neighbouring ones -- and this way every wavefront
    // holds its share of them instead of two wavefronts holding all and two none.  A 16-column tile meets a 16-sample group of the
    // window only where a tap lies: window sample kk meets column cl at tap kk - cl - e.
    const int t_off = TB + 4 * h4 - (wn * 16 + j) - e;

    fetch(0);
    commit(0, 0);
    __syncthreads();
    const int n_stage = kt / BK;
    // (64 samples per float32 partial sum, as the waveform VM's op: short differentiating kernels cancel; the flush between runs of stages, not
    // a conditional inside the stage loop -- see dsp_fir_mfma_kernel)
    for (int st0 = 0; st0 < n_stage; st0 += SCHUNK / BK) {
        const int st1 = st0 + SCHUNK / BK < n_stage ? st0 + SCHUNK / BK : n_stage;
        for (int st = st0; st < st1; ++st) {
            const int buf = st & 1, k0 = st * BK;
            if (st + 1 < n_stage) fetch(k0 + BK);
            {
                const FIR_LDS float* ab = As + buf * BM * APITCH + a_off;
#pragma unroll
                for (int g = 0; g < BK / 16; ++g) {
                    const int kb = k0 + g * 16;  // this group: window samples kb .. kb + 15
                    f4 a[MT];
#pragma unroll
                    for (int tm = 0; tm < MT; ++tm) a[tm] = *(const FIR_LDS f4*)(ab + tm * 16 * APITCH + g * 16);
                    float b[NT][4];
                    const FIR_LDS float* tb = tapz + t_off + kb;
#pragma unroll
                    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                        for (int s = 0; s < 4; ++s) b[tn][s] = tb[s - tn * 64];
#pragma unroll
                    for (int tn = 0; tn < NT; ++tn) {
                        const int c_lo = 16 * (wn + 4 * tn) + e;  // (uniform: a scalar branch around the tile's eight matrix instructions)
                        if (kb + 15 >= c_lo && kb <= c_lo + 14 + m) {
#pragma unroll
                            for (int s = 0; s < 4; ++s)
#pragma unroll
                                for (int tm = 0; tm < MT; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
                        }
                    }
                }
            }
            if (st + 1 < n_stage) commit(k0 + BK, buf ^ 1);
            __syncthreads();
        }
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tot[tm][tn][r] += (double)acc[tm][tn][r];
                acc[tm][tn] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
            }
    }
    // ---- the tile's outputs; C layout of a 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + r
    FIR_GLOBAL float* outp = (FIR_GLOBAL float*)A.out[0];
#pragma unroll
    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + wm * 32 + tm * 16 + h4 * 4 + r;
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
                const int cl = 16 * (wn + 4 * tn) + j;
                if (row < n_wf && cl < cols) outp[row * A.out_stride[0] + c0 + cl] = (float)(tot[tm][tn][r] + (double)acc[tm][tn][r]);
            }
        }
}

// One wavefront per row, after dsp_fir_store_kernel: a row with a NaN (in the slice, or anywhere in the waveform when bl_subtract's rule
// applies: scan_before / scan_after) becomes all NaN; a row with an infinity is recomputed tap by tap over the samples np.convolve
// multiplies (the product above also multiplied it into the zeros of other windows).
template <int IN>
__global__ void __launch_bounds__(256) dsp_fir_fixup_kernel(FirArgs A_, int64_t n_wf) {
    const FIR_KARG FirArgs& A = *(const FIR_KARG FirArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    const int lane = (int)threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (row >= n_wf) return;
    constexpr int ESZ = IN == 0 ? 4 : 2;
    const int n = A.n, m = A.m[0], P = A.p[0], d = A.dshift;
    const FIR_GLOBAL char* rowp = (const FIR_GLOBAL char*)A.wf + (row * A.wf_stride + A.wf_offset) * ESZ;
    const float bl = A.sub_mode ? (A.bl ? ((const FIR_GLOBAL float*)A.bl)[row * A.bl_stride] : A.bl_const) : 0.0f;
    const bool sub = A.sub_mode != 0;
    auto at = [&](int i) -> float {
        const float x = IN == 0 ? ((const FIR_GLOBAL float*)rowp)[i] : (IN == 1 ? (float)((const FIR_GLOBAL short*)rowp)[i] : (float)((const FIR_GLOBAL unsigned short*)rowp)[i]);
        return sub ? x - bl : x;
    };
    bool has_nan = false, has_inf = false;
    if (A.row_flags) {  // (the float16 form's first pass screened the rows: bit 0 an infinity, bit 1 a NaN)
        const unsigned f = ((const FIR_GLOBAL unsigned*)A.row_flags)[row];
        has_inf = (f & 1u) != 0;
        has_nan = (f & 2u) != 0;
    } else if (IN == 0) {
        const int n4 = n & ~3;
        for (int i = lane * 4; i < n4; i += 256) {
            const f4 v = *(const FIR_GLOBAL f4*)(rowp + (size_t)i * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x = sub ? v[u] - bl : v[u];
                has_nan |= (x != x);
                has_inf |= !(__builtin_fabsf(x) <= 3.4028234663852886e38f);
            }
        }
        for (int i = n4 + lane; i < n; i += 64) {
            const float x = at(i);
            has_nan |= (x != x);
            has_inf |= !(__builtin_fabsf(x) <= 3.4028234663852886e38f);
        }
        if (sub) {
            for (int i = -A.scan_before + lane; i < 0; i += 64) has_nan |= (at(i) != at(i));
            for (int i = n + lane; i < n + A.scan_after; i += 64) has_nan |= (at(i) != at(i));
        }
    } else {  // integer samples: only the baseline can be a NaN or an infinity
        has_nan = sub && (bl != bl);
        has_inf = sub && !(__builtin_fabsf(bl) <= 3.4028234663852886e38f);
    }
    has_nan = __any(has_nan);
    has_inf = __any(has_inf);
    if (!has_nan && !has_inf) return;
    FIR_GLOBAL float* outp = (FIR_GLOBAL float*)A.out[0] + row * A.out_stride[0];
    if (has_nan) {
        for (int c = lane; c < P; c += 64) outp[c] = quiet_nan<float>();
        return;
    }
    const FIR_GLOBAL float* kp = (const FIR_GLOBAL float*)A.taps[0];
    for (int c = lane; c < P; c += 64) {
        float s = 0.0f;
        for (int t = 0; t < m; ++t) {  // sample c - d + t meets the reversed kernel's tap t = kernel[m - 1 - t]
            const int i = c - d + t;
            if (i >= 0 && i < n) s = __builtin_fmaf(at(i), kp[m - 1 - t], s);
        }
        outp[c] = s;
    }
}

}  // namespace

extern "C" int dsp_internal_fir_store_lds_bytes(int kend) { return (((TB + kend + 3) & ~3) + 2 * BM * APITCH) * 4; }

extern "C" int dsp_internal_launch_fir_store(const FirArgs* A, int64_t n_wf, int lds_bytes, hipStream_t stream) {
    if (n_wf <= 0 || A->p[0] <= 0) return 0;
    const dim3 grid((unsigned)((A->p[0] + BN - 1) / BN), (unsigned)((n_wf + BM - 1) / BM));
    const dim3 fix((unsigned)((n_wf + 3) / 4));
    switch (A->in_kind) {
        case 0:
            hipLaunchKernelGGL(dsp_fir_store_kernel<0>, grid, dim3(512), lds_bytes, stream, *A, n_wf);
            hipLaunchKernelGGL(dsp_fir_fixup_kernel<0>, fix, dim3(256), 0, stream, *A, n_wf);
            break;
        case 1:
            hipLaunchKernelGGL(dsp_fir_store_kernel<1>, grid, dim3(512), lds_bytes, stream, *A, n_wf);
            hipLaunchKernelGGL(dsp_fir_fixup_kernel<1>, fix, dim3(256), 0, stream, *A, n_wf);
            break;
        default:
            hipLaunchKernelGGL(dsp_fir_store_kernel<2>, grid, dim3(512), lds_bytes, stream, *A, n_wf);
            hipLaunchKernelGGL(dsp_fir_fixup_kernel<2>, fix, dim3(256), 0, stream, *A, n_wf);
            break;
    }
    return (int)hipGetLastError();
}

// the rows-with-a-NaN-or-an-infinity pass alone (behind the float16 form of the kept-output kernel, dsp_fir_f16.hip)
extern "C" int dsp_internal_fir_fixup(const FirArgs* A, int64_t n_wf, hipStream_t stream) {
    if (n_wf <= 0) return 0;
    const dim3 fix((unsigned)((n_wf + 3) / 4));
    switch (A->in_kind) {
        case 0: hipLaunchKernelGGL(dsp_fir_fixup_kernel<0>, fix, dim3(256), 0, stream, *A, n_wf); break;
        case 1: hipLaunchKernelGGL(dsp_fir_fixup_kernel<1>, fix, dim3(256), 0, stream, *A, n_wf); break;
        default: hipLaunchKernelGGL(dsp_fir_fixup_kernel<2>, fix, dim3(256), 0, stream, *A, n_wf); break;
    }
    return (int)hipGetLastError();
}

extern "C" int dsp_internal_set_fir_store_lds(int lds_bytes) {
    const void* k[3] = {reinterpret_cast<const void*>(&dsp_fir_store_kernel<0>), reinterpret_cast<const void*>(&dsp_fir_store_kernel<1>),
                        reinterpret_cast<const void*>(&dsp_fir_store_kernel<2>)};
    for (int i = 0; i < 3; ++i) {
        const int rc = (int)hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" const char* dsp_internal_fir_store_kernel_name() { return "dsp_fir_store_kernel"; }

extern "C" int dsp_internal_fir_mfma_lds_bytes(int kend) { return (((BN + kend + 3) & ~3) + 2 * BM * APITCH + BM * 4 * 2) * 4; }

extern "C" int dsp_internal_launch_fir_mfma(const FirArgs* A, int64_t n_wf, int lds_bytes, hipStream_t stream) {
    if (n_wf <= 0 || A->n_kernels <= 0) return 0;
    const dim3 grid((unsigned)A->n_kernels, (unsigned)((n_wf + BM - 1) / BM));
    switch (A->in_kind) {
        case 0: hipLaunchKernelGGL(dsp_fir_mfma_kernel<0>, grid, dim3(512), lds_bytes, stream, *A, n_wf); break;
        case 1: hipLaunchKernelGGL(dsp_fir_mfma_kernel<1>, grid, dim3(512), lds_bytes, stream, *A, n_wf); break;
        default: hipLaunchKernelGGL(dsp_fir_mfma_kernel<2>, grid, dim3(512), lds_bytes, stream, *A, n_wf); break;
    }
    return (int)hipGetLastError();
}

extern "C" int dsp_internal_set_fir_mfma_lds(int lds_bytes) {
    const void* k[3] = {reinterpret_cast<const void*>(&dsp_fir_mfma_kernel<0>), reinterpret_cast<const void*>(&dsp_fir_mfma_kernel<1>),
                        reinterpret_cast<const void*>(&dsp_fir_mfma_kernel<2>)};
    for (int i = 0; i < 3; ++i) {
        const int rc = (int)hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" const char* dsp_internal_fir_mfma_kernel_name() { return "dsp_fir_mfma_kernel"; }
